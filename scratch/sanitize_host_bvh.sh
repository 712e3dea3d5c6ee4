#!/bin/bash
# Host-only sanitizer runs of the threaded hierarchy builders (host_bvh.cpp): ThreadSanitizer, then ASan + UBSan.
set -e
cd "$(dirname "$0")/.."
I="-Iraytracer_amd/csrc -Iinclude"
g++ -std=c++17 -O1 -g -fsanitize=thread $I scratch/sanitize_host_bvh.cpp raytracer_amd/csrc/host_bvh.cpp -o /tmp/host_bvh_tsan -lpthread
RT_BUILD_THREADS=8 /tmp/host_bvh_tsan
g++ -std=c++17 -O1 -g -fsanitize=address,undefined $I scratch/sanitize_host_bvh.cpp raytracer_amd/csrc/host_bvh.cpp -o /tmp/host_bvh_asan -lpthread
RT_BUILD_THREADS=8 /tmp/host_bvh_asan
