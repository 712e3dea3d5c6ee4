#!/bin/bash
# Builds and runs scratch/wide_model.cpp on the C2 scene (host only).  Usage: scratch/wide_model.sh [spheres] [pixel_step]
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch/build
g++ -O2 -std=c++17 -fopenmp -fPIC -shared scratch/wide_model.cpp raytracer_amd/csrc/host_bvh.cpp -o scratch/build/wide_model.so
python - "$@" <<'P'
import ctypes as C, sys
from raytracer_amd import workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
step = int(sys.argv[2]) if len(sys.argv) > 2 else 4
scene = workloads.bvh_stress_scene(n)
lib = C.CDLL("scratch/build/wide_model.so")
out = (C.c_double * 28)()
lib.wide_model(C.cast(scene.desc_ptr(), C.c_void_p), step, out)
P
