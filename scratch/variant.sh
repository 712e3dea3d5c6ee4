#!/bin/bash
# scratch/variant.sh <tag> <extra -D flags...>: build a variant of the library as raytracer_amd/librt_mi355x_<tag>.so
tag=$1; shift
cd $(dirname $0)/../raytracer_amd/csrc || exit 1
mkdir -p build_var/$tag
F="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -ffp-contract=off"
/opt/rocm/bin/hipcc $F "$@" -c render_wavefront.hip -o build_var/$tag/render_wavefront.o || exit 1
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 build/api.o build/host_config.o build/host_image.o build/host_scene.o build/host_bvh.o build/render_mega.o build/bvh_build.o build/bvh_order.o build/bvh_sah.o build/comm.o build_var/$tag/render_wavefront.o -ldl -o ../librt_mi355x_$tag.so
