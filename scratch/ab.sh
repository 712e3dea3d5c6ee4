#!/bin/bash
# A/B of library builds: scratch/ab.sh <lib suffixes...>   (env: STEPS, FLIGHT)
for v in "$@"; do
  echo "== $v"
  RT_MI355X_LIB=$GRAFT_REPO_ROOT/raytracer_amd/librt_mi355x_$v.so timeout -k 10 200 python bench.py --steps ${STEPS:-24} --warmup 1 --no-cpu-baseline --frames-in-flight ${FLIGHT:-1} 2>/dev/null | python scratch/line.py || exit 1
done
