#!/bin/bash
# scratch/ab2.sh <script> <lib suffixes...>: run a scratch script against variant builds
scr=$1; shift
for v in "$@"; do
  echo "== $v"
  if [ "$v" = "default" ]; then timeout -k 10 200 python $scr || exit 1; else RT_MI355X_LIB=$GRAFT_REPO_ROOT/raytracer_amd/librt_mi355x_$v.so timeout -k 10 200 python $scr || exit 1; fi
done
