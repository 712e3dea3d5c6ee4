#!/bin/bash
# scratch/ab3.sh "<cases>" <lib suffixes...>: secondary configs against variant builds
cases=$1; shift
for v in "$@"; do
  echo "== $v"
  RT_MI355X_LIB=$GRAFT_REPO_ROOT/raytracer_amd/librt_mi355x_$v.so timeout -k 10 300 python scratch/bench_configs.py $cases 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); print('  %-16s kernel %.1f ms traverse %.1f ms  %.0f Mrays/s' % (d['case'], d['kernel_ms'], d['traverse_ms'], d['Mrays_per_s']))" || exit 1
done
