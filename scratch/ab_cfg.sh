#!/bin/bash
# scratch/ab_cfg.sh <cases...>: scratch/bench_configs.py for the default library and every variant library (scratch/variant.sh)
cd $GRAFT_REPO_ROOT
for lib in "" $(ls raytracer_amd/librt_mi355x_*.so 2>/dev/null); do
  echo "== ${lib:-default}"
  export RT_MI355X_LIB=${lib:+$PWD/$lib}
  timeout -k 10 200 python3 scratch/bench_configs.py "$@" 2>/dev/null | python3 -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('   %-20s kernel %.3f ms' % (d['case'], d['kernel_ms']))" || exit 1
done
