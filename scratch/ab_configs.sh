#!/bin/bash
# scratch/ab_configs.sh <cases...>: bench_configs for the default library and every variant library present
for lib in "" $(ls raytracer_amd/librt_mi355x_*.so 2>/dev/null); do
  echo "== ${lib:-default}"
  RT_MI355X_LIB=${lib:+$PWD/$lib} timeout -k 10 300 python scratch/bench_configs.py "$@" 2>&1 | grep config | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-22s %9.3f ms  trav %8.3f  %8.1f Mrays/s  rays %d' % (d['case'], d['kernel_ms'], d['traverse_ms'], d['Mrays_per_s'], d['rays']))"
done
