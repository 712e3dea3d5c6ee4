#!/bin/bash
# scratch/ab_pairs.sh: chunks two at a time (default) vs one at a time, on the many-chunk configs
set -e
for mode in pairs one; do
  echo "== $mode"
  if [ $mode = one ]; then export RT_ONE_CHUNK_AT_A_TIME=1; fi
  timeout -k 10 400 python scratch/bench_configs.py st_dof st_motion_blur c4_combined gal_shapes 2>&1 | grep config | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-22s %9.3f ms  wall %9.3f  %8.1f Mrays/s  rays %d' % (d['case'], d['kernel_ms'], d['wall_ms'], d['Mrays_per_s'], d['rays']))"
done
