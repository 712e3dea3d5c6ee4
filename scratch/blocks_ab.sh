#!/bin/bash
B="python3 bench.py --no-cpu-baseline --no-configs"
for v in 4 3 2; do
  for s in 2 3; do
  RT_TRAVERSE_BLOCKS_PER_CU=$v timeout -k 10 200 $B --launches-in-flight $s 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('blocks/CU $v in flight $s: %.1f Mrays/s  %.4f ms/frame   batch kernel %.3f ms alone, %.3f in flight; one-frame kernel %.3f'%(d['value'], d['ms_per_step'], r['kernel_ms'], r['kernel_ms_in_flight'], r['kernel_ms_one_frame_per_launch']))" || exit 1
  done
done
