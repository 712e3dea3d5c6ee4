#!/bin/bash
# scratch/ab_one.sh <lib suffixes...>: batched bench, one launch in flight
for v in "$@"; do
  RT_MI355X_LIB=$GRAFT_REPO_ROOT/raytracer_amd/librt_mi355x_$v.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-configs 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-10s %.1f Mrays/s  %.4f ms/frame  batch kernel %.3f ms  one-frame kernel %.3f  solo %.3f'%('$v', d['value'], d['ms_per_step'], r['kernel_ms'], r['kernel_ms_one_frame_per_launch'], d['ms_per_frame_solo']))" || exit 1
done
