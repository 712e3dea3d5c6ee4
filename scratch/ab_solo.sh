#!/bin/bash
# scratch/ab_solo.sh: one-frame-at-a-time figures of the default library and every variant library (bench.py, no baselines)
for lib in "" $(ls raytracer_amd/librt_mi355x_*.so 2>/dev/null); do
  RT_MI355X_LIB=${lib:+$PWD/$lib} timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-configs 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-40s %.1f Mrays/s  solo %.3f ms  host %.3f  kernel(16 frames) %.3f ms  kernel(1 frame) %.3f ms' % ('${lib:-default}', d['value'], d['ms_per_frame_solo'], d['ms_per_frame_host_buffer'], r['kernel_ms'], r['kernel_ms_one_frame_per_launch']))" || exit 1
done
