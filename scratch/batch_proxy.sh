#!/bin/bash
# proxy for batching frames into one launch: the C2 scene at 4x / 16x the rays per launch, one frame at a time
B="python3 bench.py --no-cpu-baseline --no-configs --steps 10 --warmup 2"
for wh in "1920 1080 1" "1920 1080 8" "3840 2160 1" "3840 2160 2" "7680 4320 1"; do
  set -- $wh
  timeout -k 10 200 $B --width $1 --height $2 --frames-in-flight $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1x$2 inflight $3: ms/frame %.3f  ms per Mray %.4f  kernel_ms %.3f nodes/ray %.1f'%(d['ms_per_step'], d['ms_per_step']/(d['config']['rays_per_frame']/1e6), d['roofline']['kernel_ms'], d['config']['node_fetches_per_ray']))" || exit 1
done
