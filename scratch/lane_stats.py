"""scratch/lane_stats.py: lanes of the node steps of k_traverse by state.  Needs a library built with the instrumentation:
scratch/variant.sh lanes -DRT_LANE_STATS, then RT_MI355X_LIB=raytracer_amd/librt_mi355x_lanes.so python scratch/lane_stats.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import workloads
rt.init(0)
dev = torch.device("cuda", 0)
for w, h in ((1920, 1080), (7680, 4320)):
    scene = rt.Scene(workloads.bvh_stress_scene(1_000_000, w, h))
    p = workloads.default_config().params(rng="seeded", seed=1)
    buf = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
    c = torch.zeros(8, dtype=torch.int64, device=dev)
    scene.render_device(p, buf.data_ptr(), c.data_ptr(), collect_counters=True)
    torch.cuda.synchronize()
    v = c.cpu().numpy()
    rays = w * h                       # one ray per pixel; the library adds its word 7 (here: the steps) to the ray counter
    steps = max(1, int(v[0]) - rays)
    print("%dx%d: rays %d, node steps of a wave %d, idle lanes per step %.1f, lanes without node work %.1f, lanes at a node %.1f of 64"
          % (w, h, rays, steps, v[5] / steps, v[6] / steps, 64 - (v[5] + v[6]) / steps))
    scene.close()
