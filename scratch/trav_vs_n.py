"""k_traverse time against the number of rays (C2 frame cut into N parts, part 0 rendered)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import workloads
rt.init(0)
cfg = workloads.default_config()
scene = rt.Scene(workloads.bvh_stress_scene())
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
buf = torch.empty(1080 * 1920 * 3, dtype=torch.uint8, device=dev)
cnt = torch.zeros(8, dtype=torch.int64, device=dev)
for parts in [int(a) for a in sys.argv[1:]] or (1, 2, 4, 8, 16, 32, 64, 135):
    p = cfg.params(rng="d0"); p.tile_rows, p.part_index, p.part_count = 8, 0, parts
    best = None
    for it in range(6):
        cnt.zero_()
        scene.render_device(p, buf.data_ptr(), cnt.data_ptr(), collect_counters=(it == 0), stream=stream.cuda_stream)
        torch.cuda.synchronize()
        c = cnt.cpu().tolist()
        ms, nl = scene.traverse_time(stream.cuda_stream)
        if it == 0: c0 = c
        else: best = ms * 1e6 if best is None else min(best, ms * 1e6)
    print("parts %3d rays %8d traverse %.3f ms | fetches/ray %.1f max fetches of a ray %d, max iterations of a wave %d -> %.2f us/iteration"
          % (parts, c0[0], best * 1e-6, c0[4] / max(1, c0[0]), c0[5], c0[6], best * 1e-3 / max(1, c0[6])), flush=True)
