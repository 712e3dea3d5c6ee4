"""Per-rank step time when the C2 frame is cut into N parts (one GPU renders part 0 only), F frames in flight."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import workloads
rt.init(0)
cfg = workloads.default_config()
sd = workloads.bvh_stress_scene()
scene = rt.Scene(sd)
dev = torch.device("cuda", 0)
parts_list = [int(a) for a in os.environ.get("PARTS", "1,8").split(",")]
f_list = [int(a) for a in os.environ.get("FLIGHT", "1,2,4,8").split(",")]
streams = [torch.cuda.Stream(dev) for _ in range(max(f_list))]
bufs = [torch.empty(1080 * 1920 * 3, dtype=torch.uint8, device=dev) for _ in streams]
for parts in parts_list:
    p = cfg.params(rng="d0"); p.tile_rows, p.part_index, p.part_count = int(os.environ.get("TILE", "8")), int(os.environ.get("PART", "0")), parts
    for F in f_list:
        def run(K):
            for i in range(K):
                s = i % F
                scene.render_device(p, bufs[s].data_ptr(), stream=streams[s].cuda_stream)
        run(2 * F); torch.cuda.synchronize()
        K = 40
        t0 = time.perf_counter(); run(K); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3 / K
        tt = [scene.traverse_time(s.cuda_stream) for s in streams[:F]]
        print("parts %d  in flight %d: %.3f ms/step  (k_traverse avg %.3f ms)" % (parts, F, dt, sum(a for a, _ in tt) / max(1, sum(b for _, b in tt))), flush=True)
