"""scratch/fp32_variant.py [spheres]: the C2 frame with RT_SCHEDULE_WAVEFRONT and RT_SCHEDULE_WAVEFRONT_FP32: ms per frame at 16
frames per launch, and the pixels of the FP32 frame that differ from the parity frame by more than one level."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as rt
from raytracer_amd import workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rt.init(0)
dev = torch.device("cuda", 0)
scene = rt.Scene(workloads.bvh_stress_scene(n, 1920, 1080))
stream = torch.cuda.Stream(dev)
F = 16
buf = torch.zeros((F, 1080, 1920, 3), dtype=torch.uint8, device=dev)
frames = {}
for sampling in ("d0", "seeded"):
    for name, sched in (("fp64", rt.RT_SCHEDULE_WAVEFRONT), ("fp32", rt.RT_SCHEDULE_WAVEFRONT_FP32)):
        p = workloads.default_config().params(rng=sampling, seed=12345)
        p.schedule = sched
        for _ in range(3):
            scene.render_frames_device(p, F, buf.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize()
        t = time.perf_counter()
        reps = 8
        for _ in range(reps):
            scene.render_frames_device(p, F, buf.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) * 1e3 / (reps * F)
        frames[name] = buf[0].cpu().numpy().astype(np.int32)
        img, st = scene.render(p, stats=True)
        print("%s %s: %.4f ms per frame (%.1f Mrays/s), prim tests per ray %.3f, node fetches per ray %.2f" % (sampling, name, ms, 1920 * 1080 / ms / 1e3, st.prims_tested / st.rays, st.nodes_fetched / st.rays), flush=True)
    d = np.abs(frames["fp64"] - frames["fp32"]).max(axis=2)
    print("%s: pixels off by more than one level: %d of %d (max |diff| %d), by exactly one: %d" % (sampling, int((d > 1).sum()), d.size, int(d.max()), int((d == 1).sum())), flush=True)
scene.close()
