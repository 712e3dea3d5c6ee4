"""scratch/frames_timing.py: ms per frame of the C2 frame rendered F frames per launch (rt_render_frames_device), for a whole
frame and for part 0 of N (what one rank of an N-GPU job renders), against F single-frame renders in flight on F streams."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import workloads
rt.init(0)
dev = torch.device("cuda", 0)
scene = rt.Scene(workloads.bvh_stress_scene(1_000_000, 1920, 1080))
base = workloads.default_config().params(rng="d0")
streams = [torch.cuda.Stream(dev) for _ in range(2)]
for parts in (1, 2, 4, 8):
    p = base.copy()
    p.tile_rows, p.part_index, p.part_count = 8, 0, parts
    rows = rt.part_rows(1080, 8, 0, parts)
    for F in (1, 2, 4, 8, 16, 32):
        bufs = [torch.zeros((F, rows, 1920, 3), dtype=torch.uint8, device=dev) for _ in streams]
        torch.cuda.synchronize()
        def run(n):
            for i in range(n):
                s = streams[i % 2]
                scene.render_frames_device(p, F, bufs[i % 2].data_ptr(), stream=s.cuda_stream)
        run(4)
        torch.cuda.synchronize()
        n = max(6, 64 // F)
        t = time.perf_counter()
        run(n)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) * 1e3 / (n * F)
        print("parts %d  frames/launch %2d  (2 launches in flight): %.4f ms per frame share  -> %.1f Mrays/s per GPU" % (parts, F, ms, rows * 1920 / ms / 1e3), flush=True)
