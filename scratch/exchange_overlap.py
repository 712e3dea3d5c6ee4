"""scratch/exchange_overlap.py: does an RCCL exchange overlap a render on the same GPU?  One rank that sends its slices to itself
(RT_COMM_LOOPBACK=1): launches of F C2 frames on one stream, their exchange (grouped ncclSend / ncclRecv + k_deinterleave) on
another, S slice sets alternating (what bench.py does on every rank of a multi-GPU job).  Prints ms per frame with and without
the exchange.  Environment: RT_TRAVERSE_SPARE_BLOCKS, RT_TRAVERSE_BLOCKS_PER_CU."""
import os, sys, time
os.environ["RT_COMM_LOOPBACK"] = "1"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import workloads
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rt.init(0)
dev = torch.device("cuda", 0)
rt.comm_init(0, 1, rt.comm_unique_id())
scene = rt.Scene(workloads.bvh_stress_scene(1_000_000, 1920, 1080))
p = workloads.default_config().params(rng="seeded", seed=1)
render, gather = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
for S in (2, 3):
    slices = [torch.zeros((F, 1080, 1920, 3), dtype=torch.uint8, device=dev) for _ in range(S)]
    frames = [torch.zeros((F, 1080, 1920, 3), dtype=torch.uint8, device=dev) for _ in range(S)]
    for exchange in (False, True):
        done = [None] * S
        def run(n):
            for k in range(n):
                s = k % S
                with torch.cuda.stream(render):
                    if done[s] is not None:
                        render.wait_event(done[s])
                    scene.render_frames_device(p, F, slices[s].data_ptr(), stream=render.cuda_stream)
                    ev = torch.cuda.Event(); ev.record(render)
                if exchange:
                    with torch.cuda.stream(gather):
                        gather.wait_event(ev)
                        rt.gather_frames(slices[s].data_ptr(), F, 1920, 1080, 8, frames[s].data_ptr(), stream=gather.cuda_stream)
                        done[s] = torch.cuda.Event(); done[s].record(gather)
        run(S); torch.cuda.synchronize()
        n = 12
        t = time.perf_counter(); run(n); torch.cuda.synchronize()
        ms = (time.perf_counter() - t) * 1e3 / (n * F)
        print("sets %d  %-16s %.4f ms per frame" % (S, "with exchange" if exchange else "render only", ms), flush=True)
    if exchange:
        assert torch.equal(frames[0], slices[0])
rt.comm_destroy()
