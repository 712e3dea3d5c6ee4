"""scratch/two_streams.py <case>: passes of a many-sample frame rendered one at a time on one stream, and two at a time on two
streams (each stream has its own ray queues): do the small launches and read-backs of one pass's chunks hide behind the other's?"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import threading
import torch
import raytracer_amd as rt
from raytracer_amd import workloads
name = sys.argv[1] if len(sys.argv) > 1 else "st_motion_blur"
cfgs = {c[1]: c for c in workloads.BASELINE_CONFIGS}
label, _, w, h, spp, rng = cfgs[name]
rt.init(0)
dev = torch.device("cuda", 0)
with tempfile.TemporaryDirectory() as scratch:
    sf, p = workloads.golden_scene(name, w, h, spp, rng, sky_dir=scratch)
    scene = rt.Scene(sf)
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    bufs = [torch.zeros((h, w, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    def passes(k, n):
        torch.cuda.set_device(dev)
        q = p.copy()
        for i in range(n):
            q.seed = p.seed + 2 * i + k
            scene.render_device(q, bufs[k].data_ptr(), stream=streams[k].cuda_stream)
    passes(0, 1); passes(1, 1); torch.cuda.synchronize()
    t = time.perf_counter(); passes(0, 4); torch.cuda.synchronize()
    one = (time.perf_counter() - t) * 1e3 / 4
    t = time.perf_counter()
    th = [threading.Thread(target=passes, args=(k, 4)) for k in range(2)]
    [x.start() for x in th]; [x.join() for x in th]; torch.cuda.synchronize()
    two = (time.perf_counter() - t) * 1e3 / 8
    print("%s: %.2f ms per pass one at a time, %.2f ms per pass with two streams" % (name, one, two))
    scene.close()
