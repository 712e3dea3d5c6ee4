"""scratch/frames_configs.py: the 1-spp BASELINE configs rendered 1 / 4 / 16 frames per call (rt_render_frames_device)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import workloads
rt.init(0)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev)
with tempfile.TemporaryDirectory(prefix="rt_fc_") as scratch:
    for label, name, w, h, spp, rng in workloads.BASELINE_CONFIGS:
        if spp != 1:
            continue
        sf, p = workloads.golden_scene(name, w, h, spp, rng, sky_dir=scratch)
        sc = rt.Scene(sf)
        for F in (1, 4, 16):
            buf = torch.zeros((F, h, w, 3), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            for _ in range(2):
                sc.render_frames_device(p, F, buf.data_ptr(), stream=st.cuda_stream)
            torch.cuda.synchronize()
            n = max(2, 32 // F)
            t = time.perf_counter()
            for _ in range(n):
                sc.render_frames_device(p, F, buf.data_ptr(), stream=st.cuda_stream)
            torch.cuda.synchronize()
            print("%-22s %2d frames per call: %.3f ms per frame" % (name, F, (time.perf_counter() - t) * 1e3 / (n * F)), flush=True)
        sc.close()
