"""scratch/accumulate_timing.py: config 5's loop -- passes of 16 spp of the 4K motion-blur frame added up in HBM by rt_accumulate_device --
with chunks two at a time (default) and one at a time (RT_ONE_CHUNK_AT_A_TIME=1)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import workloads
cfgs = {c[1]: c for c in workloads.BASELINE_CONFIGS}
label, name, w, h, spp, rng = cfgs["st_motion_blur"]
rt.init(0)
dev = torch.device("cuda", 0)
with tempfile.TemporaryDirectory() as scratch:
    sf, p = workloads.golden_scene(name, w, h, spp, rng, sky_dir=scratch)
    scene = rt.Scene(sf)
    d_sum = torch.zeros((h, w, 3), dtype=torch.float64, device=dev)
    st = torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    scene.accumulate_device(p, d_sum.data_ptr(), 0, 2, stream=st.cuda_stream)
    torch.cuda.synchronize()
    n = 6
    t = time.perf_counter()
    scene.accumulate_device(p, d_sum.data_ptr(), 2, n, stream=st.cuda_stream)
    torch.cuda.synchronize()
    print("%d passes of %d spp at %dx%d: %.2f ms per pass (%s)" % (n, spp, w, h, (time.perf_counter() - t) * 1e3 / n,
          "chunks one at a time" if os.environ.get("RT_ONE_CHUNK_AT_A_TIME") else "chunks two at a time"))
    scene.close()
