"""Secondary measurements (not the driver's bench line): the other BASELINE configs at full resolution on one GPU.
Scenes are the committed golden fixtures with only their `resolution` line changed.  Prints one JSON line each."""
import json, os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import helpers
import raytracer_amd as rt

CASES = [  # label, golden case, width, height, spp, rng
    ("C1 default scene, Whitted 1 spp", "c1_default", 1920, 1080, 1, "d0"),
    ("C1 + --shadows --fresnel", "c1_shadows_fresnel", 1920, 1080, 1, "d0"),
    ("C3 dof: soft shadows + glossy + DoF, 64 spp", "st_dof", 1920, 1080, 64, "seeded"),
    ("C4: config 4 as specified (c4_combined)", "c4_combined", 1920, 1080, 1, "d0"),
    ("C4-like: displaced shapes only", "own_complex", 1920, 1080, 1, "d0"),
    ("C4-like: glass + HDR environment", "own_glass_env", 1920, 1080, 1, "d0"),
    ("C5-like: motion blur + area lights, 4K 16 spp", "st_motion_blur", 3840, 2160, 16, "seeded"),
    ("gallery: 3900 quads", "gal_shapes", 1920, 1080, 1, "seeded"),
    ("gallery: 2647 quads", "gal_bike", 1920, 1080, 1, "seeded"),
]

import tempfile
from raytracer_amd import workloads
_SCRATCH = tempfile.mkdtemp(prefix="rt_cfg_")

def load(name, w, h, spp, rng):
    return workloads.golden_scene(name, w, h, spp, rng, sky_dir=_SCRATCH)

def main():
    rt.init(0)
    only = sys.argv[1:]
    for label, name, w, h, spp, rng in CASES:
        if only and name not in only: continue
        spp = int(os.environ.get("RT_BENCH_SPP", spp)) if spp > 1 else spp
        sf, params = load(name, w, h, spp, rng)
        if os.environ.get("RT_BENCH_SCHEDULE") == "mega":
            params.schedule = rt.RT_SCHEDULE_MEGAKERNEL
        scene = rt.Scene(sf)
        import torch
        dev = torch.device("cuda", 0)
        stream = torch.cuda.Stream(dev)
        buf = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
        img, st = scene.render(params, stats=True)   # counting pass (the statistics variant of the kernels): rays per frame
        torch.cuda.synchronize(dev)
        for _ in range(2):                           # warm-up on the timed stream (allocates its queues)
            scene.render_device(params, buf.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        reps = int(os.environ.get("RT_BENCH_REPS", "3"))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            e0.record(stream)
            for _ in range(reps):
                scene.render_device(params, buf.data_ptr(), stream=stream.cuda_stream)
            e1.record(stream)
        torch.cuda.synchronize(dev)
        wall = (time.perf_counter() - t0) * 1e3 / reps
        ms = e0.elapsed_time(e1) / reps
        print(json.dumps({"config": label, "case": name, "resolution": [w, h], "spp": spp, "rays": int(st.rays), "rays_traced": int(st.rays) - int(st.rays_coincident),
                          "kernel_ms": round(ms, 3), "traverse_ms": round(st.traverse_ms, 3), "wall_ms": round(wall, 3),
                          "Mrays_per_s": round(st.rays / ms / 1e3, 1), "mean_pixel": round(float(img.mean()), 2)}), flush=True)
        scene.close()

if __name__ == "__main__":
    try:
        main()
    finally:
        import shutil
        shutil.rmtree(_SCRATCH, ignore_errors=True)
