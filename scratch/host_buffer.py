"""rt_render / rt_render_frames into HOST memory on the C2 frame: ms per frame for one frame per call (fresh buffer, reused buffer
with and without registration) and for F frames per call.  Usage: python scratch/host_buffer.py [spheres]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as rt
from raytracer_amd import workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rt.init(0)
cfg = workloads.default_config()
scene = rt.Scene(workloads.bvh_stress_scene(n, 1920, 1080))
p = cfg.params(rng="seeded", seed=12345)
buf = np.zeros((1080, 1920, 3), dtype=np.uint8)
for _ in range(3):
    scene.render(p, out=buf)
def timed(f, reps=20):
    f()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t) * 1e3 / reps
print("mode", os.environ.get("RT_HOST_REGISTER", "second (default)"))
print("rt_render, same buffer every call      %.3f ms per frame" % timed(lambda: scene.render(p, out=buf)))
print("rt_render, a new buffer every call     %.3f ms per frame" % timed(lambda: scene.render(p)))
for F in (2, 4, 8, 16, 64):
    out = np.zeros((F, 1080, 1920, 3), dtype=np.uint8)
    ms = timed(lambda: scene.render_frames(p, F, out=out), reps=max(2, 32 // F))
    print("rt_render_frames, %2d frames per call    %.3f ms per frame" % (F, ms / F))
one = scene.render(p)
many = scene.render_frames(p, 5)
pp = p.copy()
for f in range(5):
    pp.seed = p.seed + f
    assert np.array_equal(many[f], scene.render(pp)), f
print("frames of rt_render_frames == rt_render with seed + f")
