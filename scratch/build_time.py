"""Scene creation for the C2 frame (10^6 ellipsoids) with each builder: phases (RT_BUILD_TIMING) and the render cost of the result."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_BUILD_TIMING"] = "1"
import numpy as np
import raytracer_amd as rt
from raytracer_amd import workloads
rt.init(0)
sd = workloads.bvh_stress_scene(int(os.environ.get("SPHERES", "1000000")))
params = workloads.default_config().params(rng="d0")
frames = {}
for builder in os.environ.get("BUILDERS", "sah,lbvh,lbvh,sah").split(","):
    rt.set_bvh_builder(builder)
    t0 = time.perf_counter(); scene = rt.Scene(sd); t1 = time.perf_counter()
    img = scene.bvh_nodes()
    f, st = scene.render(params, stats=True)
    ts = []
    for _ in range(5):
        t2 = time.perf_counter(); scene.render(params); ts.append(time.perf_counter() - t2)
    print("%s: rt_scene_create %.1f ms (device build %.2f ms), %d nodes, depth %d; rt_render (host buffer) %.2f ms, kernels %.2f ms, traversal %.2f ms; nodes fetched/ray %.1f, box tests/ray %.1f, prim tests/ray %.2f"
          % (img["builder"], (t1 - t0) * 1e3, img["device_build_ms"], img["info"].n_nodes, img["info"].max_depth, min(ts) * 1e3,
             st.kernel_ms, st.traverse_ms, st.nodes_fetched / max(1, st.rays), st.nodes_visited / max(1, st.rays), st.prims_tested / max(1, st.rays)), flush=True)
    frames.setdefault(builder, f)
    scene.close()
ks = list(frames)
for k in ks[1:]:
    print("frames %s == %s: %s" % (ks[0], k, np.array_equal(frames[ks[0]], frames[k])))
