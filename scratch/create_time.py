import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import workloads
rt.init(0)
t0 = time.perf_counter(); sd = workloads.bvh_stress_scene(); t1 = time.perf_counter()
scene = rt.Scene(sd); t2 = time.perf_counter()
info = scene.bvh_info()
print("scene description %.2f s, rt_scene_create (2 host BVH builds + collapse + upload) %.2f s, %d nodes, depth %d" % (t1 - t0, t2 - t1, info.n_nodes, info.max_depth))
