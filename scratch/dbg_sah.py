import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import raytracer_amd as rt
from test_bvh_device_image import _random_scene
rt.init(0)
def key(nodes):
    out = []
    for nd in nodes:
        leaves = tuple(sorted(int(c) for c in nd["child"] if c < 0))
        out.append((tuple(nd["org"].tolist()), int(nd["exps"]), tuple(int(x) for x in nd["qlo"]), tuple(int(x) for x in nd["qhi"]), leaves))
    return out
for n in [64, 65, 66, 80, 100, 128, 129, 130, 200, 256, 257, 300, 512, 513, 700, 1000, 5000]:
    sb = _random_scene(n, 11 + n, False)
    rt.set_bvh_builder("sah_gpu"); sc = rt.Scene(sb); img = sc.bvh_nodes(); sc.close(); rt.set_bvh_builder("default")
    host = rt.bvh_device_image(sb)
    a, b = set(k[4] for k in key(img["nodes"])), set(k[4] for k in key(host["nodes"]))
    print(n, img["info"].n_nodes, host["info"].n_nodes, img["info"].max_depth, host["info"].max_depth, "leaf-sets only in gpu:", len(a - b), "only in host:", len(b - a), flush=True)
