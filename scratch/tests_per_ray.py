"""scratch/tests_per_ray.py <cases...>: box tests and primitive tests per ray of a config's frame (counting pass)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import workloads
rt.init(0)
cfgs = {c[1]: c for c in workloads.BASELINE_CONFIGS}
tmp = tempfile.mkdtemp()
for name in sys.argv[1:]:
    label, _, w, h, spp, rng = cfgs[name]
    sf, p = workloads.golden_scene(name, w // 4, h // 4, min(spp, 4), rng, sky_dir=tmp)
    scene = rt.Scene(sf)
    img, st = scene.render(p, stats=True)
    print("%s: rays %d (traced %d), box tests per traced ray %.2f, primitive tests per traced ray %.2f" % (name, st.rays, st.rays - st.rays_coincident, st.nodes_visited / (st.rays - st.rays_coincident), st.prims_tested / (st.rays - st.rays_coincident)))
    scene.close()
