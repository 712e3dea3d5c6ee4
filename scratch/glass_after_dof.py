"""st_dof once, closed; then own_glass_env several times in the same process (scratch: is the slow frame a one-off after the allocation?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc
import raytracer_amd as rt
rt.init(0)
def case(name): return [c for c in bc.CASES if c[1] == name][0]
sf, params = bc.load(*case("st_dof")[1:])
s = rt.Scene(sf); s.render(params); s.render(params); s.close()
sf, params = bc.load(*case("own_glass_env")[1:])
scene = rt.Scene(sf)
for i in range(6):
    t0 = time.perf_counter()
    img, st = scene.render(params, stats=True)
    print("render %d: kernel %.1f ms traverse %.1f ms wall %.1f ms" % (i, st.kernel_ms, st.traverse_ms, (time.perf_counter() - t0) * 1e3), flush=True)
