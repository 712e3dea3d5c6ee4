"""Renders one secondary config several times in one process and prints the kernel time of each (scratch: run-to-run spread)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc
import raytracer_amd as rt
rt.init(0)
name = sys.argv[1] if len(sys.argv) > 1 else "own_glass_env"
case = [c for c in bc.CASES if c[1] == name][0]
sf, params = bc.load(*case[1:])
scene = rt.Scene(sf)
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    t0 = time.perf_counter()
    img, st = scene.render(params, stats=True)
    print("render %d: kernel %.1f ms traverse %.1f ms wall %.1f ms rays %d mean %.3f" % (i, st.kernel_ms, st.traverse_ms, (time.perf_counter() - t0) * 1e3, st.rays, float(img.mean())), flush=True)
