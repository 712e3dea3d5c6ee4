"""scratch/psnr_margins.py: PSNR of the seeded renders of the stochastic fixtures against the reference's 256-spp images, both
schedules, seeds 3 and 4 (what tests/test_gpu_stochastic.py asserts), with the reference's own figure and the bar."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
import raytracer_amd as rt
rt.init(0)
for name in helpers.STOCHASTIC:
    spp = helpers.MANIFEST[name]["spp_lo"]
    for sched in (rt.RT_SCHEDULE_WAVEFRONT, rt.RT_SCHEDULE_MEGAKERNEL):
        scene_file, params, hi, lo, bar = helpers.load_stochastic(name, spp, seed=3)
        params.schedule = sched
        scene = rt.Scene(scene_file)
        vals = []
        for seed in (3, 4, 5, 6, 7, 8):
            params.seed = seed
            vals.append(helpers.psnr(scene.render(params), hi))
        scene.close()
        print("%-18s %s  reference's own %.2f  bar %.2f  ours %s" % (name, "wavefront" if sched == rt.RT_SCHEDULE_WAVEFRONT else "megakernel", bar, bar - 0.5, " ".join("%.2f" % v for v in vals)), flush=True)
