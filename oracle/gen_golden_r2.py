"""TEST INFRASTRUCTURE ONLY: the fixtures added in round 2, rendered by the REAL reference (oracle/_ref/ref_raytracer) like
those of gen_golden.py, and merged into tests/golden/manifest.json:

  c4_combined           BASELINE config 4 as specified: COMPLEX_PLANE of the displacement example +
                        textured sphere + bump-mapped cube + metal sphere + refractive sphere + the analytic 4096x2048 HDR
                        environment, --shadows --fresnel --tonemap aces.  The 100 MB sky is NOT committed: both sides write
                        it by formula (raytracer_amd.workloads.analytic_sky).
  c1_1080p              config 1 at its own resolution (1920x1080)
  fin_dof_1080p, fin_motion_blur_1080p   the scenes of configs 3 and 5 at 1080p, deterministic sampling
  c2_100k_rows8.npz     every 8th row of the 1080p frame of config 2 with 10^5 ellipsoids (the largest the reference loads
                        in seconds): the GPU test no longer depends on the reference binary travelling to the GPU box.
Run here (needs /root/reference):  make -C oracle ref && python oracle/gen_golden_r2.py
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import gen_golden as g  # noqa: E402
import run_ref  # noqa: E402
from raytracer_amd import workloads  # noqa: E402

C4_SCENE = "HDR_BACKGROUND ../Textures/sky4k.pfm\n" + g.CAM % ("99999.0", "10.0") + g.LIGHTS % "0.0" + """
COMPLEX_PLANE
  translation -1.0 1.0 -0.8
  rotation_euler_radians 0.0 0.0 -0.66
  scale 7.0 9.0 4.0
  ambient 0.3 0.3 0.3
  diffuse 0.7 0.8 0.5
  specular 0.3 0.3 0.3
  bump_map_file ../Textures/bumps.ppm
END_COMPLEX_PLANE
SPHERE
  translation -1.6 1.4 0.45
  rotation_euler_radians 0.3 -0.2 0.9
  scale 1.2 1.0 1.1
  diffuse 1.0 1.0 1.0
  shininess 20.0
  texture_file ../Textures/checker_grad.ppm
END_SPHERE
CUBE
  translation 1.7 -1.3 0.1
  rotation_euler_radians 0.1 0.2 -0.6
  scale 0.8 0.8 0.9
  diffuse 0.9 0.9 1.0
  texture_file ../Textures/atlas.ppm
  bump_map_file ../Textures/bumps.ppm
  reflectivity 0.1
END_CUBE
SPHERE
  translation 0.6 1.9 0.35
  scale 1.0 1.0 1.0
  diffuse 0.9 0.6 0.2
  reflectivity 0.7
  shininess 200.0
  material metal
END_SPHERE
SPHERE
  translation 0.9 -0.1 0.6
  scale 0.9 0.9 0.9
  diffuse 0.9 1.0 0.9
  shininess 80.0
  transparency 0.9
  refractive_index 1.5
  material glass
END_SPHERE
"""


def main():
    with open(os.path.join(g.GOLD, "manifest.json")) as f:
        manifest = json.load(f)
    cfg_default = open(g.DEFAULT_CFG).read()

    def store(name, text, cfg, flags, img, source):
        with open(os.path.join(g.GOLD, "scenes", name + ".txt"), "w", newline="") as f:
            f.write(text)
        with open(os.path.join(g.GOLD, "configs", name + ".json"), "w") as f:
            f.write(cfg)
        np.savez_compressed(os.path.join(g.GOLD, "images", name + ".npz"), image=img)
        manifest[name] = dict(flags=list(flags), rng="d0", width=int(img.shape[1]), height=int(img.shape[0]), source=source)
        print("%-26s %4dx%-4d mean %.2f" % (name, img.shape[1], img.shape[0], img.mean()))

    # config 4 as specified: the sky goes into a scratch texture directory next to the committed small textures
    with tempfile.TemporaryDirectory(prefix="rt_sky_") as sky_dir:
        workloads.write_pfm_float32(os.path.join(sky_dir, "sky4k.pfm"), workloads.analytic_sky())
        text = run_ref.override_resolution(C4_SCENE, 240, 135)
        flags = ["--shadows", "--fresnel", "--exposure", "0.04", "--tonemap", "aces"]
        img, _ = run_ref.run_reference(text, cfg_default, flags=flags + ["--parallel"], rng="d0", texture_dirs=[g.TEX, sky_dir])
        store("c4_combined", text, cfg_default, flags, img, "own scene (oracle/gen_golden_r2.py) + analytic 4096x2048 sky (workloads.analytic_sky)")

    # full-size frames of the configs the reference can render here in seconds to minutes
    ex = g.EX
    for name, scene, cfg_path, flags in (
            ("c1_1080p", os.path.join(g.REF, "ASCII", "scene.txt"), g.DEFAULT_CFG, []),
            ("fin_dof_1080p", os.path.join(ex, "final", "dof", "dof.txt"), os.path.join(ex, "final", "dof", "config.json"),
             ["--exposure", "0.04", "--fresnel", "--shadows", "--glossy"]),
            ("fin_motion_blur_1080p", os.path.join(ex, "final", "motion_blur", "motion_blur.txt"), os.path.join(ex, "final", "motion_blur", "config.json"),
             ["--exposure", "0.2", "--shadows", "--motion-blur", "0.5"])):
        text = run_ref.override_resolution(g.ref_scene(scene), 1920, 1080)
        for src, dst in g.TEXTURE_MAP.items():
            text = text.replace("../Textures/" + src, "../Textures/" + dst)
        cfg = open(cfg_path).read()
        img, _ = run_ref.run_reference(text, cfg, flags=flags + ["--parallel"], rng="d0", texture_dirs=[g.TEX])
        store(name, text, cfg, flags, img, os.path.relpath(scene, g.REF) + " at 1920x1080")

    # config 2 with 10^5 ellipsoids: rows 0, 8, 16, ... of the reference's 1080p frame
    text = workloads.bvh_stress_scene_text(100_000, 1920, 1080)
    img, _ = run_ref.run_reference(text, cfg_default, flags=["--parallel"], rng="d0")
    np.savez_compressed(os.path.join(g.GOLD, "images", "c2_100k_rows8.npz"), rows=img[::8], row_step=8, n_spheres=100_000)
    print("c2_100k_rows8: %d rows, mean %.2f" % (img[::8].shape[0], img.mean()))

    with open(os.path.join(g.GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
