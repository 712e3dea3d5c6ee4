"""TEST INFRASTRUCTURE ONLY: generates the committed fixtures under tests/golden/.

For every case below it renders the scene with the REAL reference renderer (oracle/_ref/ref_raytracer = the
unmodified sources under /root/reference/Code compiled with oracle/ref_shim.h, see oracle/Makefile) and stores
  tests/golden/scenes/<case>.txt      the exact scene text that was rendered (resolution line rewritten, texture
                                       paths pointed at the small synthetic textures of tests/golden/Textures/)
  tests/golden/configs/<case>.json    the config.json text that was used
  tests/golden/images/<case>.npz      the reference's output image (uint8 HxWx3)
  tests/golden/manifest.json          flags / rng mode per case
Fixtures are data (inputs + expected outputs); no reference source is stored.  Run here (needs /root/reference):
    make -C oracle ref && python oracle/gen_golden.py
"""
import json
import os
import re
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import run_ref  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
TEX = os.path.join(GOLD, "Textures")
DEFAULT_CFG = os.path.join(REF, "Code", "config.json")
EX = os.path.join(REF, "ASCII", "examples")


# ------------------------------------------------------------------ synthetic textures (this repo's own data)
def write_ppm(path, img):
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img, dtype=np.uint8).tobytes())


def write_ppm_ascii(path, img):
    with open(path, "w") as f:
        f.write("P3\n# ascii variant\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        for row in img:
            f.write(" ".join(str(int(v)) for v in row.reshape(-1)) + "\n")


def write_pfm(path, img, big_endian=False, scale=1.0):
    h, w, _ = img.shape
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n%s\n" % (w, h, (b"%g" % scale) if big_endian else (b"-%g" % scale)))
        f.write(np.ascontiguousarray(img, dtype=">f4" if big_endian else "<f4").tobytes())


def make_textures():
    os.makedirs(TEX, exist_ok=True)
    yy, xx = np.mgrid[0:48, 0:64]
    checker = (((xx // 8) + (yy // 8)) % 2).astype(np.float64)
    grad = xx / 63.0
    img = np.stack([40 + 200 * checker * grad, 60 + 150 * (1 - checker), 30 + 220 * (yy / 47.0)], axis=-1)
    write_ppm(os.path.join(TEX, "checker_grad.ppm"), img.astype(np.uint8))
    # smooth bumps for normal / displacement mapping
    yy, xx = np.mgrid[0:60, 0:80]
    hgt = 0.5 + 0.5 * np.sin(xx * 0.45) * np.cos(yy * 0.37)
    bump = np.stack([255 * hgt, 255 * hgt * 0.9, 255 * (1 - 0.5 * hgt)], axis=-1)
    write_ppm(os.path.join(TEX, "bumps.ppm"), bump.astype(np.uint8))
    # cube atlas (4x3 cross) with distinct colours per cell, written as ASCII P3 to cover that reader
    atlas = np.zeros((36, 48, 3), dtype=np.uint8)
    for cy in range(3):
        for cx in range(4):
            atlas[cy * 12:(cy + 1) * 12, cx * 12:(cx + 1) * 12] = [40 + 60 * cx, 50 + 90 * cy, 250 - 50 * cx - 30 * cy]
    atlas[::4, :, 0] = 255
    write_ppm_ascii(os.path.join(TEX, "atlas.ppm"), atlas)
    # analytic sky: gradient + sun disc, little-endian PFM; plus a big-endian scaled copy
    h, w = 32, 64
    v, u = np.mgrid[0:h, 0:w]
    sky = np.stack([0.3 + 0.5 * v / (h - 1), 0.4 + 0.4 * v / (h - 1), 0.9 - 0.3 * u / (w - 1)], axis=-1)
    sun = np.exp(-(((u - 40) / 3.0) ** 2 + ((v - 22) / 3.0) ** 2))
    sky = sky + 6.0 * sun[..., None]
    write_pfm(os.path.join(TEX, "sky.pfm"), sky.astype(np.float32))
    write_pfm(os.path.join(TEX, "sky_be.pfm"), (sky * 0.5).astype(np.float32), big_endian=True, scale=2.0)


# ------------------------------------------------------------------ this repo's own scenes (inputs we wrote)
CAM = """CAMERA
  location 7.358891 -6.925791 4.958309
  gaze_direction -0.651558 0.614170 -0.445271
  up_vector -0.324013 0.305421 0.895396
  focal_length 50.000000
  sensor_size 36.000000 24.000000
  resolution 160 90
  f_stop %s
  focal_distance %s
END_CAMERA
"""
LIGHTS = """POINT_LIGHT
  location 6.767676 -3.678335 5.903862
  intensity 1000.0 1000.0 1000.0
  radius %s
END_POINT_LIGHT
POINT_LIGHT
  location -4.0 -6.0 7.0
  intensity 300.0 500.0 700.0
  radius 0.0
END_POINT_LIGHT
"""
FLOOR = """PLANE
  corner -12.0 -12.0 -1.0
  corner 12.0 -12.0 -1.0
  corner -12.0 12.0 -1.0
  corner 12.0 12.0 -1.0
  ambient 0.25 0.35 0.35
  diffuse 0.9 0.9 0.9
  specular 0.3 0.3 0.3
  shininess 50.0
  reflectivity 0.2
%s
END_PLANE
"""

OWN_SCENES = {
    # textures on all three analytic shapes + bump maps (nearest for sphere/cube, bilinear for plane)
    "own_textured": CAM % ("99999.0", "10.0") + LIGHTS % "0.0" + FLOOR % "  texture_file ../Textures/checker_grad.ppm\n  bump_map_file ../Textures/bumps.ppm" + """
SPHERE
  translation -1.0 1.0 0.4
  rotation_euler_radians 0.3 -0.2 0.9
  scale 1.4 1.1 1.2
  diffuse 1.0 1.0 1.0
  shininess 20.0
  texture_file ../Textures/checker_grad.ppm
  bump_map_file ../Textures/bumps.ppm
END_SPHERE
CUBE
  translation 1.6 -1.2 0.0
  rotation_euler_radians 0.1 0.2 -0.6
  scale 0.9 0.8 1.0
  diffuse 1.0 1.0 1.0
  texture_file ../Textures/atlas.ppm
  bump_map_file ../Textures/bumps.ppm
  reflectivity 0.1
END_CUBE
""",
    # the three displaced (ray-marched) shapes
    "own_complex": CAM % ("99999.0", "10.0") + LIGHTS % "0.0" + """
COMPLEX_PLANE
  translation -1.0 1.0 -0.8
  rotation_euler_radians 0.0 0.0 -0.66
  scale 7.0 9.0 4.0
  ambient 0.3 0.3 0.3
  diffuse 0.7 0.8 0.5
  specular 0.3 0.3 0.3
  bump_map_file ../Textures/bumps.ppm
END_COMPLEX_PLANE
COMPLEX_SPHERE
  translation -1.2 1.0 0.5
  rotation_euler_radians 0.2 0.1 0.4
  scale 1.3 1.3 1.1
  diffuse 0.9 0.5 0.4
  bump_map_file ../Textures/bumps.ppm
  texture_file ../Textures/checker_grad.ppm
END_COMPLEX_SPHERE
COMPLEX_CUBE
  translation 1.5 -1.0 0.1
  rotation_euler_radians 0.0 0.0 -0.5
  scale 0.9 0.9 0.9
  diffuse 0.4 0.5 0.9
  bump_map_file ../Textures/bumps.ppm
  reflectivity 0.2
END_COMPLEX_CUBE
""",
    # glass + metal + environment map + total internal reflection candidates
    "own_glass_env": "HDR_BACKGROUND ../Textures/sky.pfm\n" + CAM % ("99999.0", "10.0") + LIGHTS % "0.0" + FLOOR % "" + """
SPHERE
  translation -0.5 0.5 0.3
  scale 1.3 1.3 1.3
  diffuse 0.9 1.0 0.9
  shininess 80.0
  transparency 0.9
  refractive_index 1.5
  material glass
END_SPHERE
CUBE
  translation 2.0 -1.5 -0.15
  rotation_euler_radians 0.0 0.0 0.4
  scale 0.8 0.8 0.8
  diffuse 0.8 0.9 1.0
  transparency 0.8
  refractive_index 1.33
  reflectivity 0.1
END_CUBE
SPHERE
  translation -3.0 -1.5 0.0
  scale 1.0 1.0 1.0
  diffuse 0.9 0.6 0.2
  reflectivity 0.7
  shininess 200.0
  material metal
END_SPHERE
""",
    "own_env_be": "HDR_BACKGROUND ../Textures/sky_be.pfm\n" + CAM % ("99999.0", "10.0") + LIGHTS % "0.0" + """
SPHERE
  translation 0.0 0.0 0.0
  scale 1.5 1.5 1.5
  reflectivity 0.9
  shininess 100.0
END_SPHERE
""",
    # moving shapes, area light, real aperture: the deterministic projections of motion blur / soft shadows / DoF
    "own_motion": CAM % ("2.0", "9.0") + LIGHTS % "0.6" + FLOOR % "  velocity 0.0 0.0 0.2" + """
SPHERE
  translation -1.0 1.0 0.4
  scale 1.2 1.2 1.2
  diffuse 0.2 0.3 0.9
  reflectivity 0.4
  velocity 1.5 0.8 0.0
  material metal
END_SPHERE
CUBE
  translation 1.6 -1.2 0.0
  rotation_euler_radians 0.0 0.0 0.3
  scale 0.8 0.8 1.0
  diffuse 0.9 0.3 0.2
  velocity -0.5 1.0 0.3
END_CUBE
COMPLEX_SPHERE
  translation -3.5 -2.0 0.0
  scale 0.9 0.9 0.9
  velocity 0.0 1.0 0.0
  bump_map_file ../Textures/bumps.ppm
END_COMPLEX_SPHERE
""",
    # parser edge cases: comments, unknown keys, missing texture file, plane with 3 corners (skipped), shininess 0
    # (NaN specular -> 255), two cameras (last wins), shape closed by another block's END token
    "own_parser_edges": "# a comment line\n\n" + CAM % ("99999.0", "10.0") + LIGHTS % "0.0" + FLOOR % "  texture_file ../Textures/does_not_exist.ppm\n  some_unknown_key 1 2 3" + """
PLANE
  corner 0 0 0
  corner 1 0 0
  corner 0 1 0
END_PLANE
SPHERE
  translation 0.0 0.0 0.5
  scale 1.5 1.5 1.5
  shininess 0.0
  diffuse 0.2 0.9 0.3
END_SPHERE
CUBE
  translation -3.0 2.5 0.0
  scale 0.7 0.7 0.7
  diffuse 0.9 0.1 0.1
END_SPHERE
CAMERA
  location 7.0 -7.0 5.2
END_CAMERA
""",
}

SKY_SCENES = ("own_glass_env", "own_env_be")


def ref_scene(path):
    with open(path, "r", newline="") as f:
        return f.read()


def cases():
    c = []

    def add(name, scene, config=DEFAULT_CFG, flags=(), res=(160, 90), own=False):
        c.append(dict(name=name, scene=scene, config=config, flags=list(flags), res=res, own=own))

    S = os.path.join(REF, "ASCII", "scene.txt")
    add("c1_default", S)
    add("c1_shadows_fresnel", S, flags=["--shadows", "--fresnel"])
    add("c1_normals", S, flags=["--normals"])
    add("c1_nobvh", S, flags=["--no-bvh", "--shadows"])
    add("c1_default_320", S, res=(320, 180))
    for n in (1, 2, 3, 10, 100):
        add("bvh_%d_normals" % n, os.path.join(REF, "ASCII", "BVH_tests", "scene_%d.txt" % n), flags=["--normals"], res=(192, 188))
    add("bvh_100_normals_nobvh", os.path.join(REF, "ASCII", "BVH_tests", "scene_100.txt"), flags=["--normals", "--no-bvh"], res=(192, 188))
    add("bvh_100_shaded", os.path.join(REF, "ASCII", "BVH_tests", "scene_100.txt"), flags=["--shadows"], res=(192, 188))
    add("m2_white", os.path.join(EX, "M2", "scene_white.txt"), os.path.join(EX, "M2", "config_white.json"), ["--exposure", "0.6"])
    add("m2_green", os.path.join(EX, "M2", "scene_green.txt"), os.path.join(EX, "M2", "config_green.json"), ["--exposure", "0.6"])
    add("m3_basic", os.path.join(EX, "M3", "basic", "basic.txt"), os.path.join(EX, "M3", "basic", "config.json"), ["--exposure", "0.04"])
    add("m3_refl_refr", os.path.join(EX, "M3", "refraction_reflection", "reflection_refraction.txt"),
        os.path.join(EX, "M3", "refraction_reflection", "config.json"), ["--exposure", "0.06"])
    add("m3_refl_refr_320", os.path.join(EX, "M3", "refraction_reflection", "reflection_refraction.txt"),
        os.path.join(EX, "M3", "refraction_reflection", "config.json"), ["--exposure", "0.06", "--fresnel", "--shadows"], res=(320, 180))
    for n in ("fresnel_cube", "fresnel_sphere"):
        add("m3_" + n, os.path.join(EX, "M3", "fresnel", n + ".txt"), os.path.join(EX, "M3", "fresnel", "config.json"),
            ["--exposure", "0.04", "--fresnel", "--shadows"])
    add("m3_anti_aliasing", os.path.join(EX, "M3", "anti_aliasing", "anti_aliasing.txt"),
        os.path.join(EX, "M3", "anti_aliasing", "config.json"), ["--exposure", "0.04", "--aa", "4"])
    for n in ("metal", "glass"):
        add("ex_" + n, os.path.join(EX, "exceptionality", "metal", n + ".txt"), os.path.join(EX, "exceptionality", "metal", "config.json"),
            ["--exposure", "0.04", "--fresnel", "--shadows"])
    for tm in ("reinhard", "aces", "filmic"):
        add("ex_tonemap_" + tm, os.path.join(EX, "exceptionality", "tonemapping", "scene_tonemapping.txt"), DEFAULT_CFG,
            ["--shadows", "--tonemap", tm])
    add("ex_exposure_high", os.path.join(EX, "exceptionality", "exposure", "exposure.txt"), DEFAULT_CFG, ["--exposure", "0.16", "--shadows"])
    add("ex_backgrounds", os.path.join(EX, "exceptionality", "backgrounds", "background.txt"),
        os.path.join(EX, "exceptionality", "backgrounds", "config.json"), ["--exposure", "0.02", "--shadows", "--fresnel"])
    add("ex_normal_mapping_1", os.path.join(EX, "exceptionality", "normal_mapping", "normal_mapping_1.txt"),
        os.path.join(EX, "exceptionality", "normal_mapping", "config.json"), ["--exposure", "0.04", "--shadows"])
    add("ex_normal_mapping_2", os.path.join(EX, "exceptionality", "normal_mapping", "normal_mapping_2.txt"),
        os.path.join(EX, "exceptionality", "normal_mapping", "config.json"), ["--exposure", "0.08", "--shadows"])
    add("ex_displacement", os.path.join(EX, "exceptionality", "displacement_mapping", "displacement_mapping.txt"), DEFAULT_CFG,
        ["--exposure", "0.04", "--shadows", "--tonemap", "aces"])
    add("fin_glossy_off", os.path.join(EX, "final", "glossy", "glossy.txt"), os.path.join(EX, "final", "glossy", "config.json"), ["--exposure", "0.04"])
    add("fin_glossy_on", os.path.join(EX, "final", "glossy", "glossy.txt"), os.path.join(EX, "final", "glossy", "config.json"), ["--exposure", "0.04", "--glossy"])
    add("fin_dof", os.path.join(EX, "final", "dof", "dof.txt"), os.path.join(EX, "final", "dof", "config.json"),
        ["--exposure", "0.04", "--fresnel", "--shadows", "--glossy"])
    add("fin_no_dof", os.path.join(EX, "final", "dof", "no_dof.txt"), os.path.join(EX, "final", "dof", "config.json"), ["--exposure", "0.04", "--fresnel", "--shadows"])
    add("fin_motion_blur", os.path.join(EX, "final", "motion_blur", "motion_blur.txt"), os.path.join(EX, "final", "motion_blur", "config.json"),
        ["--exposure", "0.2", "--shadows", "--motion-blur", "0.5"])
    add("fin_soft_shadows", os.path.join(EX, "final", "soft_shadows", "soft_shadows.txt"), os.path.join(EX, "final", "soft_shadows", "config.json"),
        ["--exposure", "0.04", "--shadows"])
    add("gal_shapes", os.path.join(EX, "gallery", "shapes", "shapes.txt"), os.path.join(EX, "gallery", "shapes", "config.json"), ["--shadows"], res=(128, 72))
    add("gal_bike", os.path.join(EX, "gallery", "bike", "bike.txt"), DEFAULT_CFG, ["--shadows", "--fresnel"], res=(128, 72))
    for name in OWN_SCENES:
        fl = ["--shadows", "--fresnel", "--exposure", "0.04"]
        if name == "own_complex":
            fl = ["--shadows", "--exposure", "0.04", "--tonemap", "filmic"]
        if name == "own_motion":
            fl = ["--shadows", "--exposure", "0.04", "--motion-blur", "0.8", "--glossy"]
        add(name, name, DEFAULT_CFG, fl, own=True)
    return c


# textures of the reference tree that are referenced by its example scenes -> this repo's small stand-ins
TEXTURE_MAP = {
    "checker_board_gradient.ppm": "bumps.ppm",
}


def prepare_scene_text(case):
    text = OWN_SCENES[case["scene"]] if case["own"] else ref_scene(case["scene"])
    text = run_ref.override_resolution(text, *case["res"])
    for src, dst in TEXTURE_MAP.items():
        text = text.replace("../Textures/" + src, "../Textures/" + dst)
    return text


# Stochastic cases (north_star: "a PSNR bound against a high-spp reference for stochastic features"): the reference
# itself, seeded RNG shim, single thread.  Stored: a high-spp image and an independent low-spp image whose PSNR against
# the high-spp one is the bar an equally sampled GPU render has to reach (minus 0.5 dB).
def stochastic_cases():
    c = []

    def add(name, scene, config, flags, res=(96, 54), hi=256, lo=16):
        c.append(dict(name=name, scene=scene, config=config, flags=list(flags), res=res, own=False, hi=hi, lo=lo))

    add("st_dof", os.path.join(EX, "final", "dof", "dof.txt"), os.path.join(EX, "final", "dof", "config.json"),
        ["--exposure", "0.04", "--fresnel", "--shadows", "--glossy"])
    add("st_soft_shadows", os.path.join(EX, "final", "soft_shadows", "soft_shadows.txt"), os.path.join(EX, "final", "soft_shadows", "config.json"),
        ["--exposure", "0.04", "--shadows"])
    add("st_motion_blur", os.path.join(EX, "final", "motion_blur", "motion_blur.txt"), os.path.join(EX, "final", "motion_blur", "config.json"),
        ["--exposure", "0.2", "--shadows", "--motion-blur", "0.5"])
    add("st_glossy", os.path.join(EX, "final", "glossy", "glossy.txt"), os.path.join(EX, "final", "glossy", "config.json"),
        ["--exposure", "0.04", "--glossy"])
    return c


def main():
    make_textures()
    for d in ("scenes", "configs", "images"):
        os.makedirs(os.path.join(GOLD, d), exist_ok=True)
    manifest = {}
    for case in cases():
        text = prepare_scene_text(case)
        with open(case["config"]) as f:
            cfg = f.read()
        img, out = run_ref.run_reference(text, cfg, flags=case["flags"] + ["--parallel"], rng="d0", texture_dirs=[TEX])
        name = case["name"]
        with open(os.path.join(GOLD, "scenes", name + ".txt"), "w", newline="") as f:
            f.write(text)
        with open(os.path.join(GOLD, "configs", name + ".json"), "w") as f:
            f.write(cfg)
        np.savez_compressed(os.path.join(GOLD, "images", name + ".npz"), image=img)
        manifest[name] = dict(flags=case["flags"], rng="d0", width=int(img.shape[1]), height=int(img.shape[0]),
                              source=("own scene (oracle/gen_golden.py)" if case["own"] else os.path.relpath(case["scene"], REF)))
        print("%-26s %4dx%-4d mean %.2f" % (name, img.shape[1], img.shape[0], img.mean()))
    for case in stochastic_cases():
        text = prepare_scene_text(case)
        with open(case["config"]) as f:
            cfg = f.read()
        name = case["name"]
        hi, _ = run_ref.run_reference(text, cfg, flags=case["flags"] + ["--aa", str(case["hi"])], rng="seed:1", texture_dirs=[TEX], threads=1)
        lo, _ = run_ref.run_reference(text, cfg, flags=case["flags"] + ["--aa", str(case["lo"])], rng="seed:2", texture_dirs=[TEX], threads=1)
        with open(os.path.join(GOLD, "scenes", name + ".txt"), "w", newline="") as f:
            f.write(text)
        with open(os.path.join(GOLD, "configs", name + ".json"), "w") as f:
            f.write(cfg)
        np.savez_compressed(os.path.join(GOLD, "images", name + ".npz"), image=hi, image_lo=lo)
        mse = np.mean((hi.astype(np.float64) - lo.astype(np.float64)) ** 2)
        psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
        manifest[name] = dict(flags=case["flags"], rng="seeded", width=int(hi.shape[1]), height=int(hi.shape[0]), spp_hi=case["hi"],
                              spp_lo=case["lo"], psnr_lo_vs_hi=round(float(psnr), 3), source=os.path.relpath(case["scene"], REF))
        print("%-26s %4dx%-4d hi %d spp, lo %d spp: PSNR(lo,hi) = %.2f dB" % (name, hi.shape[1], hi.shape[0], case["hi"], case["lo"], psnr))
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
