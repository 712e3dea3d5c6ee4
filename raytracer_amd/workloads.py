"""Synthetic workloads of BASELINE.json / SURVEY 8(d), built in memory (nothing is read from the reference tree)."""
import os

import numpy as np

import raytracer_amd as rt

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_CONFIG_PATH = os.path.join(_HERE, "default_config.json")  # same keys and values as the shipped Code/config.json


def default_config():
    return rt.Config(path=DEFAULT_CONFIG_PATH)


def bvh_stress_scene(n_spheres=1_000_000, width=1920, height=1080, seed=12345, shutter_time=0.5):
    """Config C2: camera and light of ASCII/BVH_tests/scene_3.txt:8-25 with `resolution` set to width x height and
    n_spheres SPHERE blocks with the default material: translation ~ U([-0.8,1.8] x [-0.8,1.6] x [-1.2,1.05]) (the
    extent of scene_100.txt), per-axis scale ~ U(0.5 r, 1.5 r) with r = 0.35 / cbrt(N), rotation about z ~ U(-pi, pi),
    numpy default_rng(seed)."""
    rng = np.random.default_rng(seed)
    n = int(n_spheres)
    trs = np.zeros((n, 9), dtype=np.float64)
    trs[:, 0] = rng.uniform(-0.8, 1.8, n)
    trs[:, 1] = rng.uniform(-0.8, 1.6, n)
    trs[:, 2] = rng.uniform(-1.2, 1.05, n)
    trs[:, 5] = rng.uniform(-np.pi, np.pi, n)
    r = 0.35 / np.cbrt(n)
    trs[:, 6:9] = rng.uniform(0.5 * r, 1.5 * r, (n, 3))
    shapes = rt.shapes_from_trs(trs, rt.RT_SHAPE_SPHERE, 0)
    cam = rt.RtCamera()
    cam.location[:] = [2.5, -2.5, 1.5]
    cam.gaze_direction[:] = [-0.506798, 0.760196, -0.406519]
    cam.up_vector[:] = [-0.225496, 0.338244, 0.913642]
    cam.focal_length, cam.sensor_width, cam.sensor_height = 50.0, 36.0, 24.0
    cam.resolution_x, cam.resolution_y = width, height
    cam.f_stop, cam.focal_distance = 99999.0, 10.0
    light = rt.RtLight()
    light.position[:] = [0.233330, 0.717368, 1.959181]
    light.intensity[:] = [10.0, 10.0, 10.0]
    light.radius = 0.0
    return rt.SceneBuilder(cam, [light], shapes, [rt.default_material()], shutter_time=shutter_time)


def bvh_stress_scene_text(n_spheres, width=1920, height=1080, seed=12345):
    """The same workload as scene-file text in the reference's grammar (scene.cpp:124-546; layout of
    ASCII/BVH_tests/scene_3.txt), 9 decimals.  For runs of the reference binary next to this library on one input: parse
    this text with rt.SceneFile so that both sides see the rounded numbers."""
    rng = np.random.default_rng(seed)
    n = int(n_spheres)
    tx, ty, tz = rng.uniform(-0.8, 1.8, n), rng.uniform(-0.8, 1.6, n), rng.uniform(-1.2, 1.05, n)
    rz = rng.uniform(-np.pi, np.pi, n)
    r = 0.35 / np.cbrt(n)
    sc = rng.uniform(0.5 * r, 1.5 * r, (n, 3))
    head = ("CAMERA\n  location 2.500000 -2.500000 1.500000\n  gaze_direction -0.506798 0.760196 -0.406519\n"
            "  up_vector -0.225496 0.338244 0.913642\n  focal_length 50.000000\n  sensor_size 36.000000 24.000000\n"
            "  resolution %d %d\n  f_stop 99999.0\n  focal_distance 10.0\nEND_CAMERA\n\n"
            "POINT_LIGHT\n  location 0.233330 0.717368 1.959181\n  intensity 10.000000 10.000000 10.000000\n  radius 0.000000\n"
            "END_POINT_LIGHT\n\n" % (width, height))
    blocks = ["SPHERE\n  translation %.9f %.9f %.9f\n  rotation_euler_radians 0.000000000 0.000000000 %.9f\n  scale %.9f %.9f %.9f\nEND_SPHERE\n"
              % (tx[i], ty[i], tz[i], rz[i], sc[i, 0], sc[i, 1], sc[i, 2]) for i in range(n)]
    return head + "\n".join(blocks)


def analytic_sky(width=4096, height=2048):
    """The HDR environment of BASELINE config 4 ("4k HDR environment"): the reference tree's .pfm / .hdr files are
    missing blobs, so the sky is written by formula, seed-free -- a vertical gradient, a warm horizon band, a sun disc of
    radiance 40 and a faint checker so that a wrong texel shows.  float32 [height, width, 3], row 0 = first row of the PFM
    payload (v = 0, direction -z, HDRImage.cpp:112-182)."""
    v = (np.arange(height, dtype=np.float64) / (height - 1))[:, None]
    u = (np.arange(width, dtype=np.float64) / (width - 1))[None, :]
    sky = np.empty((height, width, 3), dtype=np.float64)
    horizon = np.exp(-((v - 0.5) / 0.08) ** 2)
    sky[..., 0] = 0.15 + 0.55 * v + 0.5 * horizon + 0.0 * u
    sky[..., 1] = 0.25 + 0.50 * v + 0.3 * horizon + 0.0 * u
    sky[..., 2] = 0.95 - 0.35 * v + 0.1 * np.cos(2.0 * np.pi * u)
    sun = np.exp(-(((u - 0.62) / 0.012) ** 2 + ((v - 0.71) / 0.024) ** 2))
    sky += 40.0 * sun[..., None]
    checker = ((np.arange(height)[:, None] // 64 + np.arange(width)[None, :] // 64) % 2).astype(np.float64)
    sky *= (0.96 + 0.04 * checker)[..., None]
    return sky.astype(np.float32)


def write_pfm_float32(path, img):
    """Little-endian "PF" file of a float32 [H, W, 3] array, rows in array order (what HDRImage::load reads)."""
    h, w, _ = img.shape
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
        f.write(np.ascontiguousarray(img, dtype="<f4").tobytes())


# ---------------------------------------------------------------------------------------- the other BASELINE configs
# Scenes of configs 1, 3, 4 and 5 as bench.py and the profiling scripts render them: the committed fixtures under
# tests/golden/ (scene text + config text + reference-style flags; the reference tree itself is never read at run time)
# with only the `resolution` line and the sample count changed.
_REPO = os.path.dirname(_HERE)
GOLDEN_DIR = os.path.join(_REPO, "tests", "golden")
BASELINE_CONFIGS = [   # label, fixture, width, height, samples per pixel, sampling
    ("C1: ASCII/scene.txt, Whitted 1 spp", "c1_default", 1920, 1080, 1, "seeded"),
    ("C1 + --shadows --fresnel", "c1_shadows_fresnel", 1920, 1080, 1, "seeded"),
    ("C3: dof.txt, soft shadows + glossy + depth of field, 64 spp", "st_dof", 1920, 1080, 64, "seeded"),
    ("C4: displaced + textured + bump-mapped shapes, metal + glass, 4096x2048 HDR environment, ACES", "c4_combined", 1920, 1080, 1, "seeded"),
    ("C5 share: motion_blur.txt, 3840x2160, 16 of the 1024 spp (cost is linear in spp)", "st_motion_blur", 3840, 2160, 16, "seeded"),
]


def _flags_to_kwargs(flags):
    """Reference CLI flags (main.cpp:82-216) -> (Config.params keyword arguments, shutter time override or None)."""
    kw, shutter, i = {}, None, 0
    simple = {"--shadows": "shadows", "--glossy": "glossy", "--fresnel": "fresnel", "--normals": "normals", "--no-bvh": "no_bvh"}
    while i < len(flags):
        f = flags[i]
        if f in simple:
            kw[simple[f]] = 1
        elif f == "--aa":
            kw["aa"] = int(flags[i + 1]); i += 1
        elif f == "--exposure":
            kw["exposure"] = float(flags[i + 1]); i += 1
        elif f == "--tonemap":
            kw["tonemap"] = flags[i + 1]; i += 1
        elif f == "--motion-blur":
            shutter = float(flags[i + 1]); i += 1
        i += 1
    return kw, shutter


def golden_scene(name, width=None, height=None, spp=None, rng="d0", seed=7, sky_dir=None):
    """-> (SceneFile, RtRenderParams) of a committed fixture, optionally at another resolution / sample count.  A scene
    that names ../Textures/sky4k.pfm (config 4) gets the analytic sky written into `sky_dir` (a scratch directory the
    caller owns) next to links to the committed small textures."""
    import json
    import re
    with open(os.path.join(GOLDEN_DIR, "manifest.json")) as f:
        flags = json.load(f)[name]["flags"]
    with open(os.path.join(GOLDEN_DIR, "configs", name + ".json")) as f:
        cfg = rt.Config(text=f.read())
    kw, shutter = _flags_to_kwargs(flags)
    if spp is not None:
        kw["aa"] = spp
    if rng == "seeded":
        kw["seed"] = seed
    with open(os.path.join(GOLDEN_DIR, "scenes", name + ".txt"), newline="") as f:
        text = f.read()
    if width is not None:
        text, n = re.subn(r"(?im)^(\s*resolution\s+)\d+\s+\d+", lambda m: "%s%d %d" % (m.group(1), width, height), text)
        assert n == 1, name
    root = os.path.join(GOLDEN_DIR, "scenes")
    if "../Textures/sky4k.pfm" in text:
        assert sky_dir is not None, "this scene needs a scratch directory for the analytic sky"
        os.makedirs(os.path.join(sky_dir, "scenes"), exist_ok=True)
        os.makedirs(os.path.join(sky_dir, "Textures"), exist_ok=True)
        tex = os.path.join(GOLDEN_DIR, "Textures")
        for f in os.listdir(tex):
            if not os.path.lexists(os.path.join(sky_dir, "Textures", f)):
                os.symlink(os.path.join(tex, f), os.path.join(sky_dir, "Textures", f))
        sky = os.path.join(sky_dir, "Textures", "sky4k.pfm")
        if not os.path.exists(sky):
            write_pfm_float32(sky, analytic_sky())
        root = os.path.join(sky_dir, "scenes")
    scene = rt.SceneFile(text=text, texture_root=root, config=cfg, shutter_time=shutter)
    return scene, cfg.params(rng=rng, **kw)
