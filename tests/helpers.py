"""Shared helpers of the parity tests: load a golden case (scene text, config, reference-style flags) into the
objects of raytracer_amd, exactly the way the reference's main() would interpret them (Code/main.cpp:82-226)."""
import json
import os

import numpy as np

import raytracer_amd as rt

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLD, "manifest.json")) as _f:
    MANIFEST = json.load(_f)
CASES = sorted(k for k, v in MANIFEST.items() if v["rng"] == "d0")          # deterministic: 8-bit max|diff| <= 1
STOCHASTIC = sorted(k for k, v in MANIFEST.items() if v["rng"] == "seeded")  # PSNR against a high-spp reference render


def flags_to_kwargs(flags):
    """Reference CLI flags -> (apply_flags kwargs, shutter_time override or None).  Unknown tokens are ignored, as
    main.cpp:219-226 does."""
    kw, shutter, i = {}, None, 0
    while i < len(flags):
        f = flags[i]
        if f == "--aa":
            kw["aa"] = int(flags[i + 1]); i += 1
        elif f == "--exposure":
            kw["exposure"] = float(flags[i + 1]); i += 1
        elif f == "--shadows":
            kw["shadows"] = 1
        elif f == "--glossy":
            kw["glossy"] = 1
        elif f == "--fresnel":
            kw["fresnel"] = 1
        elif f == "--normals":
            kw["normals"] = 1
        elif f == "--no-bvh":
            kw["no_bvh"] = 1
        elif f == "--tonemap":
            kw["tonemap"] = flags[i + 1]; i += 1
        elif f == "--motion-blur":
            shutter = float(flags[i + 1]); i += 1
        i += 1
    return kw, shutter


_SKY_ROOT = None


def sky_texture_root():
    """A texture root like tests/golden/scenes whose ../Textures also holds sky4k.pfm, the analytic 4096x2048 HDR environment
    of config 4 (100 MB: written by formula once per test session, never committed; the golden was rendered by the
    reference with the same file, oracle/gen_golden_r2.py)."""
    global _SKY_ROOT
    if _SKY_ROOT is None:
        import atexit
        import shutil
        import tempfile
        from raytracer_amd import workloads
        root = tempfile.mkdtemp(prefix="rt_sky_")
        atexit.register(shutil.rmtree, root, ignore_errors=True)
        os.makedirs(os.path.join(root, "scenes"))
        os.makedirs(os.path.join(root, "Textures"))
        tex = os.path.join(GOLD, "Textures")
        for f in os.listdir(tex):
            os.symlink(os.path.join(tex, f), os.path.join(root, "Textures", f))
        workloads.write_pfm_float32(os.path.join(root, "Textures", "sky4k.pfm"), workloads.analytic_sky())
        _SKY_ROOT = os.path.join(root, "scenes")
    return _SKY_ROOT


def load_case(name, rng="d0", **extra):
    """-> (SceneFile, RtRenderParams, golden image or None)"""
    with open(os.path.join(GOLD, "configs", name + ".json")) as f:
        cfg = rt.Config(text=f.read())
    kw, shutter = flags_to_kwargs(MANIFEST[name]["flags"])
    kw.update(extra)
    with open(os.path.join(GOLD, "scenes", name + ".txt"), newline="") as f:
        text = f.read()
    texture_root = sky_texture_root() if "../Textures/sky4k.pfm" in text else os.path.join(GOLD, "scenes")
    scene = rt.SceneFile(text=text, texture_root=texture_root, config=cfg, shutter_time=shutter)
    params = cfg.params(rng=rng, **kw)
    img_path = os.path.join(GOLD, "images", name + ".npz")
    golden = np.load(img_path)["image"] if os.path.exists(img_path) else None
    return scene, params, golden


def diff_stats(a, b):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    return int(d.max()), int((d > 0).sum()), int((d > 1).sum())


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)


def load_stochastic(name, spp, seed):
    """-> (SceneFile, params at `spp` samples with the seeded RNG, reference hi-spp image, reference lo-spp image,
    PSNR(lo, hi) of the reference's own two renders)"""
    scene, params, hi = load_case(name, rng="seeded", aa=spp, seed=seed)
    lo = np.load(os.path.join(GOLD, "images", name + ".npz"))["image_lo"]
    return scene, params, hi, lo, MANIFEST[name]["psnr_lo_vs_hi"]


def bias_bars(name):
    """Reference-internal noise figures of a stochastic case from its SIXTEEN independent 256-spp renders by the real
    reference (`hi_stack`, oracle/gen_golden_r3.py).  -> (hi = stack[0], mean of the sixteen as float, floor, halves):
      floor   PSNR of one 256-spp render against another, from the mean squared error over all 120 pairs (one pair alone
              scatters by +-1 dB where only a few penumbra pixels carry the noise: soft shadows 51.9 ... 54.0 dB)
      halves  PSNR of the mean of eight of the renders against the mean of the other eight."""
    stack = np.load(os.path.join(GOLD, "images", name + ".npz"))["hi_stack"].astype(np.float64)
    n = len(stack)
    pair_mse = np.mean([np.mean((stack[i] - stack[j]) ** 2) for i in range(n) for j in range(i + 1, n)])
    return stack[0].astype(np.uint8), stack.mean(axis=0), float(10.0 * np.log10(255.0 ** 2 / pair_mse)), psnr(stack[: n // 2].mean(axis=0), stack[n // 2:].mean(axis=0))


def check_bias(img, name, label=""):
    """Bar (1), the north-star bound at a sample count where it can see bias: a >= 1024-spp render `img` (uint8) must be no
    further from the reference's 256-spp render than the reference's own second 256-spp render is (floor), minus 0.5 dB.
    With 1024 samples the render's own noise is a quarter of the reference's 256-spp variance v, so noise alone leaves it
    ~2 dB ABOVE the floor (mse 1.25 v against 2 v).  Returns the failure message or None."""
    hi, _, floor, _ = bias_bars(name)
    p_hi = psnr(img, hi)
    print("bias %-16s %-18s PSNR vs ref_256 %.2f (floor %.2f)" % (name, label, p_hi, floor))
    return None if p_hi >= floor - 0.5 else "%s: %.2f dB against the reference's 256-spp floor %.2f dB" % (name, p_hi, floor)


def check_bias16(images, name, label=""):
    """Bar (2), the sharp one.  `images`: sixteen independent 256-spp byte images of the device under test.  Their mean is
    compared with the mean of the reference's sixteen: the same estimator on both sides (4096 samples per pixel, each render
    truncated to bytes once, as the reference does), so truncation acts alike and the difference of the two means is noise
    of 2 x (v / 4096 + truncation noise / 16) plus bias^2.  Two such means reach PSNR(mean of 8, mean of the other 8) + 3 dB
    by noise alone; the bar is 1 dB below that.  A sampler parameter that is 10 % off fails it (tests/test_stochastic_cpu.py)."""
    assert len(images) == 16
    _, mean16, _, halves = bias_bars(name)
    got = psnr(np.mean(np.stack(images).astype(np.float64), axis=0), mean16)
    print("bias16 %-16s %-18s PSNR of the two means of 16: %.2f (noise alone %.2f, bar %.2f)" % (name, label, got, halves + 3.0, halves + 2.0))
    return None if got >= halves + 2.0 else "%s: means of 16 x 256 spp differ: %.2f dB, noise alone gives %.2f dB" % (name, got, halves + 3.0)
