"""rtcli end to end on the GPU: reference flags + explicit paths -> PPM on disk identical to the golden image."""
import os
import subprocess

import numpy as np
import pytest

import helpers
import raytracer_amd as rt

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rtcli_renders_golden(tmp_path):
    name = "c1_shadows_fresnel"
    out = str(tmp_path / "o.ppm")
    cmd = [os.path.join(ROOT, "raytracer_amd", "rtcli"), "--scene", os.path.join(helpers.GOLD, "scenes", name + ".txt"),
           "--config", os.path.join(helpers.GOLD, "configs", name + ".json"), "--texture-root", os.path.join(helpers.GOLD, "scenes"),
           "--out", out, "--pfm", str(tmp_path / "o.pfm"), "--d0", "--bogus-flag", "7"] + helpers.MANIFEST[name]["flags"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert p.returncode == 0, p.stdout[-2000:]
    assert "Shadows enabled" in p.stdout and "Image saved to" in p.stdout
    img = rt.read_ppm(out)
    golden = np.load(os.path.join(helpers.GOLD, "images", name + ".npz"))["image"]
    mx, nd, n1 = helpers.diff_stats(img, golden)
    assert mx <= 1
    # --pfm: the same frame before tone mapping (float32, bottom row first); tone-mapped on the host it is the PPM again
    from raytracer_amd import progressive
    lin = rt.read_pfm(str(tmp_path / "o.pfm"))[::-1]
    mx, nd, n1 = helpers.diff_stats(progressive.to_image(lin, rt.RT_TONEMAP_NONE), img)
    assert mx <= 1 and nd <= 0.01 * img.size


def test_rtcli_bvh_builder_flag(tmp_path):
    """--bvh-builder lbvh: the hierarchy built on the GPU, the same picture."""
    name = "bvh_100_shaded"
    outs = []
    for builder in ("sah", "lbvh"):
        out = str(tmp_path / (builder + ".ppm"))
        cmd = [os.path.join(ROOT, "raytracer_amd", "rtcli"), "--scene", os.path.join(helpers.GOLD, "scenes", name + ".txt"),
               "--config", os.path.join(helpers.GOLD, "configs", name + ".json"), "--texture-root", os.path.join(helpers.GOLD, "scenes"),
               "--out", out, "--d0", "--bvh-builder", builder] + helpers.MANIFEST[name]["flags"]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
        assert p.returncode == 0, p.stdout[-2000:]
        outs.append(rt.read_ppm(out))
    golden = np.load(os.path.join(helpers.GOLD, "images", name + ".npz"))["image"]
    assert np.array_equal(outs[0], outs[1]) and helpers.diff_stats(outs[1], golden)[0] <= 1


def test_rtcli_missing_scene_fails(tmp_path):
    p = subprocess.run([os.path.join(ROOT, "raytracer_amd", "rtcli"), "--scene", "/nonexistent.txt", "--config", "/nonexistent.json"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert p.returncode == 1 and "Failed to open scene file" in p.stdout
