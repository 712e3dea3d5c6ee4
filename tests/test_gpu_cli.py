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


def _reference_tree(tmp_path, scenes):
    """The directory layout the reference's hard-coded relative paths expect (main.cpp:49,518,530):
    X/Code/build (cwd), X/Code/config.json, X/ASCII/..., X/Output/, X/Textures -- textures are looked up in "..".  """
    import shutil
    x = tmp_path / "X"
    (x / "Code" / "build").mkdir(parents=True)
    (x / "ASCII" / "BVH_tests").mkdir(parents=True)
    (x / "Output").mkdir()
    shutil.copy(os.path.join(helpers.GOLD, "configs", "c1_default.json"), x / "Code" / "config.json")
    for dst, name in scenes.items():
        shutil.copy(os.path.join(helpers.GOLD, "scenes", name + ".txt"), x / "ASCII" / dst)
    return x


def test_rtcli_time_flag_writes_the_references_timing_log(tmp_path):
    """--time 2 (main.cpp:486-507,546-561): Output/testing/<timestamp>/ with a copy of the scene, output_<ts>_<i>.ppm per run
    and timing_log.txt = "args: [<all args, blank after each>]" then one "[<seconds>, <image path>]" line per run."""
    import re
    x = _reference_tree(tmp_path, {"scene.txt": "c1_default"})
    exe = os.path.join(ROOT, "raytracer_amd", "rtcli")
    p = subprocess.run([exe, "--time", "2", "--d0", "--shadows"], cwd=x / "Code" / "build", stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=120)
    assert p.returncode == 0, p.stdout[-2000:]
    assert "Timing enabled: 2 runs." in p.stdout and "--- Starting Run 2 of 2 ---" in p.stdout and "Timing log saved to:" in p.stdout
    runs = sorted((x / "Output" / "testing").iterdir())
    assert len(runs) == 1 and re.fullmatch(r"\d{4}-\d\d-\d\d_\d\d-\d\d-\d\d", runs[0].name)
    ts = runs[0].name
    assert (runs[0] / "scene.txt").read_bytes() == (x / "ASCII" / "scene.txt").read_bytes()
    log = (runs[0] / "timing_log.txt").read_text().splitlines()
    assert log[0] == "args: [--time 2 --d0 --shadows ]"
    assert len(log) == 3
    imgs = []
    for i, line in enumerate(log[1:], 1):
        m = re.fullmatch(r"\[([0-9.e+-]+), (.*)\]", line)
        assert m and float(m.group(1)) > 0.0
        assert m.group(2) == "../../Output/testing/%s/output_%s_%d.ppm" % (ts, ts, i)
        imgs.append(rt.read_ppm(str(x / "Code" / "build" / m.group(2))))
    assert np.array_equal(imgs[0], imgs[1])
    scene_file, params, golden = helpers.load_case("c1_shadows_fresnel", fresnel=0)   # the same scene without --fresnel
    assert imgs[0].shape == golden.shape


def test_rtcli_bvh_testing_sweep(tmp_path):
    """--bvh_testing (main.cpp:362-479): every ASCII/BVH_tests/scene_<N>.txt three times with and three times without the BVH;
    "<average seconds> <N>" lines sorted by N in bvh_test.txt / no_bvh_test.txt, first image of each as bvh_N.ppm / no_bvh_N.ppm,
    the scene folder copied, flags.txt.  The reference's own invariant: both pictures of a scene are the same."""
    import re
    x = _reference_tree(tmp_path, {"BVH_tests/scene_100.txt": "bvh_100_normals", "BVH_tests/scene_10.txt": "bvh_10_normals",
                                   "BVH_tests/notes.md": "bvh_10_normals"})
    exe = os.path.join(ROOT, "raytracer_amd", "rtcli")
    p = subprocess.run([exe, "--normals", "--bvh_testing", "--d0"], cwd=x / "Code" / "build", stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:]
    assert "--- Testing Scene X=10 ---" in p.stdout and "Run 3/3 [BVH OFF]" in p.stdout and "BVH Testing Complete." in p.stdout
    out = next((x / "Output" / "testing").iterdir())
    assert (out / "flags.txt").read_text() == "Flags used: --normals --bvh_testing --d0 \n"
    assert sorted(f.name for f in (out / "BVH_tests").iterdir()) == ["notes.md", "scene_10.txt", "scene_100.txt"]
    for name in ("bvh_test.txt", "no_bvh_test.txt"):
        lines = (out / name).read_text().splitlines()
        assert len(lines) == 2
        vals = [re.fullmatch(r"([0-9.e+-]+) (\d+)", ln) for ln in lines]
        assert all(vals) and [int(v.group(2)) for v in vals] == [10, 100] and all(float(v.group(1)) > 0 for v in vals)
    for n, case in ((10, "bvh_10_normals"), (100, "bvh_100_normals")):
        a, b = rt.read_ppm(str(out / ("bvh_%d.ppm" % n))), rt.read_ppm(str(out / ("no_bvh_%d.ppm" % n)))
        assert np.array_equal(a, b)
        golden = np.load(os.path.join(helpers.GOLD, "images", case + ".npz"))["image"]
        assert helpers.diff_stats(a, golden)[0] <= 1


def test_rtcli_gpus_beyond_the_box_fails_cleanly(tmp_path):
    """--gpus 2 on a one-GPU box: rank 1 finds no device, nobody creates a communicator, nothing is written, exit 1 (no hang)."""
    if __import__("torch").cuda.device_count() >= 2:
        pytest.skip("needs a one-GPU box")
    name = "c1_default"
    out = str(tmp_path / "o.ppm")
    cmd = [os.path.join(ROOT, "raytracer_amd", "rtcli"), "--gpus", "2", "--scene", os.path.join(helpers.GOLD, "scenes", name + ".txt"),
           "--config", os.path.join(helpers.GOLD, "configs", name + ".json"), "--out", out, "--d0"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert p.returncode == 1 and "not every rank got its GPU" in p.stdout and not os.path.exists(out)
