"""k_tail: the deep levels of a small scene's ray tree in one launch, chosen from the level sizes of the previous frame of the same
shape.  The first frame of a scene goes level by level; the following ones must be the same picture with the same ray count."""
import numpy as np
import pytest

import helpers
import raytracer_amd as rt

CASES = ["c1_default", "c1_shadows_fresnel", "c4_combined", "own_glass_env", "m3_refl_refr", "ex_glass", "own_textured", "fin_dof"]
CASES = [c for c in CASES if c in helpers.MANIFEST]


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_frames_after_the_first_end_in_one_launch_and_equal_it(gpu, name, capfd, monkeypatch):
    monkeypatch.setenv("RT_DEBUG_LEVELS", "1")
    scene_file, params, golden = helpers.load_case(name)
    scene = rt.Scene(scene_file)
    frames, rays = [], []
    for _ in range(3):
        img, st = scene.render(params, stats=True)
        frames.append(img)
        rays.append(st.rays)
    scene.close()
    err = capfd.readouterr().err
    small = scene_file.n_shapes <= 16 if hasattr(scene_file, "n_shapes") else None
    assert rays[0] == rays[1] == rays[2], rays
    for k, img in enumerate(frames):
        mx, nd, _ = helpers.diff_stats(img, golden)
        assert mx <= 1, (name, k, mx, nd)
    for k in (1, 2):   # the order of the FP64 additions may move a value across a truncation boundary
        mx, nd, _ = helpers.diff_stats(frames[k], frames[0])
        assert mx <= 1 and nd <= 0.002 * frames[0].size, (name, k, mx, nd)
    if name in ("c1_default", "c1_shadows_fresnel"):   # twelve levels of which ten are nearly empty: the tail must have been taken
        assert "k_tail" in err, err[-400:]
    print(name, "small" if small else "", "tail used" if "k_tail" in err else "no tail", rays[0])


@pytest.mark.gpu
def test_tail_with_seeded_sampling_and_many_frames_per_call(gpu, capfd, monkeypatch):
    """Seeded frames (the tail's phases draw from the rays' own streams, so the pictures are reproducible), and 16 frames per
    call, whose levels are 16 times as large and whose record is kept apart from the single frames'."""
    monkeypatch.setenv("RT_DEBUG_LEVELS", "1")
    scene_file, params, hi, lo, floor = helpers.load_stochastic("st_soft_shadows", 16, seed=1000)
    scene = rt.Scene(scene_file)
    a = scene.render(params)
    b = scene.render(params)
    c = scene.render(params)
    for img in (b, c):
        mx, nd, _ = helpers.diff_stats(img, a)
        assert mx <= 1 and nd <= 0.002 * a.size, (mx, nd)
    scene.close()
    scene_file, params, golden = helpers.load_case("c1_shadows_fresnel")
    params.rng_mode, params.seed = rt.RT_RNG_SEEDED, 77
    scene = rt.Scene(scene_file)
    first = scene.render_frames(params, 16).copy()
    capfd.readouterr()
    again = scene.render_frames(params, 16)
    err = capfd.readouterr().err
    single = scene.render(params)
    scene.close()
    assert "k_tail" in err, err[-300:]
    for f in range(16):
        mx, nd, _ = helpers.diff_stats(again[f], first[f])
        assert mx <= 1 and nd <= 0.002 * first[f].size, (f, mx, nd)
    mx, nd, _ = helpers.diff_stats(single, first[0])
    assert mx <= 1 and nd <= 0.002 * single.size, (mx, nd)
