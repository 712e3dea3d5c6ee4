"""GPU: linear-radiance output (rt_render_linear) and progressive accumulation built on it."""
import os

import numpy as np
import pytest

import helpers
import oracle_py
import raytracer_amd as rt
from raytracer_amd import progressive

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["c1_shadows_fresnel", "bvh_100_shaded", "own_glass_env", "ex_tonemap_aces"])
def test_linear_matches_oracle(gpu, name):
    """Same FP64 arithmetic on both sides: libm and summation order only (tolerance 1e-9 relative to the brightest value)."""
    scene_file, params, golden = helpers.load_case(name)
    ref = oracle_py.render_float(scene_file, params)
    scene = rt.Scene(scene_file)
    lin = scene.render_linear(params)
    img = scene.render(params)
    scene.close()
    finite = np.isfinite(ref) & np.isfinite(lin)
    assert np.array_equal(np.isfinite(ref), np.isfinite(lin))
    scale = max(1.0, float(np.abs(ref[finite]).max()))
    close = np.abs(lin[finite] - ref[finite]) <= 1e-9 * scale
    assert close.mean() > 0.999, (name, float(close.mean()))   # a handful of pixels sit on a tie / refraction boundary
    mx, nd, n1 = helpers.diff_stats(progressive.to_image(lin, int(params.tonemap)), img)
    assert mx <= 1 and nd <= 0.001 * img.size


def test_progressive_passes_and_resume(gpu, tmp_path):
    """4 passes x 4 spp are a 16-spp estimate: the PSNR of the reference's own independent 16-spp render against its
    256-spp render, minus 1.5 dB for the spread between two 16-spp realisations; 16 passes (64 spp) must beat that
    16-spp figure by 2 dB (expected +6 dB minus the noise left in the 256-spp target).  A run resumed from a checkpoint
    continues the series of seeds and ends on the same picture as an uninterrupted one."""
    scene_file, params, hi, lo, psnr_ref = helpers.load_stochastic("st_soft_shadows", 4, 100)
    scene = rt.Scene(scene_file)
    whole = progressive.ProgressiveRender(scene, params).add_passes(4)
    assert whole.samples == 16
    assert helpers.psnr(whole.image(), hi) >= psnr_ref - 1.5
    first = progressive.ProgressiveRender(scene, params).add_passes(2)
    path = os.path.join(tmp_path, "ckpt.npz")
    first.save(path)
    resumed = progressive.ProgressiveRender(scene, params).load(path).add_passes(2)
    assert resumed.passes == 4
    assert np.allclose(resumed.linear(), whole.linear(), rtol=1e-12, atol=1e-12)
    mx, nd, n1 = helpers.diff_stats(resumed.image(), whole.image())
    assert mx <= 1
    more = progressive.ProgressiveRender(scene, params).load(path).add_passes(14)
    assert more.samples == 64
    assert helpers.psnr(more.image(), hi) >= psnr_ref + 2.0
    scene.close()
