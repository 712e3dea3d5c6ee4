"""GPU stochastic parity: seeded HIP renders (both schedules) of the distributed-effect scenes against the real
reference's high-spp render.  Bar (north_star / BASELINE.md): PSNR(gpu N spp, reference hi spp) >=
PSNR(reference N spp, reference hi spp) - 0.5 dB; and different seeds must give different, equally good images."""
import numpy as np
import pytest

import helpers
import raytracer_amd as rt

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("schedule", [rt.RT_SCHEDULE_WAVEFRONT, rt.RT_SCHEDULE_MEGAKERNEL])
@pytest.mark.parametrize("name", helpers.STOCHASTIC)
def test_psnr_against_reference_high_spp(gpu, name, schedule):
    spp = helpers.MANIFEST[name]["spp_lo"]
    scene_file, params, hi, lo, bar = helpers.load_stochastic(name, spp, seed=3)
    params.schedule = schedule
    scene = rt.Scene(scene_file)
    a = scene.render(params)
    params.seed = 4
    b = scene.render(params)
    scene.close()
    pa, pb = helpers.psnr(a, hi), helpers.psnr(b, hi)
    assert min(pa, pb) >= bar - 0.5, "%s: PSNR %.2f / %.2f dB vs reference's own %.2f dB" % (name, pa, pb, bar)
    assert not np.array_equal(a, b)


def test_seeded_render_is_reproducible_per_pixel_sample(gpu):
    """Same seed -> same random numbers per (pixel, sample): the megakernel reproduces bit for bit; the wavefront
    schedule sums a pixel's contributions with atomics, so only the order of FP64 additions may differ (<= 1 level)."""
    scene_file, params, hi, lo, bar = helpers.load_stochastic("st_soft_shadows", 4, seed=9)
    scene = rt.Scene(scene_file)
    params.schedule = rt.RT_SCHEDULE_MEGAKERNEL
    assert np.array_equal(scene.render(params), scene.render(params))
    params.schedule = rt.RT_SCHEDULE_WAVEFRONT
    mx, nd, n1 = helpers.diff_stats(scene.render(params), scene.render(params))
    scene.close()
    assert mx <= 1
