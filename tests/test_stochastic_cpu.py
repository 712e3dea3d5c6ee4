"""Stochastic features (soft shadows, glossy reflection, depth of field, motion blur): the oracle's seeded renders are
checked against the REAL reference's high-spp render by PSNR — north_star's bound: an N-spp render must be no further
from the high-spp reference than the reference's own independent N-spp render, minus 0.5 dB."""
import pytest

import helpers
import oracle_py


@pytest.mark.parametrize("name", helpers.STOCHASTIC)
def test_oracle_psnr_matches_reference_estimator(name):
    spp = helpers.MANIFEST[name]["spp_lo"]
    scene, params, hi, lo, bar = helpers.load_stochastic(name, spp, seed=11)
    img = oracle_py.render(scene, params)
    got = helpers.psnr(img, hi)
    assert got >= bar - 0.5, "%s: PSNR %.2f dB vs reference's own %.2f dB" % (name, got, bar)
