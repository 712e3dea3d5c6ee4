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


def _seeded(scene, params, seed):
    p = params.copy()
    p.seed = seed
    return oracle_py.render(scene, p)


@pytest.mark.parametrize("name", helpers.STOCHASTIC)
def test_oracle_has_no_bias_at_the_256_spp_noise_floor(name):
    """The restatement's samplers against the reference's at a level where bias shows (the 16-spp bar above is all
    variance): both bars of helpers.py against the reference's sixteen independent 256-spp renders (oracle/gen_golden_r3.py)."""
    scene, params, _, _, _ = helpers.load_stochastic(name, 1024, seed=5)
    failed = helpers.check_bias(oracle_py.render(scene, params), name, "oracle 1024 spp")
    assert failed is None, failed
    params.samples_per_pixel = 256
    failed = helpers.check_bias16([_seeded(scene, params, 1000 * k) for k in range(1, 17)], name, "oracle")
    assert failed is None, failed


# One sampler parameter of each distributed effect moved by 10 %: the scene edit, applied to the fixture's text
_TEN_PERCENT_OFF = {
    "st_soft_shadows": (r"(?m)^(\s*radius\s+)([0-9.eE+-]+)", "light radius"),          # shading.h:31-36
    "st_dof": (r"(?m)^(\s*f_stop\s+)([0-9.eE+-]+)", "aperture (f_stop)"),                  # camera.cpp:59-64,13-24
    "st_glossy": (r"(?m)^(\s*shininess\s+)([0-9.eE+-]+)", "glossy lobe (roughness = 1/sqrt(shininess))"),   # tracer.h:184-195
    "st_motion_blur": (None, "shutter time"),                                            # main.cpp:299
}


@pytest.mark.parametrize("name", helpers.STOCHASTIC)
def test_bias_bars_can_fail(name):
    """Power of the bias test: the same oracle with ONE sampler parameter off by 10 % -- the light's radius, the aperture,
    the shutter time, the width of the glossy lobe: what a wrong disk / ball / time sampler would amount to -- must FAIL
    helpers.check_bias16, although it passes the 16-spp variance bar."""
    import os
    import re
    import raytracer_amd as rt
    pattern, what = _TEN_PERCENT_OFF[name]
    with open(os.path.join(helpers.GOLD, "scenes", name + ".txt"), newline="") as f:
        text = f.read()
    with open(os.path.join(helpers.GOLD, "configs", name + ".json")) as f:
        cfg = rt.Config(text=f.read())
    kw, shutter = helpers.flags_to_kwargs(helpers.MANIFEST[name]["flags"])
    if pattern is None:
        shutter = shutter * 1.1
    else:
        scale = 1.0 / 1.1 if "f_stop" in pattern else (1.0 / 1.21 if "shininess" in pattern else 1.1)
        text, n = re.subn(pattern, lambda m: "%s%.9g" % (m.group(1), float(m.group(2)) * scale), text)
        assert n >= 1
    scene = rt.SceneFile(text=text, texture_root=os.path.join(helpers.GOLD, "scenes"), config=cfg, shutter_time=shutter)
    params = cfg.params(rng="seeded", seed=5, **dict(kw, aa=256))
    failed = helpers.check_bias16([_seeded(scene, params, 1000 * k) for k in range(1, 17)], name, "10 % off: " + what)
    assert failed is not None, "a %s that is 10 %% off passes the bias bars of %s" % (what, name)
