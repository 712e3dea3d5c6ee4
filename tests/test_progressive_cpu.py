"""Host side of float output and progressive accumulation (no GPU): the numpy tone map / 8-bit truncation of
raytracer_amd.progressive against the oracle's own (tracer.h:23-54, 297-308), and the PFM writer against the reader."""
import os

import numpy as np
import pytest

import helpers
import oracle_py
import raytracer_amd as rt
from raytracer_amd import progressive


@pytest.mark.parametrize("name", ["c1_default", "ex_tonemap_aces", "ex_tonemap_filmic", "ex_tonemap_reinhard", "ex_exposure_high"])
def test_numpy_tonemap_matches_oracle_bytes(name):
    scene, params, golden = helpers.load_case(name)
    linear = oracle_py.render_float(scene, params)
    img = progressive.to_image(linear, int(params.tonemap))
    ref = oracle_py.render(scene, params)
    mx, nd, n1 = helpers.diff_stats(img, ref)
    assert mx <= 1 and nd <= 0.001 * img.size, (mx, nd)
    mx, nd, n1 = helpers.diff_stats(img, golden)   # and therefore the reference's own picture
    assert mx <= 1


def test_quantise_edge_values():
    v = np.array([np.nan, -1.0, 0.0, 0.5, 1.0, 2.0, np.inf, -np.inf, 254.9999 / 255.0])
    assert progressive.quantise(v).tolist() == [255, 0, 0, 127, 255, 255, 255, 0, 254]


def test_pfm_roundtrip(tmp_path):
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 4, (5, 7, 3))
    path = os.path.join(tmp_path, "a.pfm")
    rt.write_pfm(path, img)
    back = rt.read_pfm(path)              # rows in file order: bottom to top (HDRImage.cpp:112-182)
    assert back.shape == (5, 7, 3)
    assert np.array_equal(back[::-1], img.astype(np.float32))
