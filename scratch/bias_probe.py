"""Bias and per-sample variance of the GPU samplers against the CPU oracle's (the restatement of the reference's samplers, itself
held to the reference's 256-spp noise floor by tests/test_stochastic_cpu.py), in LINEAR radiance (no 8-bit truncation), per
stochastic fixture:  two GPU renders (seeds A, B) of N_g spp, two oracle renders of N_o spp.
  rms(gA - gB)^2 / 2 = v_g / N_g,  rms(oA - oB)^2 / 2 = v_o / N_o   -> per-sample variances v_g, v_o (should be equal)
  rms(gA - oA)^2 should be v_g / N_g + v_o / N_o; the excess is bias^2.
Values are clamped to [0, 1] and scaled to 8-bit levels.  Usage: python scratch/bias_probe.py [N_g] [N_o]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import helpers  # noqa: E402
import oracle_py  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import progressive  # noqa: E402

N_G = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N_O = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rt.init(0)


def lv(x):
    return 255.0 * np.clip(x, 0.0, 1.0)


def rms(a, b, mask=None):
    d = (a - b) ** 2
    return float(np.sqrt(d.mean() if mask is None else d[mask].mean()))


for name in helpers.STOCHASTIC:
    scene_file, params, hi, lo, _ = helpers.load_stochastic(name, 16, seed=1000)
    scene = rt.Scene(scene_file)
    g = []
    for seed in (1000, 500000):
        p = params.copy()
        p.seed = seed
        g.append(lv(progressive.ProgressiveRender(scene, p).add_passes(N_G // 16).linear()))
    scene.close()
    o = []
    for seed in (77, 900001):
        p = params.copy()
        p.seed, p.samples_per_pixel = seed, N_O
        o.append(lv(oracle_py.render_float(scene_file, p)))
    vg, vo = rms(g[0], g[1]) ** 2 / 2 * N_G, rms(o[0], o[1]) ** 2 / 2 * N_O
    gm, om = 0.5 * (g[0] + g[1]), 0.5 * (o[0] + o[1])
    expect = np.sqrt(vg / (2 * N_G) + vo / (2 * N_O))
    got = rms(gm, om)
    d = gm - om
    worst = np.unravel_index(np.argmax(np.abs(d)), d.shape)
    print("%-16s per-sample variance gpu %.1f oracle %.1f (ratio %.3f) | rms(gpu mean - oracle mean) %.4f levels, noise alone %.4f -> bias rms %.4f | mean diff %.4f, max |diff| %.3f at %s"
          % (name, vg, vo, vg / vo, got, expect, np.sqrt(max(0.0, got * got - expect * expect)), d.mean(), np.abs(d).max(), worst), flush=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "bias_%s.npz" % name), gpu=gm, oracle=om)
