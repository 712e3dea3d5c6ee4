"""TEST INFRASTRUCTURE ONLY (oracle).

Runs the real reference renderer (oracle/_ref/ref_raytracer, compiled by oracle/Makefile from the
sources where they lie under /root/reference/Code plus the reproducible-RNG shim oracle/ref_shim.h)
on one scene and returns the image it writes.

The reference binary hard-codes cwd-relative paths (Code/main.cpp:49,518,530; scene.cpp:171,270):
    cwd = X/Code/build ; X/Code/config.json ; X/ASCII/scene.txt ; X/Output/scene_test.ppm ; X/Textures/
so a throw-away directory X with that layout is created for every run.
"""
import os
import re
import shutil
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_BIN = os.path.join(HERE, "_ref", "ref_raytracer")


def read_ppm(path):
    """Binary P6 reader (maxval 255) -> uint8 array [H, W, 3]."""
    with open(path, "rb") as f:
        data = f.read()
    m = re.match(rb"P6\s+(\d+)\s+(\d+)\s+(\d+)\s", data)
    if not m:
        raise ValueError("not a P6 ppm: %s" % path)
    w, h, mx = int(m.group(1)), int(m.group(2)), int(m.group(3))
    assert mx == 255
    px = np.frombuffer(data, dtype=np.uint8, offset=m.end(), count=w * h * 3)
    return px.reshape(h, w, 3).copy()


def override_resolution(scene_text, width, height):
    """Rewrite only the camera `resolution` line of a scene (goldens at reduced size)."""
    out, n = re.subn(r"(?m)^(\s*resolution)\s+\d+\s+\d+", r"\1 %d %d" % (width, height), scene_text)
    if n == 0:
        raise ValueError("scene has no resolution line")
    return out


def run_reference(scene_text, config_text, flags=(), rng="d0", texture_dirs=(), threads=None, keep=False):
    """Render `scene_text` with the reference binary.  Returns (image uint8 [H,W,3], stdout)."""
    if not os.path.exists(REF_BIN):
        raise FileNotFoundError("build it first: make -C oracle ref")
    root = tempfile.mkdtemp(prefix="rtref_")
    try:
        os.makedirs(os.path.join(root, "Code", "build"))
        os.makedirs(os.path.join(root, "ASCII"))
        os.makedirs(os.path.join(root, "Output"))
        os.makedirs(os.path.join(root, "Textures"))
        for d in texture_dirs:
            for name in os.listdir(d):
                src = os.path.abspath(os.path.join(d, name))
                if os.path.isfile(src):
                    os.symlink(src, os.path.join(root, "Textures", name))
        with open(os.path.join(root, "Code", "config.json"), "w") as f:
            f.write(config_text)
        with open(os.path.join(root, "ASCII", "scene.txt"), "w") as f:
            f.write(scene_text)
        env = dict(os.environ)
        env["RT_ORACLE_RNG"] = rng
        if threads is not None:
            env["OMP_NUM_THREADS"] = str(threads)
        p = subprocess.run([REF_BIN] + list(flags), cwd=os.path.join(root, "Code", "build"), env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=False)
        if p.returncode != 0:
            raise RuntimeError("reference failed (%d):\n%s" % (p.returncode, p.stdout[-2000:]))
        img = read_ppm(os.path.join(root, "Output", "scene_test.ppm"))
        return img, p.stdout
    finally:
        if not keep:
            shutil.rmtree(root, ignore_errors=True)
