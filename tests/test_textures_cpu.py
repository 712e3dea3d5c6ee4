"""Texture loading of the scene parser (Code/utilities/scene.cpp:31-91): PPM directly, JPG / PNG through a PIL conversion in
a child process (python3, no shell) -- the parsed scene holds the same RGB8 rows either way.  No GPU needed."""
import ctypes as C
import os

import numpy as np
import pytest
from PIL import Image

import raytracer_amd as rt

SCENE = """CAMERA
location 0 -5 1
gaze_direction 0 1 0
up_vector 0 0 1
focal_length 50
sensor_size 36 24
resolution 32 18
END_CAMERA
SPHERE
translation 0 0 1
rotation_euler_radians 0 0 0
scale 1 1 1
texture_file %s
END_SPHERE
"""


def _texture_of(text, root):
    cfg = rt.Config(text="{}")
    sf = rt.SceneFile(text=text, texture_root=str(root), config=cfg)
    d = sf.desc
    if d.n_textures == 0:
        return None
    t = d.textures[0]
    return np.ctypeslib.as_array(t.rgb, shape=(t.height, t.width, 3)).copy()


@pytest.fixture
def picture():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(13, 21, 3), dtype=np.uint8)
    img[:4] = (img[:4] // 64) * 64   # some flat areas too
    return img


def test_png_texture_equals_the_same_image_as_ppm(tmp_path, picture):
    rt.write_ppm(str(tmp_path / "t.ppm"), picture)
    Image.fromarray(picture, "RGB").save(tmp_path / "t.png")
    a = _texture_of(SCENE % "t.ppm", tmp_path)
    b = _texture_of(SCENE % "t.png", tmp_path)
    assert np.array_equal(a, picture) and np.array_equal(b, picture)


def test_png_with_alpha_and_palette_is_converted_to_rgb(tmp_path, picture):
    rgba = np.dstack([picture, np.full(picture.shape[:2], 128, dtype=np.uint8)])
    Image.fromarray(rgba, "RGBA").save(tmp_path / "a.PNG")     # extension compared case-insensitively (scene.cpp:35-40)
    expect = np.asarray(Image.open(tmp_path / "a.PNG").convert("RGB"))
    assert np.array_equal(_texture_of(SCENE % "a.PNG", tmp_path), expect)


def test_jpg_texture_is_what_pil_decodes(tmp_path, picture):
    big = np.kron(picture, np.ones((8, 8, 1), dtype=np.uint8))
    Image.fromarray(big, "RGB").save(tmp_path / "t.jpg", quality=90)
    expect = np.asarray(Image.open(tmp_path / "t.jpg").convert("RGB"))
    got = _texture_of(SCENE % "t.jpg", tmp_path)
    assert got.shape == big.shape and np.array_equal(got, expect)
    assert np.abs(got.astype(int) - big.astype(int)).mean() < 12      # and it is the picture, lossy


def test_missing_or_broken_texture_only_warns(tmp_path):
    """scene.cpp:80-83: a texture that cannot be loaded leaves the shape untextured; the scene still parses."""
    assert _texture_of(SCENE % "nothing.png", tmp_path) is None
    (tmp_path / "broken.jpg").write_bytes(b"not a jpeg")
    assert _texture_of(SCENE % "broken.jpg", tmp_path) is None


def test_texture_name_is_never_run_by_a_shell(tmp_path, picture):
    """The reference splices the name into `python3 -c "..."` for std::system (scene.cpp:49-60); here the name travels as an
    argument of a child process, so quotes, backticks and $(...) are just characters of a file name."""
    marker = tmp_path / "pwned"
    for evil in ("x$(touch %s).png" % marker, "x`touch %s`.png" % marker, "x');open('%s','w');('.png" % marker, "x\";touch %s;\".png" % marker):
        assert _texture_of(SCENE % evil.replace(" ", "_"), tmp_path) is None
    assert not marker.exists() and not os.path.exists(str(marker).replace(" ", "_"))
    odd = "it's a $(name) `x`.png"
    Image.fromarray(picture, "RGB").save(tmp_path / odd)
    got = _texture_of(SCENE % odd, tmp_path)
    # the parser reads one whitespace-delimited token as the file name (scene.cpp:438), so a name with blanks is cut short
    # and simply not found; without blanks the odd characters load fine
    odd2 = "it's_a_$(name)_`x`.png"
    Image.fromarray(picture, "RGB").save(tmp_path / odd2)
    assert np.array_equal(_texture_of(SCENE % odd2, tmp_path), picture)
    assert got is None or np.array_equal(got, picture)
