"""CPU-side tests: the C-ABI library loads and exports every declared symbol, and the host logic kept from the
reference (config.json reader, scene parser, BVH build, PPM/PFM I/O, tile partition) behaves like the reference's."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import helpers
import raytracer_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rt_mi355x.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", header))
    assert declared == set(rt.ABI_SYMBOLS), declared ^ set(rt.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(rt.lib, name), name


def test_struct_sizes_match_header_layout():
    assert C.sizeof(rt.RtShape) == 8 + 8 * (16 + 16 + 12 + 3 + 1)
    assert C.sizeof(rt.RtMaterial) == 8 * 13 + 16
    assert C.sizeof(rt.RtLight) == 56
    assert C.sizeof(rt.RtCamera) == 8 * 12 + 8 + 16


def test_render_without_init_or_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(rt.RtError) as e:
        rt.init(0)
    assert e.value.code == rt.RT_ERR_DEVICE
    scene_file, params, _ = helpers.load_case("c1_default")
    with pytest.raises(rt.RtError) as e:
        rt.Scene(scene_file)
    assert e.value.code == rt.RT_ERR_NOT_INIT


CONFIG_TEXT = """{
  "settings": { // comment
    "samples_per_pixel": 4,
    "max_bounces": 7
  },
  "image": {
    "exposure": 0.25,
    "shutter_time": 2
  },
  "flags": {
    "on": true,
    "off": false
  },
  "background": {
    "r": 1
  }
}"""


def test_config_grammar_and_typing():
    cfg = rt.Config(text=CONFIG_TEXT)
    assert cfg.get_int("settings.samples_per_pixel", 1) == 4
    assert cfg.get_int("image.exposure", 9) == 0            # double read as int truncates (config.cpp:83)
    assert cfg.get_double("image.shutter_time", 0.0) == 2.0  # int read as double (config.cpp:92)
    assert cfg.get_bool("flags.on", False) is True and cfg.get_bool("flags.off", True) is False
    assert cfg.get_bool("settings.max_bounces", True) is True  # wrong type -> default (config.cpp:99)
    assert cfg.get_double("missing.key", 1.5) == 1.5
    p = cfg.params()
    assert (p.samples_per_pixel, p.max_depth, p.max_bounces) == (4, 7, 7)
    assert p.exposure == 0.25 and p.background[0] == 1.0 and p.background[1] == 0.7
    # call-site defaults differ when advanced.epsilon is absent (scene.cpp:95 vs plane.cpp:83)
    assert p.eps_scene == 1e-4 and p.eps_triangle == 1e-3 and p.eps_march == 1e-3 and p.eps_shadow == 1e-4
    assert p.ray_march_steps == 64 and p.step_multiplier == 0.8


def test_missing_config_file_uses_defaults():
    cfg = rt.Config(path="/nonexistent/config.json")
    assert cfg.loaded is False
    p = cfg.params()
    assert (p.samples_per_pixel, p.max_depth, p.max_bounces, p.shadow_samples) == (1, 10, 5, 4)
    assert list(p.ambient_light) == [0.15, 0.15, 0.25]


def test_scene_parser_counts_and_defaults():
    scene, params, _ = helpers.load_case("c1_default")
    d = scene.desc
    assert (d.n_shapes, d.n_lights, d.camera.resolution_x, d.camera.resolution_y) == (4, 1, 160, 90)
    assert [d.shapes[i].type for i in range(4)] == [rt.RT_SHAPE_SPHERE, rt.RT_SHAPE_PLANE, rt.RT_SHAPE_SPHERE, rt.RT_SHAPE_CUBE]
    assert d.lights[0].radius == 0.5 and d.camera.f_stop == 99999.0
    assert d.shutter_time == 0.5  # image.shutter_time of the shipped config (main.cpp:62)
    m = d.materials[d.shapes[0].material]
    assert m.shininess == 10.0 and m.reflectivity == 0.3 and m.is_metal == 0
    # T*Rz*Ry*Rx*S of an unrotated sphere: diagonal scale + translation, inverse exact
    t = np.array(d.shapes[0].transform).reshape(4, 4)
    assert np.allclose(np.diag(t)[:3], 1.509919) and np.allclose(t[:3, 3], [-1.608119, 1.469529, 0.337436])
    assert np.allclose(np.array(d.shapes[0].inv_transform).reshape(4, 4) @ t, np.eye(4), atol=1e-15)


def test_scene_parser_edge_cases():
    scene, params, _ = helpers.load_case("own_parser_edges")
    d = scene.desc
    # 3-corner plane skipped (scene.cpp:380-384); CUBE block closed by END_SPHERE becomes a sphere (scene.cpp:268)
    assert d.n_shapes == 3
    assert [d.shapes[i].type for i in range(3)] == [rt.RT_SHAPE_PLANE, rt.RT_SHAPE_SPHERE, rt.RT_SHAPE_SPHERE]
    assert d.materials[d.shapes[0].material].texture == -1  # missing texture file: warning, untextured
    assert list(d.camera.location) == [7.0, -7.0, 5.2]  # second CAMERA block wins, other fields persist
    assert d.camera.focal_length == 50.0


def test_scene_errors_match_reference():
    cfg = rt.Config(text="{}")
    with pytest.raises(rt.RtError) as e:
        rt.SceneFile(text="SPHERE\nEND_SPHERE\n", config=cfg)
    assert "No camera data found" in str(e.value)
    with pytest.raises(rt.RtError) as e:
        rt.SceneFile(text="CAMERA\n location 1 2\nEND_CAMERA\n", config=cfg)
    assert "Error reading vector components" in str(e.value)
    with pytest.raises(rt.RtError) as e:
        rt.SceneFile(path="/nonexistent/scene.txt", config=cfg)
    assert "Failed to open scene file" in str(e.value)


def test_crlf_scene_parses():
    scene, _, _ = helpers.load_case("gal_bike")
    assert scene.desc.n_shapes == 2647 + 0 or scene.desc.n_shapes > 2000


def test_bvh_build_matches_reference_structure():
    scene, params, _ = helpers.load_case("bvh_100_normals")
    info, order = rt.bvh_build_host(scene)
    assert info.n_prims == 100 and sorted(order.tolist()) == list(range(100))
    assert info.node_bytes == 64
    # median split: leaves hold 1-2 shapes, interior nodes = leaves - 1, depth = ceil(log2(100/2)) or so
    assert 49 <= info.n_nodes <= 99 and 5 <= info.max_depth <= 7
    scene1, _, _ = helpers.load_case("bvh_1_normals")
    info1, order1 = rt.bvh_build_host(scene1)
    assert info1.n_nodes == 0 and info1.max_depth == 0 and order1.tolist() == [0]
    # root box = union of the 8 transformed corners of every unit sphere (transformed_shape.h:42-77)
    assert all(info.root_min[k] < info.root_max[k] for k in range(3))


def test_bvh_million_sphere_build_is_balanced():
    rng = np.random.default_rng(7)
    n = 200_000
    trs = np.zeros((n, 9))
    trs[:, 0:3] = rng.uniform(-1, 1, (n, 3))
    trs[:, 5] = rng.uniform(-3, 3, n)
    trs[:, 6:9] = rng.uniform(0.002, 0.006, (n, 3))
    shapes = rt.shapes_from_trs(trs)
    cam = rt.RtCamera()
    cam.location[:] = [3, -3, 2]; cam.gaze_direction[:] = [-0.6, 0.6, -0.4]; cam.up_vector[:] = [0, 0, 1]
    cam.focal_length, cam.sensor_width, cam.sensor_height = 50, 36, 24
    cam.resolution_x, cam.resolution_y, cam.f_stop, cam.focal_distance = 64, 36, 99999.0, 10.0
    sb = rt.SceneBuilder(cam, [], shapes, [rt.default_material()])
    info, order = rt.bvh_build_host(sb)
    assert info.n_prims == n and info.max_depth <= 18 and len(set(order.tolist())) == n


def test_part_rows_and_scatter_roundtrip():
    h, w = 1080, 16
    frame = np.arange(h * w * 3, dtype=np.uint32).astype(np.uint8).reshape(h, w, 3)
    for parts, tile in ((1, 8), (2, 8), (3, 16), (8, 8), (8, 16), (7, 5)):
        rows = [rt.part_rows(h, tile, k, parts) for k in range(parts)]
        assert sum(rows) == h
        out = np.zeros_like(frame)
        n_tiles = (h + tile - 1) // tile
        for k in range(parts):
            sl = np.concatenate([frame[t * tile:min(h, (t + 1) * tile)] for t in range(k, n_tiles, parts)] or [frame[:0]])
            assert sl.shape[0] == rows[k]
            rt.part_scatter(sl, out, tile, k, parts)
        assert np.array_equal(out, frame)


def test_ppm_roundtrip_and_header(tmp_path):
    img = (np.arange(5 * 7 * 3) % 256).astype(np.uint8).reshape(5, 7, 3)
    path = str(tmp_path / "a.ppm")
    rt.write_ppm(path, img)
    raw = open(path, "rb").read()
    assert raw.startswith(b"P6\n7 5\n255\n") and len(raw) == 11 + 105  # Image.cpp:177-179
    assert np.array_equal(rt.read_ppm(path), img)
    tex = rt.read_ppm(os.path.join(helpers.GOLD, "Textures", "atlas.ppm"))  # ASCII P3 with a comment line
    assert tex.shape == (36, 48, 3)


def test_pfm_endianness_and_scale():
    a = rt.read_pfm(os.path.join(helpers.GOLD, "Textures", "sky.pfm"))
    b = rt.read_pfm(os.path.join(helpers.GOLD, "Textures", "sky_be.pfm"))  # big-endian, scale 2 of half the values
    assert a.shape == (32, 64, 3) and np.allclose(a, b, rtol=1e-6)


def test_config_reader_against_the_reference_class():
    """tests/golden/config_cases.json: texts (the reference tree's own config files and the corner cases of its line grammar)
    with the answers of the REFERENCE's Config class itself for every candidate key and all three getters, two defaults
    each (oracle/gen_config_cases.py).  rt_config_* must give the same answers -- same keys stored, same types, same
    prefix-parsed numbers, same defaults on a type mismatch."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_cases.json")) as f:
        cases = json.load(f)
    asked = held = 0
    for case in cases:
        cfg = rt.Config(text=case["text"])
        for key, ans in case["answers"].items():
            got_i = [cfg.get_int(key, -7), cfg.get_int(key, 13)]
            got_d = [cfg.get_double(key, -7.5), cfg.get_double(key, 13.25)]
            got_b = [int(cfg.get_bool(key, False)), int(cfg.get_bool(key, True))]
            assert got_i == [int(v) for v in ans["i"]], (case["text"], key, got_i, ans["i"])
            want_d = [float(v) for v in ans["d"]]
            assert all((a == b) or (a != a and b != b) for a, b in zip(got_d, want_d)), (case["text"], key, got_d, ans["d"])
            assert got_b == [int(v) for v in ans["b"]], (case["text"], key, got_b, ans["b"])
            asked += 1
            held += got_i[0] == got_i[1] or got_b[0] == got_b[1]
    assert asked > 300 and held > 150
