"""Host logic of the traversal hierarchy (no GPU): the 4-wide 8-bit quantised image the device traverses must be a
conservative cover of the reference's per-shape boxes (transformed_shape.h:42-77, plane.cpp:11-46) -- every node's
dequantised child box contains, with the outward pad, the boxes of all primitives below that child -- and must reach
every primitive exactly once within the depth the traversal stack allows."""
import numpy as np
import pytest

import helpers
import raytracer_amd as rt


def _child_boxes(node):
    """[4, 6] dequantised boxes (lo xyz, hi xyz) of one node, in float64 (exact: org is FP32, steps are powers of two)."""
    out = np.zeros((4, 6))
    for a in range(3):
        step = rt.node4_steps(node)[a]
        assert step > 0.0 and np.frexp(step)[0] == 0.5, "grid steps are powers of two, never zero"
        for c in range(4):
            out[c, a] = float(node["org"][a]) + float((int(node["qlo"][a]) >> (8 * c)) & 0xFF) * step
            out[c, 3 + a] = float(node["org"][a]) + float((int(node["qhi"][a]) >> (8 * c)) & 0xFF) * step
    return out


def _check_image(img, n_prims, check_root_box=True):
    nodes, boxes, pad = img["nodes"], img["prim_boxes"], img["pad"]
    seen = np.zeros(n_prims, dtype=np.int32)
    max_depth = 0

    def leaf_prims(ref):
        code = ~ref
        first, count = code >> 1, (code & 1) + 1
        return list(range(first, first + count))

    def walk(ref, depth):
        """-> (lo, hi) union of the primitive boxes below ref"""
        nonlocal max_depth
        if ref < 0:
            ps = leaf_prims(ref)
            for p in ps:
                seen[p] += 1
            return boxes[ps, 0:3].min(axis=0), boxes[ps, 3:6].max(axis=0)
        max_depth = max(max_depth, depth + 1)
        node = nodes[ref]
        cb = _child_boxes(node)
        lo_all, hi_all = np.full(3, np.inf), np.full(3, -np.inf)
        used = 0
        for c in range(4):
            ch = int(node["child"][c])
            if ch == rt.EMPTY_CHILD:
                assert all(((int(node["qlo"][a]) >> (8 * c)) & 0xFF) == 255 and ((int(node["qhi"][a]) >> (8 * c)) & 0xFF) == 0
                           for a in range(3)), "unused slot must hold an inverted box"
                continue
            used += 1
            lo, hi = walk(ch, depth + 1)
            assert np.all(cb[c, 0:3] <= lo - pad), (ref, c, cb[c], lo, pad)
            assert np.all(cb[c, 3:6] >= hi + pad), (ref, c, cb[c], hi, pad)
            # and not absurdly loose: within two grid steps + pad of the true box on every axis
            for a in range(3):
                step = rt.node4_steps(node)[a]
                assert cb[c, a] >= lo[a] - pad - 2 * step and cb[c, 3 + a] <= hi[a] + pad + 2 * step
            lo_all, hi_all = np.minimum(lo_all, lo), np.maximum(hi_all, hi)
        assert used >= 2
        return lo_all, hi_all

    if n_prims == 0:
        return
    lo, hi = walk(img["root_ref"], 0)
    assert np.all(seen == 1), "every primitive exactly once"
    info = img["info"]
    assert max_depth == info.max_depth
    assert 3 * info.max_depth <= 96   # device_types.h: kTravStackDepth
    if check_root_box:
        assert np.all(np.array(info.root_min[:]) <= lo - pad) and np.all(np.array(info.root_max[:]) >= hi + pad)


@pytest.mark.parametrize("name", ["bvh_100_normals", "c1_default", "gal_shapes", "own_complex", "own_motion", "bvh_1_normals", "bvh_2_normals"])
def test_device_image_covers_reference_boxes(name):
    scene, _, _ = helpers.load_case(name)
    img = rt.bvh_device_image(scene)
    assert img["info"].n_prims == scene.desc.n_shapes and img["info"].node_bytes == 64
    _check_image(img, scene.desc.n_shapes)


def _random_scene(n, seed, clustered):
    rng = np.random.default_rng(seed)
    trs = np.zeros((n, 9))
    if clustered:   # a few tight clusters far apart plus coincident centres: stresses the grid exponents and the SAH fallback
        centres = rng.uniform(-500, 500, (5, 3))
        trs[:, 0:3] = centres[rng.integers(0, 5, n)] + rng.normal(0, 1e-3, (n, 3))
        trs[: n // 10, 0:3] = centres[0]
        trs[:, 6:9] = rng.uniform(1e-4, 5e-4, (n, 3))
    else:
        trs[:, 0:3] = rng.uniform(-1, 1, (n, 3))
        trs[:, 6:9] = rng.uniform(0.002, 0.05, (n, 3))
    trs[:, 3:6] = rng.uniform(-3, 3, (n, 3))
    shapes = rt.shapes_from_trs(trs)
    cam = rt.RtCamera()
    cam.location[:] = [3, -3, 2]; cam.gaze_direction[:] = [-0.6, 0.6, -0.4]; cam.up_vector[:] = [0, 0, 1]
    cam.focal_length, cam.sensor_width, cam.sensor_height = 50, 36, 24
    cam.resolution_x, cam.resolution_y, cam.f_stop, cam.focal_distance = 64, 36, 99999.0, 10.0
    return rt.SceneBuilder(cam, [], shapes, [rt.default_material()])


@pytest.mark.parametrize("n,clustered", [(3, False), (4, False), (5, False), (1000, False), (3000, True)])
def test_device_image_random_scenes(n, clustered):
    sb = _random_scene(n, 11 + n, clustered)
    img = rt.bvh_device_image(sb)
    assert img["info"].n_prims == n
    _check_image(img, n)
    # four children per node: about n/3 nodes for n single-primitive leaves
    if n >= 1000:
        assert img["info"].n_nodes <= 0.5 * n


def test_device_image_is_shallower_than_the_stack_for_a_large_scene():
    sb = _random_scene(200_000, 5, False)
    img = rt.bvh_device_image(sb)
    assert img["info"].n_prims == 200_000 and img["info"].max_depth <= 20
    leaves = int((img["nodes"]["child"] < 0).sum())
    assert leaves == 200_000   # single-primitive leaves from the SAH builder


@pytest.mark.parametrize("clustered", [False, True])
def test_builds_do_not_depend_on_the_number_of_host_threads(clustered, monkeypatch):
    """host_bvh.cpp hands subtrees of more than 8192 primitives to other threads; every node and leaf has a position
    known in advance, so the order, the hierarchy and its 4-wide image are the same bytes for any RT_BUILD_THREADS."""
    sb = _random_scene(70_000, 23, clustered)
    results = []
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("RT_BUILD_THREADS", threads)
        info, order = rt.bvh_build_host(sb)
        img = rt.bvh_device_image(sb)
        results.append((order.tobytes(), img["nodes"].tobytes(), img["root_ref"], img["info"].max_depth, info.n_nodes, info.max_depth))
    assert results[0] == results[1] == results[2]
    assert sorted(np.frombuffer(results[0][0], dtype=np.int32).tolist()) == list(range(70_000))


def test_traversal_boxes_of_spheres_bound_the_ellipsoid_exactly(monkeypatch):
    """The traversal hierarchy bounds a transformed sphere by its ellipsoid's own extent (host_bvh.cpp: traversal_bounds),
    not by the reference's box of eight transformed cube corners (transformed_shape.h:42-77).  Per primitive: (a) every point
    of the surface -- M p + c for unit vectors p, the matrix the reference's corner box uses -- lies inside the box, and the
    extreme points touch it (so it is exact, not merely smaller); (b) the box lies inside the reference's box, which
    RT_REFERENCE_BOXES=1 brings back; (c) for a moving sphere it covers both ends of the shutter interval."""
    n = 300
    rng = np.random.default_rng(3)
    trs = np.zeros((n, 9))
    trs[:, 0:3] = rng.uniform(-2, 2, (n, 3))
    trs[:, 3:6] = rng.uniform(-3.2, 3.2, (n, 3))
    trs[:, 6:9] = rng.uniform(0.01, 0.8, (n, 3))
    shapes = rt.shapes_from_trs(trs)
    shapes["velocity"][: n // 3] = rng.uniform(-1, 1, (n // 3, 3))
    cam = rt.RtCamera()
    cam.location[:] = [3, -3, 2]; cam.gaze_direction[:] = [-0.6, 0.6, -0.4]; cam.up_vector[:] = [0, 0, 1]
    cam.focal_length, cam.sensor_width, cam.sensor_height = 50, 36, 24
    cam.resolution_x, cam.resolution_y, cam.f_stop, cam.focal_distance = 64, 36, 99999.0, 10.0
    shutter = 0.5
    sb = rt.SceneBuilder(cam, [], shapes, [rt.default_material()], shutter_time=shutter)
    _, order = rt.bvh_build_host(sb)
    tight = rt.bvh_device_image(sb)["prim_boxes"]
    monkeypatch.setenv("RT_REFERENCE_BOXES", "1")
    loose = rt.bvh_device_image(sb)["prim_boxes"]
    monkeypatch.delenv("RT_REFERENCE_BOXES")
    _check_image(rt.bvh_device_image(sb), n)
    p = rng.normal(size=(4000, 3))
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    shrink = []
    for k in range(n):
        s = shapes[order[k]]
        m = np.array(s["transform"]).reshape(4, 4)
        pts = p @ m[:3, :3].T + m[:3, 3]
        moved = pts + np.array(s["velocity"]) * shutter
        for q in (pts, moved):
            assert np.all(q >= tight[k, 0:3] - 1e-12) and np.all(q <= tight[k, 3:6] + 1e-12)
        assert np.all(tight[k, 0:3] >= loose[k, 0:3] - 1e-12) and np.all(tight[k, 3:6] <= loose[k, 3:6] + 1e-12)
        # exact: the extreme point of the ellipsoid along +x, M M^T e_x / |M^T e_x|, touches the upper x plane (still sphere: no motion)
        if not np.any(s["velocity"]):
            row = m[0, :3]
            ext = m[:3, :3] @ (row / np.linalg.norm(row)) + m[:3, 3]
            assert abs(ext[0] - tight[k, 3]) <= 1e-8 * max(1.0, abs(ext[0]))
        shrink.append(np.prod(tight[k, 3:6] - tight[k, 0:3]) / np.prod(loose[k, 3:6] - loose[k, 0:3]))
    assert np.mean(shrink) < 0.7   # rotated ellipsoids: the boxes hold well under the volume of the corner boxes
