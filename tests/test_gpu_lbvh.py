"""The hierarchy built ON the GPU (bvh_build.hip: Morton codes, radix sort, binary radix tree,
segment-tree boxes, 4-wide collapse, 8-bit quantisation) against the same two bars as the host builder's:

* structure -- the node array read back from device memory is a conservative cover of the reference's per-shape boxes
  (transformed_shape.h:42-77, plane.cpp:11-46) and reaches every primitive exactly once within the traversal stack's
  depth: the checker of tests/test_bvh_device_image.py, unchanged;
* frames -- the closest hit does not depend on the hierarchy and ties are decided by the reference's visit order, so
  the frames must be the bytes the SAH-built scene renders, and within the north-star bar (8-bit max |diff| <= 1) of
  the reference's golden images."""
import numpy as np
import pytest

import helpers
import raytracer_amd as rt
from test_bvh_device_image import _check_image, _random_scene

pytestmark = pytest.mark.gpu


@pytest.fixture()
def lbvh(gpu):
    rt.set_bvh_builder("lbvh")
    yield
    rt.set_bvh_builder("default")


def _device_image(desc):
    scene = rt.Scene(desc)
    img = scene.bvh_nodes()
    img["prim_boxes"] = rt.bvh_device_image(desc)["prim_boxes"]   # primitive order is the reference's for every builder
    return scene, img


@pytest.mark.parametrize("name", ["bvh_100_normals", "c1_default", "gal_shapes", "gal_bike", "own_complex", "own_motion", "bvh_2_normals"])
def test_device_built_image_covers_reference_boxes(lbvh, name):
    if name not in helpers.MANIFEST:
        pytest.skip("no such golden case")
    desc, _, _ = helpers.load_case(name)
    scene, img = _device_image(desc)
    scene.close()
    assert img["builder"] == "lbvh" and img["info"].n_prims == desc.desc.n_shapes
    _check_image(img, desc.desc.n_shapes, check_root_box=False)


@pytest.mark.parametrize("n,clustered", [(2, False), (3, False), (5, False), (1000, False), (3000, True), (50_000, False)])
def test_device_built_image_random_scenes(lbvh, n, clustered):
    """clustered: five tight clusters 1000 units apart and n/10 coincident centres (equal Morton codes)."""
    sb = _random_scene(n, 11 + n, clustered)
    scene, img = _device_image(sb)
    scene.close()
    assert img["builder"] == "lbvh"
    _check_image(img, n, check_root_box=False)
    leaves = int((img["nodes"]["child"] < 0).sum())
    assert leaves == n   # single-primitive leaves
    if n >= 1000:
        assert img["info"].n_nodes <= 0.5 * n


@pytest.mark.parametrize("name", ["bvh_100_shaded", "c1_shadows_fresnel", "gal_shapes", "own_glass_env", "own_complex", "own_motion"])
def test_frames_do_not_depend_on_the_builder(gpu, name):
    desc, params, golden = helpers.load_case(name)
    frames = {}
    for builder in ("sah", "lbvh"):
        rt.set_bvh_builder(builder)
        try:
            scene = rt.Scene(desc)
        finally:
            rt.set_bvh_builder("default")
        assert scene.bvh_nodes()["builder"] == builder
        frames[builder] = scene.render(params)
        scene.close()
    mx, nd, _ = helpers.diff_stats(frames["lbvh"], golden)
    assert mx <= 1 and nd <= 0.01 * golden.size
    # FP64 atomics of the wavefront schedule may reorder a pixel's sum: <= 1 level, rarely
    mx, nd, _ = helpers.diff_stats(frames["lbvh"], frames["sah"])
    assert mx <= 1 and nd <= 1e-4 * golden.size


def test_large_scene_builds_in_milliseconds_and_renders_the_same_frame(gpu):
    """200 000 random ellipsoids: device build time, depth, and the frame against the SAH-built scene."""
    from raytracer_amd import workloads
    desc = workloads.bvh_stress_scene(200_000, 480, 270)
    params = workloads.default_config().params(rng="d0")
    out = {}
    for builder in ("sah", "lbvh"):
        rt.set_bvh_builder(builder)
        try:
            scene = rt.Scene(desc)
        finally:
            rt.set_bvh_builder("default")
        img = scene.bvh_nodes()
        assert img["builder"] == builder
        out[builder] = (scene.render(params), img)
        scene.close()
    assert np.array_equal(out["sah"][0], out["lbvh"][0])
    img = out["lbvh"][1]
    assert img["info"].max_depth <= 32 and 0.0 < img["device_build_ms"] < 50.0
    assert int((img["nodes"]["child"] < 0).sum()) == 200_000


def test_a_tree_deeper_than_the_stack_falls_back_to_the_host_builder(gpu, monkeypatch):
    """The device builder reports a tree with more 4-wide levels than the traversal stack allows (32; lowered here
    through RT_LBVH_MAX_LEVELS, since only adversarial scenes of millions of shapes reach 32) and rt_scene_create then
    uses the host builder, which bounds its depth."""
    desc, params, golden = helpers.load_case("bvh_100_shaded")
    monkeypatch.setenv("RT_LBVH_MAX_LEVELS", "2")
    rt.set_bvh_builder("lbvh")
    try:
        scene = rt.Scene(desc)
    finally:
        rt.set_bvh_builder("default")
    img = scene.bvh_nodes()
    assert img["builder"] == "sah" and img["info"].max_depth > 2
    mx, nd, _ = helpers.diff_stats(scene.render(params), golden)
    scene.close()
    assert mx <= 1


@pytest.mark.parametrize("n", [2, 3, 5, 7, 64, 1000, 1040, 66_001, 70_001, 300_000])   # 1040, 66001: levels that mix spans of 32 (one thread) and 33 (wave reduction)
def test_device_order_equals_the_host_order(gpu, n):
    """bvh_order.hip against host_bvh.cpp: build_bvh, which restates the reference's top-down median split
    (bvh.cpp:32-92): the same permutation of the shapes, element for element."""
    sb = _random_scene(n, 100 + n, False)
    _, host = rt.bvh_build_host(sb)
    dev, ms = rt.bvh_order_device(sb)
    assert np.array_equal(host, dev)


def test_device_order_reports_ties(gpu):
    """Equal box centres at a split: the reference's result depends on its std::nth_element, so the device pass refuses
    (RT_ERR_LIMIT) and rt_scene_create uses the host pass."""
    sb = _random_scene(3000, 3011, True)   # 300 coincident centres
    with pytest.raises(rt.RtError) as e:
        rt.bvh_order_device(sb)
    assert e.value.code == rt.RT_ERR_LIMIT


def test_large_scenes_take_their_order_from_the_device(gpu, monkeypatch):
    from raytracer_amd import workloads
    desc = workloads.bvh_stress_scene(100_000, 320, 180)
    params = workloads.default_config().params(rng="d0")
    scene = rt.Scene(desc)
    a, info_a = scene.render(params), scene.bvh_nodes()
    scene.close()
    monkeypatch.setenv("RT_ORDER_ON_HOST", "1")
    scene = rt.Scene(desc)
    b, info_b = scene.render(params), scene.bvh_nodes()
    scene.close()
    assert info_a["order_on_device"] and info_a["device_order_ms"] > 0 and not info_b["order_on_device"]
    assert np.array_equal(a, b) and np.array_equal(info_a["nodes"], info_b["nodes"])


@pytest.fixture()
def sah_gpu(gpu):
    rt.set_bvh_builder("sah_gpu")
    yield
    rt.set_bvh_builder("default")


@pytest.mark.parametrize("name", ["bvh_100_normals", "c1_default", "gal_shapes", "gal_bike", "own_complex", "own_motion", "bvh_2_normals"])
def test_device_sah_image_covers_reference_boxes(sah_gpu, name):
    desc, _, _ = helpers.load_case(name)
    scene, img = _device_image(desc)
    scene.close()
    assert img["builder"] == "sah_gpu"
    _check_image(img, desc.desc.n_shapes, check_root_box=False)


@pytest.mark.parametrize("n,clustered", [(2, False), (3, False), (5, False), (65, False), (1000, False), (3000, True), (50_000, False)])
def test_device_sah_is_the_host_sah_tree(sah_gpu, n, clustered):
    """bvh_sah.hip builds host_bvh.cpp's binned-SAH tree: same number of 4-wide nodes and depth as the host image and --
    compared as sets, since the two collapses number their nodes differently -- the same nodes.  (clustered: 300
    coincident centres, where the host's balanced fallback orders by index and the device cuts by position: the trees
    may then differ below that span, and only the structure is checked.)"""
    sb = _random_scene(n, 11 + n, clustered)
    scene, img = _device_image(sb)
    scene.close()
    assert img["builder"] == "sah_gpu"
    _check_image(img, n, check_root_box=False)
    host = rt.bvh_device_image(sb)
    if not clustered:
        assert img["info"].n_nodes == host["info"].n_nodes and img["info"].max_depth == host["info"].max_depth

        def boxes_of(nodes):   # node contents without the child numbering: grid + quantised planes + leaf references
            out = []
            for nd in nodes:
                leaves = tuple(sorted(int(c) for c in nd["child"] if c < 0))
                out.append((tuple(nd["org"].tolist()), rt.node4_steps(nd), tuple(sorted(zip(*[[(int(nd["qlo"][a]) >> (8 * c)) & 255 for c in range(4)] for a in range(3)] +
                                                                                      [[(int(nd["qhi"][a]) >> (8 * c)) & 255 for c in range(4)] for a in range(3)]))), leaves))
            return sorted(out)
        if n <= 1000:
            assert boxes_of(img["nodes"]) == boxes_of(host["nodes"])


def test_device_sah_renders_like_the_host_sah(gpu):
    """200 000 ellipsoids: the same frame, and (within 1 %) the same number of box and primitive tests as with the host-built tree."""
    from raytracer_amd import workloads
    desc = workloads.bvh_stress_scene(200_000, 480, 270)
    params = workloads.default_config().params(rng="d0")
    out = {}
    for builder in ("sah", "sah_gpu"):
        rt.set_bvh_builder(builder)
        try:
            scene = rt.Scene(desc)
        finally:
            rt.set_bvh_builder("default")
        assert scene.bvh_nodes()["builder"] == builder
        out[builder] = scene.render(params, stats=True)
        scene.close()
    (a, sa), (b, sb) = out["sah"], out["sah_gpu"]
    assert np.array_equal(a, b)
    # the same tree; leaves are tested a little earlier or later depending on how the waves happen to run, so the counts
    # agree closely, not exactly
    for x, y in ((sa.nodes_visited, sb.nodes_visited), (sa.prims_tested, sb.prims_tested), (sa.nodes_fetched, sb.nodes_fetched)):
        assert abs(int(x) - int(y)) <= 0.01 * int(x)


def test_large_degenerate_scene_on_the_default_path(gpu):
    """70 000 shapes in five tight clusters, 7 000 of them with one and the same centre: the device order pass refuses
    (ties at splits -> host pass), the device SAH build cuts the coincident span by position; the hierarchy must still
    cover every shape, and the frame must be the host-built one."""
    sb = _random_scene(70_000, 77, True)
    params = rt.Config(text="{}").params(rng="d0", normals=1)
    out = {}
    for builder in ("default", "sah"):
        rt.set_bvh_builder(builder)
        try:
            scene = rt.Scene(sb)
        finally:
            rt.set_bvh_builder("default")
        img = scene.bvh_nodes()
        img["prim_boxes"] = rt.bvh_device_image(sb)["prim_boxes"]
        out[builder] = (scene.render(params), img)
        scene.close()
    assert out["default"][1]["builder"] == "sah_gpu" and not out["default"][1]["order_on_device"]
    _check_image(out["default"][1], 70_000, check_root_box=False)
    assert np.array_equal(out["default"][0], out["sah"][0])


@pytest.mark.parametrize("builder", ["lbvh", "sah_gpu"])
@pytest.mark.parametrize("name", helpers.CASES)
def test_every_golden_case_with_device_built_hierarchies(gpu, name, builder):
    """All deterministic reference images (8-bit max |diff| <= 1) through hierarchies built on the GPU."""
    desc, params, golden = helpers.load_case(name)
    rt.set_bvh_builder(builder)
    try:
        scene = rt.Scene(desc)
    finally:
        rt.set_bvh_builder("default")
    built = scene.bvh_nodes()["builder"]
    img = scene.render(params)
    scene.close()
    assert built == (builder if desc.desc.n_shapes >= 2 else "sah")
    mx, nd, _ = helpers.diff_stats(img, golden)
    assert mx <= 1 and nd <= 0.01 * img.size, (name, builder, mx, nd)
