"""GPU: geometry that stresses the conservative FP32 / 8-bit box tests of the traversal -- rays exactly parallel to
the axes (infinite reciprocals, NaNs in the slab arithmetic), box faces containing the ray, coordinates far from the
origin (coarse FP32 grid), very different object sizes in one hierarchy.  The hierarchy may only cull: every picture
must equal the oracle's, whose boxes and tests are FP64 (aabb.cpp:4-38), to the parity bar."""
import numpy as np
import pytest

import helpers
import oracle_py
import raytracer_amd as rt
from raytracer_amd import workloads

pytestmark = pytest.mark.gpu


def _camera(loc, gaze, up, w, h):
    cam = rt.RtCamera()
    cam.location[:] = loc
    cam.gaze_direction[:] = gaze
    cam.up_vector[:] = up
    cam.focal_length, cam.sensor_width, cam.sensor_height = 50.0, 36.0, 24.0
    cam.resolution_x, cam.resolution_y, cam.f_stop, cam.focal_distance = w, h, 99999.0, 10.0
    return cam


def _light(pos):
    l = rt.RtLight()
    l.position[:] = pos
    l.intensity[:] = [600.0, 600.0, 600.0]
    l.radius = 0.0
    return l


def _check(sb, params):
    ref, cnt = oracle_py.render(sb, params, counters=True)
    scene = rt.Scene(sb)
    out = []
    for schedule in (rt.RT_SCHEDULE_WAVEFRONT, rt.RT_SCHEDULE_MEGAKERNEL):
        p = params.copy()
        p.schedule = schedule
        img, st = scene.render(p, stats=True)
        mx, nd, n1 = helpers.diff_stats(img, ref)
        assert mx <= 1 and nd <= 0.01 * img.size, (schedule, mx, nd)
        assert st.rays == cnt[0]
        out.append(img)
    p = params.copy()
    p.use_bvh = 0
    mx, nd, n1 = helpers.diff_stats(scene.render(p), out[0])   # the reference's own invariant: BVH on == BVH off
    assert mx <= 1 and nd <= 0.01 * out[0].size
    scene.close()
    assert ref.std() > 5.0   # the picture is not empty


def _grid_shapes(offset):
    """Axis-aligned cubes and spheres, 1.25 apart with half-size 0.5: the faces x = 0 and z = 0 of the middle ones lie on
    the planes that contain the axis-parallel rays of the odd-sized image below.  The shapes do not touch: a hit exactly
    on the contact plane of two touching boxes is the one case where a conservative hierarchy and the reference may
    differ (findings)."""
    trs, types = [], []
    for ix in range(-2, 3):
        for iz in range(-2, 3):
            for iy in range(0, 3):
                trs.append([1.25 * ix + 0.5 + offset[0], 4.0 + 2.0 * iy + offset[1], 1.25 * iz + 0.5 + offset[2], 0, 0, 0, 0.5, 0.5, 0.5])
                types.append(rt.RT_SHAPE_CUBE if (ix + iz + iy) % 2 == 0 else rt.RT_SHAPE_SPHERE)
    trs = np.array(trs)
    shapes = rt.shapes_from_trs(trs, rt.RT_SHAPE_SPHERE, 0)
    shapes["type"] = np.array(types, dtype=shapes["type"].dtype)
    return shapes


@pytest.mark.parametrize("offset", [(0.0, 0.0, 0.0), (8192.0, -4096.0, 16384.0)])
def test_axis_parallel_rays_and_faces_on_ray_planes(gpu, offset):
    cfg = workloads.default_config()
    cam = _camera([offset[0], -6.0 + offset[1], offset[2]], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0], 65, 43)   # odd: a centre ray with d.x = d.z = 0
    sb = rt.SceneBuilder(cam, [_light([offset[0] + 3.0, offset[1] - 4.0, offset[2] + 5.0])], _grid_shapes(offset), [rt.default_material()])
    _check(sb, cfg.params(rng="d0", shadows=1))


def test_tiny_and_huge_objects_in_one_hierarchy(gpu):
    rng = np.random.default_rng(5)
    n = 3000
    trs = np.zeros((n + 2, 9))
    trs[:n, 0] = rng.uniform(-2, 2, n)
    trs[:n, 1] = rng.uniform(2, 6, n)
    trs[:n, 2] = rng.uniform(-1.5, 1.5, n)
    trs[:n, 3:6] = rng.uniform(-3, 3, (n, 3))
    trs[:n, 6:9] = rng.uniform(4e-3, 8e-2, (n, 3))
    trs[n] = [0, 4000.0, 0, 0, 0, 0, 3000.0, 3000.0, 3000.0]        # a wall-sized sphere far behind
    trs[n + 1] = [0, 4, -1003.0, 0.3, 0.2, 0.1, 1000.0, 1000.0, 1000.0]   # a floor-sized rotated cube below
    shapes = rt.shapes_from_trs(trs, rt.RT_SHAPE_SPHERE, 0)
    shapes["type"][n + 1] = rt.RT_SHAPE_CUBE
    cfg = workloads.default_config()
    cam = _camera([0.0, -3.0, 0.5], [0.0, 1.0, -0.05], [0.0, 0.0, 1.0], 96, 64)
    sb = rt.SceneBuilder(cam, [_light([2.0, 0.0, 6.0])], shapes, [rt.default_material()])
    _check(sb, cfg.params(rng="d0", shadows=1))
