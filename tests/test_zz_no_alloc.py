"""rt_render_device_noalloc (include/rt_mi355x.h): the wavefront render through the library's static __device__ workspace,
for callers that cannot allocate.  Checked through the C-ABI against rt_render_device on the same inputs, twice in a
row with the second call on a smaller input (the array keeps the first call's contents), to the deterministic bar:
8-bit max |diff| <= 1 (FP64 atomics may reorder a pixel's sum), same ray tree."""
import numpy as np
import pytest
import torch

import helpers
import raytracer_amd as rt

pytestmark = pytest.mark.gpu


def _both(scene, params, rows, width, stream):
    """rt_render_device and rt_render_device_noalloc on the same input.  The headline property of the latter -- nothing is
    allocated or freed on the device -- is checked with hipMemGetInfo around the call: the free bytes of the device do not
    move (torch's caching allocator is not involved: the output buffers exist before the measurement)."""
    dev = torch.device("cuda", 0)
    a = torch.zeros((rows, width, 3), dtype=torch.uint8, device=dev)
    b = torch.full((rows, width, 3), 77, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    scene.render_device(params, a.data_ptr(), stream=stream.cuda_stream)
    torch.cuda.synchronize(dev)
    free_before = torch.cuda.mem_get_info(dev)[0]
    scene.render_device_noalloc(params, b.data_ptr(), stream=stream.cuda_stream)
    free_during = torch.cuda.mem_get_info(dev)[0]
    torch.cuda.synchronize(dev)
    free_after = torch.cuda.mem_get_info(dev)[0]
    assert free_before == free_during == free_after, (free_before, free_during, free_after)
    return a.cpu().numpy(), b.cpu().numpy()


def test_noalloc_matches_render_device_twice_second_smaller(gpu):
    stream = torch.cuda.Stream(torch.device("cuda", 0))
    # first call: glass + environment + shadows (deep ray tree, shadow chains through glass), whole frame
    big_file, big_params, big_golden = helpers.load_case("own_glass_env")
    big = rt.Scene(big_file)
    a, b = _both(big, big_params, big.height, big.width, stream)
    mx, nd, _ = helpers.diff_stats(a, b)
    assert mx <= 1 and nd <= 1e-3 * a.size, (mx, nd)
    assert helpers.diff_stats(b, big_golden)[0] <= 1
    # second call, smaller input: one third of a smaller scene's frame (a compact slice of interleaved row tiles)
    small_file, small_params, small_golden = helpers.load_case("c1_shadows_fresnel")
    small = rt.Scene(small_file)
    p = small_params.copy()
    p.tile_rows, p.part_index, p.part_count = 8, 1, 3
    rows = rt.part_rows(small.height, 8, 1, 3)
    assert rows * small.width < big.height * big.width
    a2, b2 = _both(small, p, rows, small.width, stream)
    mx, nd, _ = helpers.diff_stats(a2, b2)
    assert mx <= 1 and nd <= 1e-3 * a2.size, (mx, nd)
    frame = np.zeros((small.height, small.width, 3), dtype=np.uint8)
    rt.part_scatter(b2, frame, 8, 1, 3)
    ys = [y for y in range(small.height) if (y // 8) % 3 == 1]
    assert helpers.diff_stats(frame[ys], small_golden[ys])[0] <= 1
    # and the first input again: nothing of the smaller call is left behind
    a3, b3 = _both(big, big_params, big.height, big.width, stream)
    assert helpers.diff_stats(a3, b3)[0] <= 1 and helpers.diff_stats(b3, b)[0] <= 1
    big.close()
    small.close()


def test_noalloc_one_ray_per_pixel_and_many_samples(gpu):
    """The direct-output path (no accumulators) and a 16-spp frame whose primary rays come in chunks."""
    stream = torch.cuda.Stream(torch.device("cuda", 0))
    f, p, golden = helpers.load_case("bvh_100_shaded")
    s = rt.Scene(f)
    a, b = _both(s, p, s.height, s.width, stream)
    assert np.array_equal(a, b) and helpers.diff_stats(b, golden)[0] <= 1
    s.close()
    f, p, hi, lo, psnr_ref = helpers.load_stochastic("st_soft_shadows", 16, 5)
    s = rt.Scene(f)
    a, b = _both(s, p, s.height, s.width, stream)
    # same samples, summed in another order (other chunks and batches): flat regions whose value sits on an 8-bit boundary
    # may flip by one level; the stochastic bar is the PSNR against the reference's 256-spp render
    mx, nd, _ = helpers.diff_stats(a, b)
    assert mx <= 1 and nd <= 0.05 * a.size, (mx, nd)
    assert helpers.psnr(b, hi) >= helpers.psnr(a, hi) - 0.1
    assert helpers.psnr(b, hi) >= psnr_ref - 0.5
    s.close()


def test_noalloc_rejects_what_does_not_fit(gpu):
    f, p, _ = helpers.load_case("c1_shadows_fresnel")
    import re
    text = open(helpers.os.path.join(helpers.GOLD, "scenes", "c1_shadows_fresnel.txt")).read()
    wide = re.sub(r"resolution\s+\d+\s+\d+", "resolution 1280 720", text)
    cfg = rt.Config(text=open(helpers.os.path.join(helpers.GOLD, "configs", "c1_shadows_fresnel.json")).read())
    sf = rt.SceneFile(text=wide, texture_root=helpers.os.path.join(helpers.GOLD, "scenes"), config=cfg)
    s = rt.Scene(sf)
    buf = torch.zeros((720, 1280, 3), dtype=torch.uint8, device="cuda:0")
    with pytest.raises(rt.RtError) as e:
        s.render_device_noalloc(p, buf.data_ptr())
    assert e.value.code == rt.RT_ERR_LIMIT and "at most" in str(e.value)
    # the same frame in parts that fit: 8 parts of 90 rows x 1280 pixels... still too wide
    p8 = p.copy()
    p8.tile_rows, p8.part_index, p8.part_count = 8, 0, 8
    with pytest.raises(rt.RtError):
        s.render_device_noalloc(p8, buf.data_ptr())
    pm = p.copy()
    pm.schedule = rt.RT_SCHEDULE_MEGAKERNEL
    with pytest.raises(rt.RtError) as e:
        s.render_device_noalloc(pm, buf.data_ptr())
    assert e.value.code == rt.RT_ERR_INVALID
    s.close()
