"""GPU tests of the asynchronous side of the C-ABI: rt_render_device on several HIP streams at once (each stream has
its own ray queues inside the library), the one-ray-per-pixel direct output path, the traversal-kernel timing, and
the 10^6-primitive stress frame (BASELINE config 2) at full size through size-independent properties."""
import numpy as np
import pytest
import torch

import helpers
import oracle_py
import raytracer_amd as rt
from raytracer_amd import workloads

pytestmark = pytest.mark.gpu


def _render_on_streams(scene, params, n_streams, rounds, shape):
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    bufs = [torch.zeros(shape, dtype=torch.uint8, device=dev) for _ in range(n_streams)]
    torch.cuda.synchronize(dev)   # the fills ran on torch's default stream; the renders below do not wait for it
    for _ in range(rounds):
        for s, b in zip(streams, bufs):
            scene.render_device(params, b.data_ptr(), stream=s.cuda_stream)
    torch.cuda.synchronize(dev)
    return streams, [b.cpu().numpy() for b in bufs]


@pytest.mark.parametrize("name", ["bvh_100_shaded", "c1_shadows_fresnel"])
def test_concurrent_streams_render_the_same_frame(gpu, name):
    """bvh_100_shaded spawns no rays (asynchronous, direct output); c1_shadows_fresnel reads queue sizes back."""
    scene_file, params, golden = helpers.load_case(name)
    scene = rt.Scene(scene_file)
    ref = scene.render(params)
    streams, imgs = _render_on_streams(scene, params, 4, 3, ref.shape)
    for img in imgs:
        assert np.array_equal(img, ref)
    mx, nd, n1 = helpers.diff_stats(ref, golden)
    assert mx <= 1
    ms, launches = scene.traverse_time(streams[0].cuda_stream)
    assert launches >= 3 and ms > 0.0
    assert scene.traverse_time(streams[0].cuda_stream) == (0.0, 0)   # collected once
    scene.close()


def test_direct_output_equals_accumulated_output(gpu):
    """1 spp without secondary rays writes pixels straight from the shading kernel; 2 spp of the same deterministic
    sample goes through the FP64 accumulators and the resolve kernel.  Same picture either way."""
    scene_file, params, golden = helpers.load_case("bvh_100_shaded")
    scene = rt.Scene(scene_file)
    a = scene.render(params)
    p2 = params.copy()
    p2.samples_per_pixel = 2   # RT_RNG_D0: both samples are the pixel centre
    b = scene.render(p2)
    scene.close()
    mx, nd, n1 = helpers.diff_stats(a, b)
    assert mx <= 1 and nd <= 0.001 * a.size


def test_stress_frame_properties(gpu):
    """BASELINE config 2 at full size (10^6 ellipsoids, 1920x1080).  The CPU oracle needs ~1 s per frame on 16 threads, so
    the oracle is compared on every 8th row; the whole frame is checked through properties that do not depend on the
    size: both schedules, a 3-way partition and a frame cut into row tiles agree bit for bit, and one ray per pixel."""
    cfg = workloads.default_config()
    sd = workloads.bvh_stress_scene()
    params = cfg.params(rng="d0")
    scene = rt.Scene(sd)
    img, st = scene.render(params, stats=True)
    assert st.rays == 1920 * 1080
    ref_rows, _ = oracle_py.render(sd, params, threads=16, counters=True, rows=(0, 1080, 8))
    mx, nd, n1 = helpers.diff_stats(img[0:1080:8], ref_rows[0:1080:8])
    assert mx <= 1 and nd <= 0.01 * img[0:1080:8].size
    pm = params.copy()
    pm.schedule = rt.RT_SCHEDULE_MEGAKERNEL
    assert np.array_equal(scene.render(pm), img)
    frame = np.zeros_like(img)
    for k in range(3):
        pk = params.copy()
        pk.tile_rows, pk.part_index, pk.part_count = 8, k, 3
        part = scene.render(pk)
        rows = [y for y in range(1080) if (y // 8) % 3 == k]
        frame[rows] = part[rows]
    assert np.array_equal(frame, img)
    scene.close()


@pytest.mark.parametrize("name", ["c1_shadows_fresnel", "fin_glossy_on", "own_glass_env"])
def test_small_queues_force_chunks_and_batches(gpu, name, monkeypatch):
    """RT_QUEUE_RAYS at its minimum (4096 rays): the frame is generated in many chunks of 8-row bands and every level of
    the ray tree is shaded in batches sized by its worst-case fan-out.  Same picture, same number of rays."""
    scene_file, params, golden = helpers.load_case(name)
    scene = rt.Scene(scene_file)
    ref, st_ref = scene.render(params, stats=True)
    scene.close()
    monkeypatch.setenv("RT_QUEUE_RAYS", "4096")
    scene = rt.Scene(scene_file)   # a new scene: the queue size is fixed when a scene's queues are first allocated
    img, st = scene.render(params, stats=True)
    scene.close()
    mx, nd, n1 = helpers.diff_stats(img, ref)
    assert mx <= 1 and nd <= 0.001 * img.size
    assert st.rays == st_ref.rays
    mx, nd, n1 = helpers.diff_stats(img, golden)
    assert mx <= 1


def test_empty_scene_is_background(gpu):
    """No shapes at all: every pixel is the background colour of the config (tracer.h:282-293), on both schedules."""
    cfg = workloads.default_config()
    cam = rt.RtCamera()
    cam.location[:] = [0, -3, 0]; cam.gaze_direction[:] = [0, 1, 0]; cam.up_vector[:] = [0, 0, 1]
    cam.focal_length, cam.sensor_width, cam.sensor_height = 50, 36, 24
    cam.resolution_x, cam.resolution_y, cam.f_stop, cam.focal_distance = 48, 27, 99999.0, 10.0
    sb = rt.SceneBuilder(cam, [], rt.shapes_from_trs(np.zeros((0, 9))), [rt.default_material()])
    params = cfg.params(rng="d0")
    ref = oracle_py.render(sb, params)
    scene = rt.Scene(sb)
    for schedule in (rt.RT_SCHEDULE_WAVEFRONT, rt.RT_SCHEDULE_MEGAKERNEL):
        p = params.copy()
        p.schedule = schedule
        img, st = scene.render(p, stats=True)
        assert np.array_equal(img, ref) and st.rays == 48 * 27
        assert len(np.unique(img.reshape(-1, 3), axis=0)) == 1
    scene.close()


def test_small_queues_with_many_samples(gpu, monkeypatch):
    """16 spp through 4096-ray queues: primary rays are generated in chunks of rows x sample batches (the sample loop of
    main.cpp:288 is split), and the seeded picture is the same as with one big queue (same samples, other summation order)."""
    scene_file, params, hi, lo, psnr_ref = helpers.load_stochastic("st_soft_shadows", 16, 5)
    scene = rt.Scene(scene_file)
    ref = scene.render(params)
    scene.close()
    monkeypatch.setenv("RT_QUEUE_RAYS", "4096")
    scene = rt.Scene(scene_file)
    img, st = scene.render(params, stats=True)
    scene.close()
    mx, nd, n1 = helpers.diff_stats(img, ref)
    assert mx <= 1 and nd <= 0.01 * img.size
    assert helpers.psnr(img, hi) >= psnr_ref - 0.5 or helpers.psnr(img, hi) >= helpers.psnr(ref, hi) - 0.1


@pytest.mark.parametrize("name", ["c1_shadows_fresnel", "own_glass_env", "bvh_100_shaded"])
def test_levels_without_round_trips_equal_level_by_level(gpu, name, monkeypatch):
    """A one-chunk frame enqueues all levels of its ray tree with the queue sizes left on the device
    (render_wavefront.hip: run_levels); RT_SYNC_LEVELS=1 reads every size back and batches.  Same rays, same frame (FP64
    atomics may reorder a pixel's sum: <= 1 level)."""
    desc, params, golden = helpers.load_case(name)
    out = {}
    for mode in ("levels", "sync"):
        if mode == "sync":
            monkeypatch.setenv("RT_SYNC_LEVELS", "1")
        scene = rt.Scene(desc)
        out[mode] = scene.render(params, stats=True)
        scene.close()
    (a, sa), (b, sb) = out["levels"], out["sync"]
    assert int(sa.rays) == int(sb.rays) and sa.rays > 0
    mx, nd, _ = helpers.diff_stats(a, b)
    assert mx <= 1 and nd <= 1e-3 * a.size
    mx, nd, _ = helpers.diff_stats(a, golden)
    assert mx <= 1


def test_a_level_that_outgrows_its_queue_restarts_the_frame_level_by_level(gpu, monkeypatch):
    """Queue capacity cut to just what the primary rays' children need (RT_QUEUE_RAYS): the glass scene's third level is
    larger than that, the kernels flag the overflow without writing past the queue, and the frame is rendered again
    with batches: same ray count and frame as with ample queues."""
    desc, params, golden = helpers.load_case("own_glass_env")
    scene = rt.Scene(desc)
    ref, st_ref = scene.render(params, stats=True)
    scene.close()
    fan = 1 + 1 + (2 * desc.desc.n_lights * params.shadow_samples if params.shadows else 0)
    cap = max(4096, desc.width * desc.height * fan)
    monkeypatch.setenv("RT_QUEUE_RAYS", str(cap))
    scene = rt.Scene(desc)
    img, st = scene.render(params, stats=True)
    scene.close()
    assert int(st.rays) == int(st_ref.rays)
    mx, nd, _ = helpers.diff_stats(img, ref)
    assert mx <= 1 and nd <= 1e-3 * img.size
    # the aborted attempt leaves nothing behind in the test counters or in the traversal timing (they describe ONE frame):
    # the primitive and box tests of a scene this small do not depend on the schedule
    assert int(st.prims_tested) == int(st_ref.prims_tested) and int(st.nodes_visited) == int(st_ref.nodes_visited)
    assert 0.0 < st.traverse_ms < 3.0 * st_ref.traverse_ms + 1.0


def _render_counted(scene_file, params, monkeypatch, one_at_a_time):
    """One frame through rt_render_device without the statistics variant of the kernels (a frame that counts its tests
    renders its chunks one at a time): the picture and the rays the call adds to its device counters."""
    if one_at_a_time:
        monkeypatch.setenv("RT_ONE_CHUNK_AT_A_TIME", "1")
    else:
        monkeypatch.delenv("RT_ONE_CHUNK_AT_A_TIME", raising=False)
    dev = torch.device("cuda", 0)
    scene = rt.Scene(scene_file)
    stream = torch.cuda.Stream(dev)
    buf = torch.zeros((scene.height, scene.width, 3), dtype=torch.uint8, device=dev)
    counters = torch.zeros(8, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    for _ in range(2):   # the second frame reuses both sets of queues
        scene.render_device(params, buf.data_ptr(), d_counters_ptr=counters.data_ptr(), stream=stream.cuda_stream)
    torch.cuda.synchronize(dev)
    rt.release_workspaces()
    scene.close()
    return buf.cpu().numpy(), int(counters[0].item()) // 2


@pytest.mark.parametrize("name", ["c1_shadows_fresnel", "own_glass_env"])
def test_chunks_two_at_a_time_render_the_same_frame(gpu, name, monkeypatch):
    """A frame of four chunks or more renders them two at a time, the odd ones through a second set of queues on a
    stream and host thread of the library's own (render_wavefront.hip: ws.partner): same frame and ray count as one chunk
    at a time and as one big queue.  own_glass_env's third level outgrows a 4096-ray queue: both halves stop and the
    frame starts over level by level."""
    scene_file, params, golden = helpers.load_case(name)
    scene = rt.Scene(scene_file)
    ref, st_ref = scene.render(params, stats=True)
    scene.close()
    monkeypatch.setenv("RT_QUEUE_RAYS", "4096")
    one, rays_one = _render_counted(scene_file, params, monkeypatch, True)
    two, rays_two = _render_counted(scene_file, params, monkeypatch, False)
    assert rays_one == int(st_ref.rays) and rays_two == int(st_ref.rays)
    for img in (one, two):
        mx, nd, _ = helpers.diff_stats(img, ref)
        assert mx <= 1 and nd <= 0.001 * img.size
        assert helpers.diff_stats(img, golden)[0] <= 1


@pytest.mark.parametrize("how", ["prepare", "acquire"])
def test_a_frame_that_loses_its_second_set_of_queues_still_renders(gpu, monkeypatch, how):
    """The partner workspace of the chunk pairs is an optimisation, never a requirement: when it cannot be prepared (device
    full) the frame renders its chunks one at a time; when one of its lazily allocated queues does not fit any more in the
    middle of the frame, what both halves have added is dropped and the frame starts over alone -- same picture, same ray
    count, no error (RT_TEST_PARTNER_FAIL simulates the two out-of-memory conditions)."""
    scene_file, params, golden = helpers.load_case("c1_shadows_fresnel")
    monkeypatch.setenv("RT_QUEUE_RAYS", "4096")
    one, rays_one = _render_counted(scene_file, params, monkeypatch, True)
    monkeypatch.setenv("RT_TEST_PARTNER_FAIL", how)
    lost, rays_lost = _render_counted(scene_file, params, monkeypatch, False)
    monkeypatch.delenv("RT_TEST_PARTNER_FAIL")
    assert rays_lost == rays_one
    mx, nd, _ = helpers.diff_stats(lost, one)
    assert mx <= 1 and nd <= 0.001 * one.size and helpers.diff_stats(lost, golden)[0] <= 1


def test_chunks_two_at_a_time_with_many_samples(gpu, monkeypatch):
    """16 spp through 4096-ray queues: chunks of one pixel row band split by sample batches add into the same pixels
    from both streams (FP64 atomics)."""
    scene_file, params, hi, lo, psnr_ref = helpers.load_stochastic("st_soft_shadows", 16, 5)
    scene = rt.Scene(scene_file)
    ref = scene.render(params)
    scene.close()
    monkeypatch.setenv("RT_QUEUE_RAYS", "4096")
    one, rays_one = _render_counted(scene_file, params, monkeypatch, True)
    two, rays_two = _render_counted(scene_file, params, monkeypatch, False)
    assert rays_one == rays_two and rays_two > 0
    for img in (one, two):
        mx, nd, _ = helpers.diff_stats(img, ref)
        assert mx <= 1 and nd <= 0.01 * img.size


@pytest.mark.parametrize("name", ["st_dof", "st_motion_blur"])
def test_full_size_frames_in_chunk_pairs_equal_one_chunk_at_a_time(gpu, name, monkeypatch, tmp_path):
    """BASELINE configs 3 (1080p, 64 spp) and 5 (3840x2160, 16 of its samples) at full size, where the chunks are 2^26 .. 2^27 rays:
    two at a time and one at a time give the same ray count and the same frame up to the order of the FP64 additions."""
    label, _, w, h, spp, rng = {c[1]: c for c in workloads.BASELINE_CONFIGS}[name]
    scene_file, params = workloads.golden_scene(name, w, h, spp, rng, sky_dir=str(tmp_path))
    dev = torch.device("cuda", 0)
    scene = rt.Scene(scene_file)
    out = {}
    for mode in ("pairs", "one"):
        if mode == "one":
            monkeypatch.setenv("RT_ONE_CHUNK_AT_A_TIME", "1")
        buf = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
        counters = torch.zeros(8, dtype=torch.int64, device=dev)
        stream = torch.cuda.Stream(dev)
        torch.cuda.synchronize(dev)
        scene.render_device(params, buf.data_ptr(), d_counters_ptr=counters.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        out[mode] = (buf.cpu().numpy(), int(counters[0].item()))
    scene.close()
    rt.release_workspaces()
    assert out["pairs"][1] == out["one"][1] > w * h * spp
    mx, nd, _ = helpers.diff_stats(out["pairs"][0], out["one"][0])
    assert mx <= 1 and nd <= 1e-4 * out["one"][0].size


def test_render_on_torchs_default_stream_is_ordered_with_torch_work(gpu):
    """hip_stream NULL is HIP's null stream == torch's default stream: a torch op enqueued right after the render reads the
    finished frame without any synchronisation in between (the multi-GPU recipe of INTEGRATION.md relies on this)."""
    scene_file, params, golden = helpers.load_case("c1_shadows_fresnel")
    scene = rt.Scene(scene_file)
    ref = scene.render(params)
    dev = torch.device("cuda", 0)
    for stream_arg in (None, torch.cuda.current_stream(dev).cuda_stream):
        buf = torch.zeros(ref.shape, dtype=torch.uint8, device=dev)     # fill on the default stream, render behind it
        scene.render_device(params, buf.data_ptr(), stream=stream_arg)
        copy = buf.clone()                                              # default stream again: ordered behind the render
        mx, nd, _ = helpers.diff_stats(copy.cpu().numpy(), ref)
        assert mx <= 1 and nd <= 1e-3 * ref.size
    # a frame without secondary rays is enqueued fully asynchronously: the same ordering must hold
    scene.close()
    scene_file, params, golden = helpers.load_case("bvh_100_shaded")
    scene = rt.Scene(scene_file)
    ref = scene.render(params)
    buf = torch.zeros(ref.shape, dtype=torch.uint8, device=dev)
    scene.render_device(params, buf.data_ptr())
    assert np.array_equal(buf.clone().cpu().numpy(), ref)
    scene.close()


def test_queues_outlive_scenes_and_can_be_reserved(gpu):
    """Workspaces belong to the stream, not the scene: rt_reserve allocates and touches them ahead of time, scenes come and
    go, and frames stay the same; rt_release_workspaces frees them and the next render allocates again."""
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    before = torch.cuda.mem_get_info(dev)[0]
    rt.reserve(stream.cuda_stream, 1 << 20, 3)       # 3 queues x 2^20 rays x (96 + 32) B = 403 MB
    after = torch.cuda.mem_get_info(dev)[0]
    assert before - after >= 0.9 * 3 * (1 << 20) * 128   # the allocator may reuse a few MB it kept
    out = []
    for name in ("c1_shadows_fresnel", "own_glass_env", "c1_shadows_fresnel"):
        scene_file, params, golden = helpers.load_case(name)
        scene = rt.Scene(scene_file)
        buf = torch.zeros((scene.height, scene.width, 3), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        scene.render_device(params, buf.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        assert helpers.diff_stats(buf.cpu().numpy(), golden)[0] <= 1
        out.append(buf.cpu().numpy())
        scene.close()
    assert helpers.diff_stats(out[0], out[2])[0] <= 1
    assert torch.cuda.mem_get_info(dev)[0] <= after + (64 << 20)   # closing scenes did not free the queues
    rt.release_workspaces()
    assert torch.cuda.mem_get_info(dev)[0] >= after + 0.9 * 3 * (1 << 20) * 128 - (64 << 20)
    with pytest.raises(rt.RtError):
        rt.reserve(stream.cuda_stream, 0, 1)
    scene_file, params, golden = helpers.load_case("c1_shadows_fresnel")
    scene = rt.Scene(scene_file)
    assert helpers.diff_stats(scene.render(params), golden)[0] <= 1   # allocates again
    scene.close()


def test_history_order_of_the_chunks_changes_no_pixel(gpu):
    """k_traverse hands out the chunks of a primary-ray queue longest-first, by what the previous launch over such a queue
    recorded (k_chunk_order): scheduling only.  The first frame of a scene on a fresh stream has no history, the following ones
    do; single frames and launches of several frames, deterministic and seeded: all equal to the first / to the frames rendered
    one by one, and to the megakernel, which has no queue at all."""
    dev = torch.device("cuda", 0)
    sd = workloads.bvh_stress_scene(200_000, 1920, 1080)
    scene = rt.Scene(sd)
    p = workloads.default_config().params(rng="d0")
    stream = torch.cuda.Stream(dev)
    bufs = [torch.zeros((1080, 1920, 3), dtype=torch.uint8, device=dev) for _ in range(4)]
    torch.cuda.synchronize(dev)
    for b in bufs:
        scene.render_device(p, b.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
    pm = p.copy()
    pm.schedule = rt.RT_SCHEDULE_MEGAKERNEL
    mega = scene.render(pm)
    assert mega.any()
    for b in bufs:
        assert np.array_equal(b.cpu().numpy(), mega)
    ps = workloads.default_config().params(rng="seeded", seed=40)
    many = torch.zeros((3, 4, 1080, 1920, 3), dtype=torch.uint8, device=dev)
    for k in range(3):   # the same four frames three times: the second and third launch follow the first one's record
        scene.render_frames_device(ps, 4, many[k].data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
    assert torch.equal(many[0], many[1]) and torch.equal(many[0], many[2])
    q = ps.copy()
    for f in range(4):
        q.seed = ps.seed + f
        q.schedule = rt.RT_SCHEDULE_MEGAKERNEL
        assert np.array_equal(many[2, f].cpu().numpy(), scene.render(q)), f
    scene.close()
