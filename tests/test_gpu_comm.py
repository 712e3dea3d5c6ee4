"""GPU tests of the multi-GPU exchange behind the C-ABI (raytracer_amd/csrc/comm.hip): the k_deinterleave kernel against
the host-side rt_part_scatter, the RCCL send/recv path through a one-rank communicator that sends to itself
(RT_COMM_LOOPBACK=1; the box has one GPU, and RCCL refuses two ranks on one device), rt_render_gathered, and the whole
multi-rank control flow of bench.py started by bench.py itself (`--gpus 2 --backend gloo`: two ranks share the GPU, the
slices are staged through the host -- a rehearsal of the launcher and the bookkeeping, not of RCCL)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import helpers
import raytracer_amd as rt

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _max_rows(h, tile, world):
    return max(rt.part_rows(h, tile, k, world) for k in range(world))


@pytest.mark.parametrize("width,height,tile,world,frames", [(1920, 1080, 8, 8, 2), (160, 90, 8, 3, 1), (37, 29, 4, 2, 3),
                                                             (64, 50, 16, 5, 1), (48, 27, 8, 1, 2)])
def test_deinterleave_kernel_equals_part_scatter(gpu, width, height, tile, world, frames):
    """Random slices per (rank, frame) -> frames on the device == rt_part_scatter on the host, both with rank 0's block inside
    the gathered buffer and with rank 0's slices elsewhere (d_own); 16-byte and byte-wise copy paths (odd widths)."""
    rng = np.random.default_rng(width * 7 + world)
    mr = _max_rows(height, tile, world)
    gathered = rng.integers(0, 256, size=(world, frames, mr, width, 3), dtype=np.uint8)
    expect = np.zeros((frames, height, width, 3), dtype=np.uint8)
    for f in range(frames):
        for k in range(world):
            rows = rt.part_rows(height, tile, k, world)
            rt.part_scatter(gathered[k, f, :rows], expect[f], tile, k, world)
    dev = torch.device("cuda", 0)
    d_g = torch.from_numpy(gathered).to(dev)
    out = torch.zeros((frames, height, width, 3), dtype=torch.uint8, device=dev)
    rt.deinterleave_device(None, d_g.data_ptr(), out.data_ptr(), frames, width, height, tile, world)
    assert np.array_equal(out.cpu().numpy(), expect)
    own = d_g[0].clone()
    d_g[0].fill_(0)   # rank 0's block of the gathered buffer is not read when d_own is given
    out.zero_()
    rt.deinterleave_device(own.data_ptr(), d_g.data_ptr() if world > 1 else None, out.data_ptr(), frames, width, height, tile, world)
    assert np.array_equal(out.cpu().numpy(), expect)


def test_rccl_path_with_a_loopback_communicator(gpu, monkeypatch):
    """ncclGetUniqueId / ncclCommInitRank / grouped ncclSend + ncclRecv / k_deinterleave on the one GPU of the box: rank 0 of
    a one-rank communicator sends its slices to itself.  Frames gathered this way equal the frames rendered whole."""
    monkeypatch.setenv("RT_COMM_LOOPBACK", "1")
    assert rt.comm_world() == 1
    uid = rt.comm_unique_id()
    assert len(uid) == rt.RT_COMM_ID_BYTES
    rt.comm_init(0, 1, uid)
    try:
        with pytest.raises(rt.RtError):
            rt.comm_init(0, 1, uid)   # one communicator per process
        scene_file, params, golden = helpers.load_case("c1_shadows_fresnel")
        scene = rt.Scene(scene_file)
        whole = scene.render(params)
        img, st = scene.render_gathered(params, stats=True)
        assert helpers.diff_stats(img, whole)[0] <= 1 and helpers.diff_stats(img, golden)[0] <= 1 and st.rays > 0
        # two frames in one message, on a torch stream
        dev = torch.device("cuda", 0)
        stream = torch.cuda.Stream(dev)
        H, W = scene.height, scene.width
        slices = torch.zeros((2, H, W, 3), dtype=torch.uint8, device=dev)
        frames = torch.zeros((2, H, W, 3), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        p2 = params.copy()
        p2.normals = 1
        scene.render_device(params, slices[0].data_ptr(), stream=stream.cuda_stream)
        scene.render_device(p2, slices[1].data_ptr(), stream=stream.cuda_stream)
        rt.gather_frames(slices.data_ptr(), 2, W, H, 8, frames.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        assert torch.equal(frames, slices)
        assert helpers.diff_stats(frames[0].cpu().numpy(), whole)[0] <= 1
        scene.close()
    finally:
        rt.comm_destroy()
    assert rt.comm_world() == 1


def test_bench_starts_its_own_ranks(gpu):
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts two ranks itself and reports n_gpus 2 with the gathered frame
    equal to the frame rendered whole (gloo rehearsal on the one GPU; a small scene, no CPU baseline; batches of 3 frames per
    launch and exchange, the last batch a partial one)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--spheres", "20000",
                          "--steps", "7", "--warmup", "1", "--frames-per-launch", "3", "--no-configs", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["gathered_frame_equals_whole_frame"] is True and res["value"] > 0
    # 7 steps in launches of 3: 3 + 3 + 1, and the last exchange ships ONE frame's slices, not the set's three
    ph = res["phases"]
    per_frame = 544 * 1920 * 3   # max over the ranks of rt_part_rows(1080, 8, k, 2): 68 of the 135 tiles
    assert ph["batches"] == 3 and ph["bytes_per_exchange_per_rank"] == [3 * per_frame, 3 * per_frame, per_frame]
    for who in ("max_over_ranks", "rank0"):
        assert set(ph[who]) == {"render_ms", "gather_ms", "exchange_ms", "deinterleave_ms"} and ph[who]["render_ms"] > 0 and ph[who]["gather_ms"] > 0
    assert res["steady_state"] is None   # (--frames-per-launch given and not larger than the steady setting)


def test_bench_never_launches_more_frames_than_steps(gpu):
    """The driver's own arguments are small (--steps 20 --warmup 5) against the default launch of 16 frames per GPU of the job:
    the batch is cut to the steps, a job of several GPUs gets at least two batches (one exchange overlaps one render), and the
    line says what the same job sustains over a long region (`steady_state`)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--spheres", "20000",
                          "--steps", "5", "--warmup", "2", "--no-configs", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    per_frame = 544 * 1920 * 3   # max over the ranks of rt_part_rows(1080, 8, k, 2): 68 of the 135 tiles
    assert res["config"]["frames_per_launch"] == 3 and res["phases"]["bytes_per_exchange_per_rank"] == [3 * per_frame, 2 * per_frame]
    assert res["steady_state"]["frames_per_launch"] == 32 and res["steady_state"]["Mrays_per_s"] > 0
    assert res["config"]["gathered_frame_equals_whole_frame"] is True
