"""world_size-2 and -3 gloo runs (CPU) of the multi-GPU frame assembly used by bench.py: every rank fills the slice it
would have rendered (rows of a known frame, interleaved row tiles), the slices are gathered to rank 0 and
de-interleaved; rank 0 must end up with the known frame (and, with several frames per gather as bench.py sends them, with
every frame of the batch).  The GPU render itself is covered by the -m gpu tests
(test_row_tile_partition_reassembles)."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    import raytracer_amd as rt
    from raytracer_amd.distributed import FrameGather
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    W, H, tile = 40, int(os.environ["RT_TEST_H"]), int(os.environ["RT_TEST_TILE"])
    B = int(os.environ.get("RT_TEST_FRAMES", "1"))
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)      # same on every rank
    frame = frames[0]
    fg = FrameGather(W, H, tile, torch.device("cpu"), frames=B) if B > 1 else FrameGather(W, H, tile, torch.device("cpu"))
    rows = rt.part_rows(H, tile, rank, world)
    # what rt_render_device writes: this rank's tiles, compacted in tile order
    n_tiles = (H + tile - 1) // tile
    for b in range(B):
        mine = np.concatenate([frames[b, t * tile:min(H, (t + 1) * tile)] for t in range(rank, n_tiles, world)] or [frame[:0]])
        assert mine.shape[0] == rows
        fg.slices[b, :rows] = torch.from_numpy(mine)
    assert fg.slice.data_ptr() == fg.slices.data_ptr()
    out = fg.run()
    if rank == 0:
        assert out is not None and np.array_equal(out.numpy(), frame if B == 1 else frames), "frame mismatch"
        assert np.array_equal(fg.frame.numpy(), frame)
        # the C-ABI scatter helper must agree with the index-select path
        chk = np.zeros_like(frame)
        for k in range(world):
            rt.part_scatter(fg.gathered[k].numpy()[: rt.part_rows(H, tile, k, world)], chk, tile, k, world)
        assert np.array_equal(chk, frame)
    else:
        assert out is None
    if B > 1:
        # a partial last batch: only the first nb frames of the set were rendered; they travel and are rebuilt alone, the
        # message is nb (not B) frames' slices, and the rest of the frame buffers is left as it was
        nb = B // 2
        if rank == 0:
            fg.frames[nb:] = 9
        out = fg.run(nb)
        assert fg.last_bytes_sent == nb * fg.max_rows * W * 3
        if rank == 0:
            assert tuple(out.shape) == (nb, H, W, 3) and np.array_equal(out.numpy(), frames[:nb]), "partial batch mismatch"
            assert (fg.frames[nb:] == 9).all()
        try:
            fg.run(B + 1)
            raise SystemExit("nb > batch accepted")
        except ValueError:
            pass
    if rank == 0:
        print("OK")
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,height,tile,frames", [(2, 1080 // 8, 8, 1), (3, 70, 16, 1), (2, 9, 4, 1), (2, 70, 8, 4), (3, 33, 4, 8)])
def test_gather_and_deinterleave(tmp_path, world, height, tile, frames):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RT_TEST_H=str(height), RT_TEST_TILE=str(tile), RT_TEST_FRAMES=str(frames), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert p.returncode == 0 and "OK" in p.stdout, p.stdout[-3000:]


def test_bench_never_reports_one_gpu_for_a_multi_gpu_request():
    """`python bench.py --gpus 2` with no launcher and fewer than 2 visible GPUs (none here): non-zero exit and no result
    line -- never a single-rank number under an N > 1 request.  (With the GPUs present it starts the ranks itself:
    tests/test_gpu_comm.py.)"""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a machine with fewer than 2 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "--gpus 2 asked for" in p.stderr and "{" not in p.stdout
