"""Multi-GPU frame assembly: one process per GPU.  The reference shares one `Image` between OpenMP threads
(Code/main.cpp:242,279-320); here the frame is cut into row tiles, tile k belongs to rank k % world, every rank renders
its tiles into a compact u8 slice and the slices are gathered to rank 0 and de-interleaved.  No other collective is
needed: pixels are independent.

On GPUs the exchange is the library's own (C-ABI rt_gather_frames: grouped ncclSend / ncclRecv over RCCL / xGMI and
the k_deinterleave HIP kernel, raytracer_amd/csrc/comm.hip); `FrameGather` only owns the buffers and calls it, and
`init_comm` hands the RCCL unique id from rank 0 to the others through torch.distributed, which the launcher
(torchrun) has set up anyway.  With the "gloo" backend (CPU tests, rehearsals of the multi-rank control flow on a box
with fewer GPUs than ranks) the same assembly is done with torch.distributed.gather + index_select."""
import numpy as np
import torch
import torch.distributed as dist

import raytracer_amd as rt


def max_part_rows(height, tile_rows, world):
    return max(rt.part_rows(height, tile_rows, k, world) for k in range(world))


def deinterleave_index(height, tile_rows, world, rows_per_slot):
    """index[y] = row of the concatenated [world * rows_per_slot] gather buffer that holds frame row y."""
    src = np.empty(height, dtype=np.int64)
    n_tiles = (height + tile_rows - 1) // tile_rows
    for k in range(world):
        lr = 0
        for t in range(k, n_tiles, world):
            y0 = t * tile_rows
            rows = min(tile_rows, height - y0)
            src[y0:y0 + rows] = k * rows_per_slot + lr + np.arange(rows)
            lr += rows
    return src


def init_comm(device=None):
    """Create the library's RCCL communicator over the ranks of the default torch.distributed group (no-op for one
    rank, or when it exists): rank 0's unique id is broadcast as a byte tensor."""
    if not dist.is_initialized() or dist.get_world_size() == 1 or rt.comm_world() == dist.get_world_size():
        return
    rank, world = dist.get_rank(), dist.get_world_size()
    on_gpu = dist.get_backend() == "nccl"
    uid = torch.zeros(rt.RT_COMM_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        uid = torch.frombuffer(bytearray(rt.comm_unique_id()), dtype=torch.uint8).clone()
    if on_gpu:
        uid = uid.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.broadcast(uid, src=0)
    rt.comm_init(rank, world, bytes(uid.cpu().numpy().tobytes()))


class FrameGather:
    """Pre-allocated buffers for gathering padded slices [max_rows, W, 3] u8 to rank 0 and rebuilding the frame.

    frames > 1: the slices of that many frames ([frames, max_rows, W, 3], frame b rendered into `slices[b]`) travel in ONE
    gather and are de-interleaved by one index_select into `frames` [frames, H, W, 3]: at 8 ranks a rank's share of a
    1080p frame is 0.78 MB and ~0.12 ms of rendering, so a collective per frame is bound by its launch and host cost, not
    by xGMI; one per batch of frames is 1/frames as many, each frames times larger."""

    def __init__(self, width, height, tile_rows, device, frames=1, native=None):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.width, self.height, self.tile_rows, self.batch = width, height, tile_rows, frames
        self.max_rows = max_part_rows(height, tile_rows, self.world)
        self.slices = torch.zeros((frames, self.max_rows, width, 3), dtype=torch.uint8, device=device)
        self.slice = self.slices[0]
        self.gathered, self.gather_buffer, self.frames, self.frame, self.index, self._parts = None, None, None, None, None, None
        self.last_bytes_sent = 0
        # the library's exchange (RCCL + k_deinterleave) whenever its communicator spans this group
        self.native = self.world > 1 and self.slices.is_cuda and rt.comm_world() == self.world and rt.comm_rank() == self.rank
        if native is False:   # the caller found the library's communicator missing on some rank: torch.distributed's gather on all
            self.native = False
        if self.rank == 0 and self.native:
            self.frames = torch.empty((frames, height, width, 3), dtype=torch.uint8, device=device)
            self.frame = self.frames[0]
        elif self.rank == 0:
            self.frames = torch.empty((frames, height, width, 3), dtype=torch.uint8, device=device)
            self.frame = self.frames[0]
            if self.world > 1:
                # one contiguous buffer [world, frames, max_rows, W, 3], gathered into through per-rank views:
                # de-interleaving reads it in place
                self.gather_buffer = torch.empty((self.world * frames * self.max_rows, width, 3), dtype=torch.uint8, device=device)
                self._parts = list(self.gather_buffer.view(self.world, frames, self.max_rows, width, 3).unbind(0))
                self.gathered = [p[0] for p in self._parts]   # rank k's slice of frame 0
                src = deinterleave_index(height, tile_rows, self.world, self.max_rows)   # k * max_rows + local row
                k, lr = src // self.max_rows, src % self.max_rows
                b = np.arange(frames, dtype=np.int64)[:, None]
                flat = k[None, :] * (frames * self.max_rows) + b * self.max_rows + lr[None, :]   # [frames, H]
                self.index = torch.from_numpy(np.ascontiguousarray(flat.reshape(-1))).to(device)

    def run(self, nb=None):
        """Collective: call on every rank after its slices have been written.  nb (default: the whole batch): only the
        first nb frames of the set were rendered -- a partial last batch ships its own bytes, not the set's.  Returns on
        rank 0 the frame (frames == 1) or the nb frames, else None."""
        nb = self.batch if nb is None else int(nb)
        if not 1 <= nb <= self.batch:
            raise ValueError("FrameGather.run: nb must be 1 .. %d" % self.batch)
        self.last_bytes_sent = nb * self.max_rows * self.width * 3 if self.world > 1 else 0   # this rank's message
        if self.world == 1:
            return self.slice[: self.height] if self.batch == 1 else self.slices[:nb, : self.height]
        if self.native:
            rt.gather_frames(self.slices.data_ptr(), nb, self.width, self.height, self.tile_rows,
                             self.frames.data_ptr() if self.rank == 0 else None, stream=torch.cuda.current_stream().cuda_stream)
            if self.rank == 0:
                return self.frame if self.batch == 1 else self.frames[:nb]
            return None
        if self.slices.is_cuda and dist.get_backend() == "gloo":
            # gloo gathers host tensors only (rehearsals of the multi-rank path on a box with fewer GPUs than ranks)
            host = [torch.empty(self.slices[:nb].shape, dtype=torch.uint8) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(self.slices[:nb].cpu(), host, dst=0)
            if self.rank == 0:
                for view, part in zip(self._parts, host):
                    view[:nb].copy_(part)
        elif nb == self.batch:
            dist.gather(self.slices, self._parts, dst=0)
        else:
            parts = [torch.empty_like(self.slices[:nb]) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(self.slices[:nb].contiguous(), parts, dst=0)
            if self.rank == 0:
                for view, part in zip(self._parts, parts):
                    view[:nb].copy_(part)
        if self.rank == 0:
            # the index is frame-major: its first nb * H entries rebuild the first nb frames
            torch.index_select(self.gather_buffer, 0, self.index[: nb * self.height], out=self.frames[:nb].view(nb * self.height, self.width, 3))
            return self.frame if self.batch == 1 else self.frames[:nb]
        return None
