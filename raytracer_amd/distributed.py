"""Multi-GPU frame assembly: one process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI on ROCm, "gloo"
for CPU tests).  The reference shares one `Image` between OpenMP threads (Code/main.cpp:242,279-320); here the frame
is cut into row tiles, tile k belongs to rank k % world, every rank renders its tiles into a compact u8 slice and the
slices are gathered to rank 0 and de-interleaved.  No other collective is needed: pixels are independent."""
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


class FrameGather:
    """Pre-allocated buffers for gathering padded slices [max_rows, W, 3] u8 to rank 0 and rebuilding the frame."""

    def __init__(self, width, height, tile_rows, device):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.width, self.height, self.tile_rows = width, height, tile_rows
        self.max_rows = max_part_rows(height, tile_rows, self.world)
        self.slice = torch.zeros((self.max_rows, width, 3), dtype=torch.uint8, device=device)
        self.gathered, self.gather_buffer, self.frame, self.index = None, None, None, None
        if self.rank == 0:
            self.frame = torch.empty((height, width, 3), dtype=torch.uint8, device=device)
            if self.world > 1:
                # one contiguous buffer, gathered into through per-rank views: de-interleaving reads it in place
                self.gather_buffer = torch.empty((self.world * self.max_rows, width, 3), dtype=torch.uint8, device=device)
                self.gathered = list(self.gather_buffer.view(self.world, self.max_rows, width, 3).unbind(0))
                self.index = torch.from_numpy(deinterleave_index(height, tile_rows, self.world, self.max_rows)).to(device)

    def run(self):
        """Collective: call on every rank after its slice has been written.  Returns the frame on rank 0, else None."""
        if self.world == 1:
            return self.slice[: self.height]
        if self.slice.is_cuda and dist.get_backend() == "gloo":
            # gloo gathers host tensors only (rehearsals of the multi-rank path on a box with fewer GPUs than ranks)
            host = [torch.empty(self.slice.shape, dtype=torch.uint8) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(self.slice.cpu(), host, dst=0)
            if self.rank == 0:
                for view, part in zip(self.gathered, host):
                    view.copy_(part)
        else:
            dist.gather(self.slice, self.gathered, dst=0)
        if self.rank == 0:
            torch.index_select(self.gather_buffer, 0, self.index, out=self.frame)
            return self.frame
        return None
