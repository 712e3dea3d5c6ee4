"""Progressive, resumable accumulation on top of rt_accumulate_device.

The reference renders `samples_per_pixel` samples in one go and stores only tone-mapped bytes (Code/main.cpp:281-322),
so a 1024-spp 4K frame is all or nothing.  Here a frame is built from passes: every pass renders `spp_per_pass`
samples with its own seed into linear radiance (FP64, before tone mapping), the passes are summed, and the sum with its
pass count can be written to disk and picked up again.  The picture after N passes is the reference's estimator with
N * spp_per_pass samples: average first, then tone map and truncate (tracer.h:23-54, 297-308).
"""
import numpy as np

import raytracer_amd as rt


def tonemap(c, mode):
    """tracer.h:23-54 on an array [..., 3] of linear radiance."""
    c = np.asarray(c, dtype=np.float64)
    if mode == rt.RT_TONEMAP_REINHARD:
        return c / (1.0 + c)
    if mode == rt.RT_TONEMAP_ACES:
        return (c * (2.51 * c + 0.03)) / (c * (2.43 * c + 0.59) + 0.14)
    if mode == rt.RT_TONEMAP_FILMIC:
        A, B, Cc, D, E, F = 0.15, 0.50, 0.10, 0.20, 0.02, 0.30

        def f(x):
            return ((x * (A * x + Cc * B) + D * E) / (x * (A * x + B) + D * F)) - E / F
        return f(c) * (1.0 / f(np.float64(11.2)))
    return c


def quantise(c):
    """tracer.h:297-308: clamp to [0, 1] with std::min/max semantics (a NaN channel becomes 255), truncate to 8 bits."""
    c = np.asarray(c, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        clamped = np.where(c < 1.0, c, 1.0)          # std::min(1.0, v): NaN compares false -> 1.0
        clamped = np.where(clamped > 0.0, clamped, 0.0)
        clamped = np.where(np.isnan(c), 1.0, clamped)
    return (255.0 * clamped).astype(np.uint8)


def to_image(linear, tonemap_mode):
    """Linear radiance [H, W, 3] -> the bytes Image::setPixel would hold (main.cpp:308-320)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        return quantise(tonemap(linear, tonemap_mode))


class ProgressiveRender:
    """Sum of linear passes of one scene, kept in HBM (C-ABI rt_accumulate_device): `params.samples_per_pixel` is the
    sample count of ONE pass; pass k is rendered with seed `params.seed + k` (RT_RNG_SEEDED) so that passes are
    independent and a resumed run continues the series.  Nothing is read back until a picture (`image`: resolved on the
    device, 3 bytes per pixel) or the sum itself (`sum`, `linear`, `save`: one copy of 24 bytes per pixel) is asked for."""

    def __init__(self, scene, params):
        import torch
        if params.rng_mode != rt.RT_RNG_SEEDED:
            raise ValueError("progressive accumulation needs the seeded RNG: identical passes add nothing")
        self.scene, self.params = scene, params.copy()
        self.base_seed = int(params.seed)
        self._torch = torch
        self._dsum = torch.zeros((scene.height, scene.width, 3), dtype=torch.float64, device="cuda")
        self.passes = 0

    @property
    def samples(self):
        return self.passes * int(self.params.samples_per_pixel)

    @property
    def sum(self):
        return self._dsum.cpu().numpy()

    def add_passes(self, n=1):
        p = self.params.copy()
        p.seed = self.base_seed
        self.scene.accumulate_device(p, self._dsum.data_ptr(), self.passes, n, stream=self._torch.cuda.current_stream().cuda_stream)
        self.passes += n
        return self

    def linear(self):
        return self.sum / max(1, self.passes)

    def image(self):
        """The picture so far: average, tone map and 8-bit truncation on the device (rt_resolve_sum_device)."""
        out = self._torch.empty((self.scene.height, self.scene.width, 3), dtype=self._torch.uint8, device="cuda")
        rt.resolve_sum_device(self._dsum.data_ptr(), max(1, self.passes), self.scene.width, self.scene.height, int(self.params.tonemap),
                              out.data_ptr(), stream=self._torch.cuda.current_stream().cuda_stream)
        return out.cpu().numpy()

    def save(self, path):
        """Checkpoint: the running sum, the pass count and what the passes were rendered with."""
        s = self.sum
        np.savez_compressed(path, sum=s, passes=self.passes, base_seed=self.base_seed,
                            spp_per_pass=int(self.params.samples_per_pixel), shape=np.array(s.shape))

    def load(self, path):
        """Resume from a checkpoint written by save() for the same scene and per-pass sample count."""
        with np.load(path, allow_pickle=False) as z:
            if tuple(z["shape"]) != tuple(self._dsum.shape) or int(z["spp_per_pass"]) != int(self.params.samples_per_pixel):
                raise ValueError("checkpoint does not belong to this frame (resolution or samples per pass differ)")
            self._dsum.copy_(self._torch.from_numpy(z["sum"].astype(np.float64)))
            self.passes = int(z["passes"])
            self.base_seed = int(z["base_seed"])
        return self
