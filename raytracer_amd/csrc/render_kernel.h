// Host-callable launchers of the render kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <vector>

#include "device_types.h"

namespace rt {

// Megakernel schedule (render_mega.hip): renders this part's rows (params.part_rows rows of params.width pixels,
// compact, tile order) into d_rgb8 with one launch on `stream`.  d_counters: NULL or RT_N_COUNTERS x u64 {rays, box tests,
// primitive tests, traversal ns (wavefront only), nodes fetched, ...} that the kernel adds into.
hipError_t launch_render_mega(const DScene& scene, const DParams& params, uint8_t* d_rgb8, unsigned long long* d_counters,
                              hipStream_t stream);

// Device memory the wavefront schedule keeps between calls: one workspace per HIP stream, process-wide (api.cpp keeps the
// pool), so that queues outlive the scenes rendered through them; freed by rt_release_workspaces / rt_shutdown.
struct WavefrontWorkspace {
    size_t capacity = 0;                 // rays per queue
    std::vector<void*> queues;           // RayRec[capacity] each
    std::vector<void*> hit_buffers;      // HitRec[capacity] each, paired with queues
    std::vector<char> in_use;
    void* bundles = nullptr;             // shadow bundles of one shading launch: capacity x 128 bytes (allocated with the first frame that has shadows)
    double* accum = nullptr;             // FP64 RGB accumulator per pixel of the part
    size_t accum_pixels = 0;
    unsigned* counters = nullptr;        // [2] overflow flag, [4] entries of the queue last summed, [6..7] ray total, [32..] fetch counters
    unsigned* shard_counts = nullptr;    // tails of the queues' shards: one array per ray-tree level + one for the shadow bundles
    unsigned long long* stats = nullptr; // RT_N_COUNTERS entries: [1] box tests, [2] primitive tests, [4] nodes fetched
    unsigned long long host_counts[RT_N_COUNTERS] = {};
    std::vector<hipEvent_t> events;      // start/stop pairs around the k_traverse launches not yet collected
    size_t events_used = 0;
    int compute_units = 256;
    int rays_per_lane = 2;               // traversal grid sizing (RT_RAYS_PER_LANE)
    size_t grow = 1;                     // queue headroom factor, raised once when a frame's ray tree outgrew the queues
    size_t grown_capacity = 0;           // ... and the capacity that was granted then (may exceed the default maximum)
    bool levels_overflowed = false;      // a frame's ray tree outgrew the queues when its levels were enqueued without round trips
    unsigned long long overflowed_scene = 0;   // ... and the scene (DScene::serial) it belonged to: other scenes try again
    // fixed: the workspace lives in memory the caller provided (the library's static scratch array, rt_render_device_noalloc):
    // nothing is allocated, grown, freed or timed with events; a ray tree that needs more queues than there are fails.
    bool fixed = false;
    unsigned* host_readback = nullptr;   // fixed: pinned host words for the queue-size read-backs (else pageable stack words)
    // A frame of many chunks renders them two at a time: the odd chunks go through this second set of queues on a stream of
    // its own, driven by a second host thread, into the same accumulators (see launch_render_wavefront).
    WavefrontWorkspace* partner = nullptr;
    hipStream_t partner_stream = nullptr;
    // History order of the primary-ray queue's chunks (render_wavefront.hip: k_chunk_order): per chunk of 64 rays the most node
    // steps of one of its rays in the last launch over such a queue, and the order (longest first) the next one hands them out in.
    unsigned* chunk_cost = nullptr;
    unsigned* chunk_order = nullptr;
    size_t chunk_capacity = 0;           // chunks both arrays hold
    unsigned long long chunk_key[4] = {0, 0, 0, 0};   // scene, queue size and frame layout the record belongs to
    bool chunk_order_valid = false;
    // Level sizes of the last one-chunk frame of a small scene rendered through this workspace (render_wavefront.hip: k_tail): the
    // next frame of the same shape hands the levels from the first small one on to ONE launch.
    unsigned level_sizes[24] = {};
    unsigned long long level_key[4] = {0, 0, 0, 0};
    bool level_sizes_valid = false;
};
void wavefront_workspace_free(WavefrontWorkspace& ws);

// Wavefront schedule (render_wavefront.hip): same pixels; work is enqueued on `stream`, and the call returns after the
// last queue has been drained (it reads queue sizes back between levels).
// d_linear != NULL: instead of RGB8, write the per-pixel average before tone mapping as 3 doubles per pixel (add_linear:
// add it to what d_linear holds -- progressive accumulation on the device).
hipError_t launch_render_wavefront(const DScene& scene, const DParams& params, uint8_t* d_rgb8, unsigned long long* d_counters,
                                   hipStream_t stream, WavefrontWorkspace& ws, double* d_linear = nullptr, int add_linear = 0);
// FP32 copy of n transforms for the kernel variants with FP32 primitive tests (DScene::xforms32): 16 floats each.
hipError_t launch_make_xforms32(const DXform* xforms, int n, float* out, hipStream_t stream);
// Sum of `passes` linear pass averages (interleaved RGB doubles) -> average, tone map, 8 bits (main.cpp:308-320).
hipError_t launch_resolve_sum(const double* d_sum, unsigned long long passes, size_t npix, int tonemap, uint8_t* d_rgb8, hipStream_t stream);

// Device time of the k_traverse launches enqueued through `ws` since the last collection (their stream must have been
// synchronised: hipErrorNotReady otherwise), and how many launches that was.
hipError_t wavefront_traverse_time(WavefrontWorkspace& ws, double* ms_out, unsigned* launches_out);

// Allocate (or grow to) `n_queues` queues of `rays_per_queue` rays for this workspace and touch every page, so that the
// first frame rendered through it pays neither the allocation nor the first touch (rt_reserve).
hipError_t wavefront_reserve(WavefrontWorkspace& ws, size_t rays_per_queue, int n_queues, hipStream_t stream);

// The no-allocation variant (rt_render_device_noalloc): the workspace is carved out of a file-scope __device__ array of
// kStaticScratchBytes in render_wavefront.hip.  wavefront_static_setup: once per device, with that device current
// (rt_init): resolves the array's device address and pins the read-back words.  launch_render_wavefront_static: same
// pixels as launch_render_wavefront through queues inside that array; hipErrorInvalidValue when the part is larger than
// kStaticMaxPixels / kStaticMaxWidth, hipErrorOutOfMemory when the ray tree needs more queues than fit.
constexpr size_t kStaticScratchBytes = size_t(64) << 20;
constexpr int kStaticMaxPixels = 1 << 18;   // pixels of the rendered part (FP64 accumulators: 6 MiB of the array)
constexpr int kStaticMaxWidth = 1024;       // an 8-row band of primary rays must fit the smallest queue
hipError_t wavefront_static_setup(int device);
hipError_t launch_render_wavefront_static(const DScene& scene, const DParams& params, uint8_t* d_rgb8, hipStream_t stream,
                                          int device);

}  // namespace rt
