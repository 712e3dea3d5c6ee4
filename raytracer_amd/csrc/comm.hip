// Multi-GPU frame assembly behind the C-ABI (include/rt_mi355x.h: rt_comm_*, rt_gather_frames, rt_deinterleave_device).
//
// The reference shares ONE `Image` between its OpenMP threads (main.cpp:242,279-320).  Here the frame is cut into row
// tiles, tile k belongs to rank k % world (one process per GPU), every rank renders its tiles into a compact u8 slice
// (rt_render_device), and the one exchange step of the path is this file: the slices travel to rank 0 over RCCL/xGMI as
// ONE grouped ncclSend / ncclRecv per rank -- a rank's share of a 1080p frame is 0.78 MB at 8 ranks, so the exchange is
// bound by launch and host cost, not by a link; several frames' slices go in one message -- and k_deinterleave puts the
// rows where Image::m_pixel_data has them (Image.h:46-47).  No reduction, no other collective.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the copy the process already holds if any -- PyTorch ships one):
// single-GPU users of the library never load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "api_internal.h"

namespace {

using rt::api_fail;

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
};
Rccl g_rccl;
std::mutex g_comm_mutex;

// Rank 0 receives into a staging buffer [world][frames][max_rows][W*3] that k_deinterleave then reads.  One buffer PER
// STREAM the caller gathers on: gathers on different streams may be in flight together (operations of one communicator are
// ordered by RCCL, the buffers are not), and a buffer is only ever freed after the whole device has gone idle.
struct Staging {
    unsigned char* p = nullptr;
    size_t bytes = 0;
};
// HIP events around the two halves of a gather on the caller's stream (rt_gather_time): start, after the RCCL group, after
// k_deinterleave.  A fixed pool that wraps around: timings nobody collects are overwritten.
struct GatherEvents {
    hipEvent_t e[3] = {nullptr, nullptr, nullptr};
};
constexpr size_t kMaxGatherEvents = 512;
struct CommState {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    std::map<hipStream_t, Staging> staging;
    std::vector<GatherEvents> events;
    size_t events_used = 0;
    bool loopback = false;              // RT_COMM_LOOPBACK=1: rank 0 also sends its own slices to itself through RCCL (one-GPU tests)
};
CommState g_comm;   // every field is read and written under g_comm_mutex

int load_rccl() {
    if (g_rccl.handle) return RT_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);   // the copy this process already uses (PyTorch's), if any
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return api_fail(RT_ERR_DEVICE, std::string("RCCL is not available: ") + dlerror());
    Rccl r;
    r.handle = h;
#define RT_SYM(field, name)                                                                                  \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                                           \
    if (!r.field) return api_fail(RT_ERR_DEVICE, std::string("librccl.so.1 lacks ") + name);
    RT_SYM(get_unique_id, "ncclGetUniqueId")
    RT_SYM(comm_init_rank, "ncclCommInitRank")
    RT_SYM(comm_destroy, "ncclCommDestroy")
    RT_SYM(error_string, "ncclGetErrorString")
    RT_SYM(group_start, "ncclGroupStart")
    RT_SYM(group_end, "ncclGroupEnd")
    RT_SYM(send, "ncclSend")
    RT_SYM(recv, "ncclRecv")
#undef RT_SYM
    g_rccl = r;
    return RT_OK;
}

#define NCCL_TRY(expr)                                                                                          \
    do {                                                                                                        \
        ncclResult_t r_ = (expr);                                                                               \
        if (r_ != ncclSuccess) return api_fail(RT_ERR_DEVICE, std::string(#expr) + ": " + g_rccl.error_string(r_)); \
    } while (0)
#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) return api_fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------------ k_deinterleave
// One workgroup per (frame row, frame): row y of frame f is row lr = (tile / world) * tile_rows + y % tile_rows of the
// slice of rank k = tile % world (tile = y / tile_rows).  A row is contiguous on both sides, so this is H x frames row
// copies: 16-byte pieces when the row length and all bases allow it, bytes otherwise.  HBM-bound (2 x 6.2 MB per 1080p
// frame, microseconds); it exists so that the assembled frame never leaves the device.
typedef unsigned rt_c4 __attribute__((ext_vector_type(4)));
struct DeinterleaveArgs {
    const unsigned char* own;       // rank 0's own slices [frames][own_frame_stride], or null: rank 0's rows are in `gathered` too
    const unsigned char* gathered;  // [world][rank_stride], inside a rank [frames][frame_stride]
    unsigned char* frames;          // [frames][height][row_bytes]
    size_t own_frame_stride, rank_stride, frame_stride;
    int row_bytes, height, tile_rows, world;
};
template <bool VEC>
__global__ void __launch_bounds__(256) k_deinterleave(const DeinterleaveArgs a) {
    const int y = blockIdx.x, f = blockIdx.y;
    const int tile = y / a.tile_rows;
    const int k = tile % a.world;
    const int lr = (tile / a.world) * a.tile_rows + (y - tile * a.tile_rows);
    const unsigned char* src = (k == 0 && a.own) ? a.own + (size_t)f * a.own_frame_stride
                                                 : a.gathered + (size_t)k * a.rank_stride + (size_t)f * a.frame_stride;
    src += (size_t)lr * a.row_bytes;
    unsigned char* dst = a.frames + ((size_t)f * a.height + y) * (size_t)a.row_bytes;
    if (VEC) {
        const rt_c4* s = reinterpret_cast<const rt_c4*>(src);
        rt_c4* d = reinterpret_cast<rt_c4*>(dst);
        for (int i = threadIdx.x; i < a.row_bytes / 16; i += blockDim.x) d[i] = s[i];
    } else {
        for (int i = threadIdx.x; i < a.row_bytes; i += blockDim.x) dst[i] = src[i];
    }
}

int max_part_rows(int height, int tile_rows, int world) {
    int m = 0;
    for (int k = 0; k < world; ++k) m = rt_part_rows(height, tile_rows, k, world) > m ? rt_part_rows(height, tile_rows, k, world) : m;
    return m;
}

int launch_deinterleave(const DeinterleaveArgs& a, int frames, hipStream_t st) {
    const bool vec = a.row_bytes % 16 == 0 && a.own_frame_stride % 16 == 0 && a.rank_stride % 16 == 0 && a.frame_stride % 16 == 0 &&
                     (reinterpret_cast<uintptr_t>(a.own) | reinterpret_cast<uintptr_t>(a.gathered) | reinterpret_cast<uintptr_t>(a.frames)) % 16 == 0;
    const dim3 grid((unsigned)a.height, (unsigned)frames);
    if (vec) hipLaunchKernelGGL(k_deinterleave<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_deinterleave<false>, grid, dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_comm_unique_id(void* id_out) {
    if (!id_out) return api_fail(RT_ERR_INVALID, "id_out is NULL");
    std::lock_guard<std::mutex> lock(g_comm_mutex);
    int rc = load_rccl();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == RT_COMM_ID_BYTES, "RT_COMM_ID_BYTES");
    ncclUniqueId id;
    NCCL_TRY(g_rccl.get_unique_id(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return RT_OK;
}

int rt_comm_init(int rank, int world, const void* id) {
    if (!rt::api_ready()) return api_fail(RT_ERR_NOT_INIT, "rt_init() has not succeeded in this process");
    if (world < 1 || rank < 0 || rank >= world || !id) return api_fail(RT_ERR_INVALID, "rt_comm_init: bad rank / world / id");
    std::lock_guard<std::mutex> lock(g_comm_mutex);
    if (g_comm.comm) return api_fail(RT_ERR_INVALID, "rt_comm_init: a communicator exists already (rt_comm_destroy first)");
    int rc = load_rccl();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(rt::api_device()));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    NCCL_TRY(g_rccl.comm_init_rank(&comm, world, uid, rank));
    g_comm.comm = comm;
    g_comm.rank = rank;
    g_comm.world = world;
    const char* lb = std::getenv("RT_COMM_LOOPBACK");
    g_comm.loopback = lb && lb[0] == '1';
    return RT_OK;
}

int rt_comm_rank(void) {
    std::lock_guard<std::mutex> lock(g_comm_mutex);
    return g_comm.comm ? g_comm.rank : 0;
}
int rt_comm_world(void) {
    std::lock_guard<std::mutex> lock(g_comm_mutex);
    return g_comm.comm ? g_comm.world : 1;
}

int rt_comm_destroy(void) {
    std::lock_guard<std::mutex> lock(g_comm_mutex);
    if (!g_comm.comm && g_comm.staging.empty() && g_comm.events.empty()) return RT_OK;
    (void)hipSetDevice(rt::api_device());
    (void)hipDeviceSynchronize();
    ncclResult_t r = g_comm.comm ? g_rccl.comm_destroy(g_comm.comm) : ncclSuccess;
    for (auto& kv : g_comm.staging)
        if (kv.second.p) (void)hipFree(kv.second.p);
    for (GatherEvents& g : g_comm.events)
        for (hipEvent_t e : g.e)
            if (e) (void)hipEventDestroy(e);
    g_comm = CommState();
    if (r != ncclSuccess) return api_fail(RT_ERR_DEVICE, std::string("ncclCommDestroy: ") + g_rccl.error_string(r));
    return RT_OK;
}

int rt_deinterleave_device(const void* d_own, const void* d_gathered, void* d_frames, int frames, int width, int height, int tile_rows, int world,
                           void* hip_stream) {
    if (!rt::api_ready()) return api_fail(RT_ERR_NOT_INIT, "rt_init() has not succeeded in this process");
    if (!d_frames || frames < 1 || width <= 0 || height <= 0 || tile_rows <= 0 || world < 1 || (!d_gathered && (world > 1 || !d_own)))
        return api_fail(RT_ERR_INVALID, "rt_deinterleave_device: bad argument");
    const int max_rows = max_part_rows(height, tile_rows, world);
    DeinterleaveArgs a;
    a.own = static_cast<const unsigned char*>(d_own);
    a.gathered = static_cast<const unsigned char*>(d_gathered);
    a.frames = static_cast<unsigned char*>(d_frames);
    a.row_bytes = width * 3;
    a.frame_stride = (size_t)max_rows * a.row_bytes;
    a.own_frame_stride = a.frame_stride;
    a.rank_stride = (size_t)frames * a.frame_stride;
    a.height = height; a.tile_rows = tile_rows; a.world = world;
    return launch_deinterleave(a, frames, static_cast<hipStream_t>(hip_stream));
}

int rt_gather_frames(const void* d_slices, int frames, int width, int height, int tile_rows, void* d_frames, void* hip_stream) {
    if (!rt::api_ready()) return api_fail(RT_ERR_NOT_INIT, "rt_init() has not succeeded in this process");
    if (!d_slices || frames < 1 || width <= 0 || height <= 0 || tile_rows <= 0) return api_fail(RT_ERR_INVALID, "rt_gather_frames: bad argument");
    // The communicator and this stream's staging buffer are looked up -- and the RCCL calls issued -- under the mutex:
    // rt_comm_destroy cannot pull either away in between, and two threads cannot interleave their groups.
    std::lock_guard<std::mutex> lock(g_comm_mutex);
    const int world = g_comm.comm ? g_comm.world : 1, rank = g_comm.comm ? g_comm.rank : 0;
    if (rank == 0 && !d_frames) return api_fail(RT_ERR_INVALID, "rt_gather_frames: rank 0 needs d_frames");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const int max_rows = max_part_rows(height, tile_rows, world);
    const size_t frame_stride = (size_t)max_rows * width * 3, count = (size_t)frames * frame_stride;
    const bool self = g_comm.comm && g_comm.loopback;
    // timing events of this gather (best effort: without them the gather still runs)
    GatherEvents* ev = nullptr;
    if (world > 1 || self) {
        if (g_comm.events.size() < kMaxGatherEvents && g_comm.events_used == g_comm.events.size()) {
            GatherEvents g;
            bool ok = true;
            for (hipEvent_t& e : g.e) ok = ok && hipEventCreate(&e) == hipSuccess;
            if (ok) g_comm.events.push_back(g);
            else for (hipEvent_t e : g.e) if (e) (void)hipEventDestroy(e);
        }
        if (!g_comm.events.empty()) {
            ev = &g_comm.events[g_comm.events_used % g_comm.events.size()];
            ++g_comm.events_used;
            (void)hipEventRecord(ev->e[0], st);
        }
    }
    if (rank != 0) {
        NCCL_TRY(g_rccl.send(d_slices, count, ncclUint8, 0, g_comm.comm, st));
        if (ev) { (void)hipEventRecord(ev->e[1], st); (void)hipEventRecord(ev->e[2], st); }
        return RT_OK;
    }
    unsigned char* staging = nullptr;
    if (world > 1 || self) {
        HIP_TRY(hipSetDevice(rt::api_device()));
        const size_t need = (size_t)world * count;
        Staging& sg = g_comm.staging[st];
        if (sg.bytes < need) {
            // Grows with the first gather of a larger frame or batch on this stream; kept until rt_comm_destroy.  The old buffer
            // may still be read by a de-interleave kernel of this stream, and hipFree of memory in use is undefined: the whole
            // device goes idle first (once per size, not per gather).
            HIP_TRY(hipDeviceSynchronize());
            if (sg.p) (void)hipFree(sg.p);
            sg = Staging();
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&sg.p), need));
            sg.bytes = need;
        }
        staging = sg.p;
        // An error between ncclGroupStart and ncclGroupEnd must not leave the group open for every later call: the group is
        // always closed, and the first error of the three stages is what the caller sees.
        ncclResult_t first = g_rccl.group_start();
        if (first != ncclSuccess) return api_fail(RT_ERR_DEVICE, std::string("ncclGroupStart: ") + g_rccl.error_string(first));
        if (self) first = g_rccl.send(d_slices, count, ncclUint8, 0, g_comm.comm, st);
        for (int k = self ? 0 : 1; k < world && first == ncclSuccess; ++k)
            first = g_rccl.recv(staging + (size_t)k * count, count, ncclUint8, k, g_comm.comm, st);
        const ncclResult_t closed = g_rccl.group_end();
        if (first != ncclSuccess) return api_fail(RT_ERR_DEVICE, std::string("ncclSend / ncclRecv: ") + g_rccl.error_string(first));
        if (closed != ncclSuccess) return api_fail(RT_ERR_DEVICE, std::string("ncclGroupEnd: ") + g_rccl.error_string(closed));
    }
    if (ev) (void)hipEventRecord(ev->e[1], st);
    const int rc = rt_deinterleave_device(self ? nullptr : d_slices, staging, d_frames, frames, width, height, tile_rows, world, hip_stream);
    if (ev) (void)hipEventRecord(ev->e[2], st);
    return rc;
}

int rt_gather_time(double* exchange_ms_out, double* deinterleave_ms_out, uint64_t* gathers_out) {
    if (!exchange_ms_out || !deinterleave_ms_out || !gathers_out) return api_fail(RT_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lock(g_comm_mutex);
    double ex = 0.0, de = 0.0;
    uint64_t n = 0;
    const size_t kept = g_comm.events_used < g_comm.events.size() ? g_comm.events_used : g_comm.events.size();
    for (size_t i = 0; i < kept; ++i) {
        float a = 0.f, b = 0.f;
        hipError_t e = hipEventElapsedTime(&a, g_comm.events[i].e[0], g_comm.events[i].e[1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&b, g_comm.events[i].e[1], g_comm.events[i].e[2]);
        if (e != hipSuccess) return api_fail(RT_ERR_DEVICE, std::string("gather events (synchronise the gather streams first): ") + hipGetErrorString(e));
        ex += a;
        de += b;
        ++n;
    }
    g_comm.events_used = 0;
    *exchange_ms_out = ex;
    *deinterleave_ms_out = de;
    *gathers_out = n;
    return RT_OK;
}

}  // extern "C"
