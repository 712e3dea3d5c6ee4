// C-ABI of librt_mi355x.so (include/rt_mi355x.h): scene upload, render launch, and the host-side pieces kept
// from the reference (config.json, scene files, PPM/PFM).  No CPU rendering path exists in this library: every
// rt_render* call runs the HIP kernel of render_kernel.hip or fails with RT_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "api_internal.h"
#include "device_types.h"
#include "bvh_build.h"
#include "host_bvh.h"
#include "host_config.h"
#include "host_image.h"
#include "host_math.h"
#include "host_scene.h"
#include "render_kernel.h"

using namespace rt;

namespace {

thread_local std::string g_error;
int fail(int code, const std::string& msg) {
    g_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct DeviceState {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // rt_render_frames: device -> host copies of one batch while the next one renders
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t rendered[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr};   // rt_render_frames: per slice set
    uint64_t* d_counters = nullptr;  // RT_N_COUNTERS x u64
    // Pinned host staging of rt_render_frames (hipHostMalloc): a device -> host copy into it is one DMA at the link's rate and
    // truly asynchronous, which a copy into the caller's pageable buffer is not (the runtime stages it and the call blocks);
    // the host thread moves a batch on into the caller's buffer while the GPU renders the next one.
    uint8_t* staging[2] = {nullptr, nullptr};
    size_t staging_bytes = 0;
};
DeviceState g_dev;
// rt_render, rt_render_frames and rt_render_gathered work on the library's own stream, events, counters and staging buffers: they
// are synchronous calls and take this lock for their duration, so that threads rendering different scenes into host memory
// queue up instead of sharing that state (the *_device entry points on streams of the caller's run concurrently).
std::recursive_mutex g_host_render_mutex;

// The caller's frame buffer is ordinary (pageable) host memory: `Image::m_pixel_data` is a std::vector (Image.h:46-47).  A
// device -> host copy into pageable memory goes through the runtime's own staging buffers at a fraction of the link's rate
// (6.2 MB of a 1080p frame: ~0.5 ms); into memory that is registered with the driver it is one DMA (~0.15 ms).  By default the
// library copies through pinned staging buffers of its own (DeviceState::staging) and never touches the caller's pages.
// OPT-IN, RT_HOST_REGISTER=second|always: the buffers rt_render / rt_render_frames are handed are registered (on second / first
// sight) and stay registered -- a render loop passes the same buffer again and again -- up to kMaxPins of them (the least
// recently used one is unregistered), until rt_shutdown.  The contract that comes with it: a buffer handed to these calls must
// not be freed while the library is initialised -- a registration pins the PHYSICAL pages behind the range, and a new allocation
// that lands on the same addresses would not be what the DMA writes to.  That is why it is not the default.  Should a copy into a
// cached range fail, the range is dropped and the copy repeated the pageable way.
struct HostPin {
    uintptr_t base = 0;
    size_t bytes = 0;
    unsigned long long used = 0;
    bool ours = true;   // false: the caller had registered it itself (left alone)
};
constexpr size_t kMaxPins = 8, kMinPinBytes = size_t(256) << 10;
std::vector<HostPin> g_pins;
std::vector<HostPin> g_seen;   // ranges seen once (nothing is registered for these yet)
std::mutex g_pin_mutex;
unsigned long long g_pin_clock = 0;

void unpin_all() {
    std::lock_guard<std::mutex> lock(g_pin_mutex);
    for (const HostPin& p : g_pins)
        if (p.ours) (void)hipHostUnregister(reinterpret_cast<void*>(p.base));
    g_pins.clear();
    g_seen.clear();
}
void unpin(void* host) {
    std::lock_guard<std::mutex> lock(g_pin_mutex);
    const uintptr_t a = reinterpret_cast<uintptr_t>(host);
    for (size_t i = 0; i < g_pins.size(); ++i)
        if (a >= g_pins[i].base && a < g_pins[i].base + g_pins[i].bytes) {
            if (g_pins[i].ours) (void)hipHostUnregister(reinterpret_cast<void*>(g_pins[i].base));
            g_pins.erase(g_pins.begin() + static_cast<long>(i));
            return;
        }
}
// true: [host, host + bytes) is registered with the driver now.  Off unless RT_HOST_REGISTER asks for it (see above): `second`
// registers a buffer the SECOND time it is seen (registering costs about as much as one pageable copy, so a buffer that is used
// once is better off without), `always` at first sight.
bool pin_host(void* host, size_t bytes) {
    static const int mode = [] {
        const char* e = std::getenv("RT_HOST_REGISTER");
        if (!e || std::getenv("RT_NO_HOST_REGISTER")) return 0;
        return std::string(e) == "always" ? 2 : std::string(e) == "second" ? 1 : 0;
    }();
    if (mode == 0 || !host || bytes < kMinPinBytes) return false;
    std::lock_guard<std::mutex> lock(g_pin_mutex);
    const uintptr_t a = reinterpret_cast<uintptr_t>(host);
    for (HostPin& p : g_pins)
        if (a >= p.base && a + bytes <= p.base + p.bytes) { p.used = ++g_pin_clock; return true; }
    if (mode == 1) {
        bool seen = false;
        for (size_t i = 0; i < g_seen.size(); ++i)
            if (g_seen[i].base == a && g_seen[i].bytes == bytes) { seen = true; g_seen.erase(g_seen.begin() + static_cast<long>(i)); break; }
        if (!seen) {
            if (g_seen.size() >= kMaxPins) g_seen.erase(g_seen.begin());
            g_seen.push_back(HostPin{a, bytes, 0, false});
            return false;
        }
    }
    // a range that overlaps a cached one without lying inside it (the caller's buffer grew or moved a little): drop the old one
    for (size_t i = 0; i < g_pins.size();) {
        if (a < g_pins[i].base + g_pins[i].bytes && g_pins[i].base < a + bytes) {
            if (g_pins[i].ours) (void)hipHostUnregister(reinterpret_cast<void*>(g_pins[i].base));
            g_pins.erase(g_pins.begin() + static_cast<long>(i));
        } else {
            ++i;
        }
    }
    if (g_pins.size() >= kMaxPins) {
        size_t lru = 0;
        for (size_t i = 1; i < g_pins.size(); ++i) if (g_pins[i].used < g_pins[lru].used) lru = i;
        if (g_pins[lru].ours) (void)hipHostUnregister(reinterpret_cast<void*>(g_pins[lru].base));
        g_pins.erase(g_pins.begin() + static_cast<long>(lru));
    }
    const hipError_t e = hipHostRegister(host, bytes, hipHostRegisterDefault);
    if (e == hipSuccess || e == hipErrorHostMemoryAlreadyRegistered) {
        if (e != hipSuccess) (void)hipGetLastError();
        g_pins.push_back(HostPin{a, bytes, ++g_pin_clock, e == hipSuccess});
        return true;
    }
    (void)hipGetLastError();   // cannot be registered (e.g. a read-only mapping): the pageable copy still works
    return false;
}
// memcpy of a batch of frames from the pinned staging buffer into the caller's buffer: one thread moves ~10 GB/s, which would bound
// rt_render_frames at ~0.6 ms per 1080p frame; pieces of 16 MB and more go to up to four threads.
void host_copy(uint8_t* dst, const uint8_t* src, size_t bytes) {
    const size_t threads = std::min<size_t>(4, bytes / (size_t(16) << 20));
    if (threads < 2) { std::memcpy(dst, src, bytes); return; }
    std::vector<std::thread> pool;
    const size_t piece = (bytes / threads + 4095) & ~size_t(4095);
    try {
        for (size_t t = 1; t < threads; ++t) {
            const size_t b = t * piece, e = std::min(bytes, b + piece);
            if (b < e) pool.emplace_back([=] { std::memcpy(dst + b, src + b, e - b); });
        }
    } catch (...) {   // no thread to be had: this one copies the rest
        size_t done = (pool.size() + 1) * piece;
        if (done < bytes) std::memcpy(dst + done, src + done, bytes - done);
    }
    std::memcpy(dst, src, std::min(bytes, piece));
    for (std::thread& th : pool) th.join();
}

// Device -> host copy of a frame (or several) into the caller's buffer on `st`, through the registration cache.
hipError_t copy_out(void* host, const void* dev, size_t bytes, hipStream_t st) {
    const bool pinned = pin_host(host, bytes);
    hipError_t e = hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess && pinned) {   // a stale range: forget it and copy the ordinary way
        (void)hipGetLastError();
        unpin(host);
        e = hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st);
    }
    return e;
}

// Workspaces of the wavefront schedule (ray queues, accumulators, counters): one per HIP stream, process-wide, so that
// they outlive the scenes rendered through them -- a service that opens and closes scenes keeps its queues, and the
// first frame of a new scene does not pay for allocating and first touching gigabytes (rt_reserve does, when asked).
// Freed by rt_release_workspaces / rt_shutdown.  The mutex guards the map, not the workspaces: one render at a time per stream.
std::map<hipStream_t, rt::WavefrontWorkspace> g_wf;
std::mutex g_wf_mutex;
unsigned long long g_scene_serial = 0;

rt::WavefrontWorkspace* workspace_of(hipStream_t st, bool create) {
    std::lock_guard<std::mutex> lock(g_wf_mutex);
    if (create) return &g_wf[st];   // std::map: references stay valid when other streams are added
    auto it = g_wf.find(st);
    return it == g_wf.end() ? nullptr : &it->second;
}

// std::vector without the zero fill of resize(): the big per-primitive arrays are written in full right after.
template <typename T>
struct NoInit : std::allocator<T> {
    template <typename U> struct rebind { using other = NoInit<U>; };
    NoInit() = default;
    template <typename U> NoInit(const NoInit<U>&) {}
    template <typename U> void construct(U* p) { ::new (static_cast<void*>(p)) U; }
    template <typename U, typename... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
template <typename T> using RawVector = std::vector<T, NoInit<T>>;

template <typename T, typename A>
int upload(const std::vector<T, A>& v, T** out) {
    *out = nullptr;
    if (v.empty()) return RT_OK;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(out), v.size() * sizeof(T)));
    HIP_TRY(hipMemcpy(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

}  // namespace

struct rt_scene_s {
    HostBvh bvh;                      // reference topology: fixes the primitive (visit) order
    HostBvh trav;                     // what the device traverses (host_bvh.h: build_traversal_bvh)
    int n_shapes = 0;
    int32_t n_nodes4 = 0, depth4 = 0;   // the 4-wide device image of trav
    int builder = RT_BVH_SAH;         // what built the device image (RT_BVH_*)
    double node_pad = 0;              // outward pad of its boxes
    double build_ms = 0;              // RT_BVH_LBVH: device time of the build
    bool order_on_device = false;     // the primitive order came from bvh_order.hip
    double order_ms = 0;
    int width = 0, height = 0;
    DScene d{};                       // device view (pointers are device pointers)
    std::vector<void*> allocations;   // everything to hipFree
    uint8_t* d_frame = nullptr;       // scratch framebuffer for rt_render
    size_t d_frame_bytes = 0;
    uint8_t* d_batch[2] = {nullptr, nullptr};   // rt_render_frames: two sets of frame slices that alternate
    size_t d_batch_bytes = 0;
    bool on_device = false;
};
struct rt_config_s {
    ConfigData cfg;
};
struct rt_scene_file_s {
    std::unique_ptr<SceneFile> scene;
};

namespace rt {
int api_fail(int code, const std::string& msg) { return fail(code, msg); }
bool api_ready() { return g_dev.ready; }
int api_device() { return g_dev.device; }
}  // namespace rt

static int validate_desc(const RtSceneDesc* d) {
    if (!d) return fail(RT_ERR_INVALID, "scene description is NULL");
    if (d->camera.resolution_x <= 0 || d->camera.resolution_y <= 0)
        return fail(RT_ERR_INVALID, "Image width and height must be positive.");  // Image.cpp:75-77
    if (d->n_shapes < 0 || d->n_lights < 0 || d->n_materials < 0 || d->n_textures < 0) return fail(RT_ERR_INVALID, "negative count");
    if ((d->n_shapes && !d->shapes) || (d->n_lights && !d->lights) || (d->n_materials && !d->materials) || (d->n_textures && !d->textures))
        return fail(RT_ERR_INVALID, "array pointer is NULL");
    for (int i = 0; i < d->n_shapes; ++i) {
        const RtShape& s = d->shapes[i];
        if (s.type < RT_SHAPE_SPHERE || s.type > RT_SHAPE_COMPLEX_PLANE) return fail(RT_ERR_INVALID, "shape " + std::to_string(i) + ": unknown type");
        if (s.material < 0 || s.material >= d->n_materials) return fail(RT_ERR_INVALID, "shape " + std::to_string(i) + ": material index out of range");
        // The device keeps rows 0..2 and element [3][3] of the inverse: shapes are affine (scene.cpp:285 only composes T,R,S).
        if (s.type != RT_SHAPE_PLANE && (s.inv_transform[12] != 0.0 || s.inv_transform[13] != 0.0 || s.inv_transform[14] != 0.0))
            return fail(RT_ERR_INVALID, "shape " + std::to_string(i) + ": projective transforms are not supported");
    }
    for (int i = 0; i < d->n_materials; ++i) {
        const RtMaterial& m = d->materials[i];
        if (m.texture >= d->n_textures || m.bump_map >= d->n_textures) return fail(RT_ERR_INVALID, "material " + std::to_string(i) + ": texture index out of range");
    }
    for (int i = 0; i < d->n_textures; ++i)
        if (d->textures[i].width <= 0 || d->textures[i].height <= 0 || !d->textures[i].rgb) return fail(RT_ERR_INVALID, "texture " + std::to_string(i) + " is empty");
    if (d->env.present && d->env.rgb && (d->env.width <= 0 || d->env.height <= 0)) return fail(RT_ERR_INVALID, "environment map has no size");
    return RT_OK;
}

// Host-side derivation of the device arrays from the description + hierarchy.
struct HostArrays {
    RawVector<DPrim> prims;
    RawVector<int32_t> file_order;
    RawVector<DXform> xforms;
    RawVector<DQuad> quads;
    std::vector<DMaterial> materials;
    std::vector<DLight> lights;
    int32_t features = 0;
    int32_t spawning_materials = 0;
    int32_t transparent_materials = 0;
    int32_t area_lights = 0;
    DCamera cam{};
};

static HostArrays derive_arrays(const RtSceneDesc& d, const HostBvh& bvh) {
    HostArrays h;
    h.prims.resize(d.n_shapes);
    h.file_order.resize(d.n_shapes);
    // positions in xforms / quads: a running count over the primitives in order; the records are then filled on host threads
    RawVector<int32_t> data_index(d.n_shapes);
    int32_t n_xforms = 0, n_quads = 0;
    {
        RawVector<int32_t> types(d.n_shapes);   // gathered on threads (the shape records are 400 bytes apart), counted in order
        parallel_ranges(static_cast<size_t>(d.n_shapes), [&](size_t begin, size_t end) {
            for (size_t k = begin; k < end; ++k) types[k] = d.shapes[bvh.order[k]].type;
        });
        for (int k = 0; k < d.n_shapes; ++k) {
            const int32_t type = types[k];
            if (type == RT_SHAPE_PLANE) {
                data_index[k] = n_quads++;
                h.features |= RT_FEAT_QUAD;
            } else {
                data_index[k] = n_xforms++;
                h.features |= (type == RT_SHAPE_SPHERE) ? RT_FEAT_SPHERE : (type == RT_SHAPE_CUBE) ? RT_FEAT_CUBE : RT_FEAT_COMPLEX;
            }
        }
    }
    h.xforms.resize(n_xforms);
    h.quads.resize(n_quads);
    parallel_ranges(static_cast<size_t>(d.n_shapes), [&](size_t begin, size_t end) {
        for (size_t k = begin; k < end; ++k) {
            int fi = bvh.order[k];
            const RtShape& s = d.shapes[fi];
            DPrim& p = h.prims[k];
            p.type = s.type;
            p.material = s.material;
            p.file_index = fi;
            p.data = data_index[k];
            h.file_order[fi] = static_cast<int32_t>(k);
            if (s.type == RT_SHAPE_PLANE) {
                // plane.cpp:49-74
                DQuad q{};
                V3 c0(s.corners), c1(s.corners + 3), c2(s.corners + 6), c3(s.corners + 9);
                V3 e1 = c1 - c0, e2 = c2 - c0;
                store(q.v0, c0); store(q.e1, e1); store(q.e2, e2);
                store(q.w0, c1); store(q.f1, c3 - c1); store(q.f2, c2 - c1);
                store(q.n, normalize(cross(e1, e2)));
                std::memcpy(q.vel, s.velocity, sizeof(q.vel));
                h.quads[p.data] = q;
            } else {
                DXform x{};
                std::memcpy(x.inv, s.inv_transform, 12 * sizeof(double));
                x.inv33 = s.inv_transform[15];
                std::memcpy(x.vel, s.velocity, sizeof(x.vel));
                x.disp = s.max_displacement;
                h.xforms[p.data] = x;
            }
        }
    });
    h.materials.resize(d.n_materials);
    for (int i = 0; i < d.n_materials; ++i) {
        const RtMaterial& m = d.materials[i];
        DMaterial& o = h.materials[i];
        std::memcpy(o.ambient, m.ambient, 24);
        std::memcpy(o.diffuse, m.diffuse, 24);
        std::memcpy(o.specular, m.specular, 24);
        o.shininess = m.shininess;
        o.reflectivity = m.reflectivity;
        o.transparency = m.transparency;
        o.ior = m.refractive_index;
        o.texture = m.texture < 0 ? -1 : m.texture;
        o.bump = m.bump_map < 0 ? -1 : m.bump_map;
        o.is_metal = m.is_metal ? 1 : 0;
        o.pad = 0;
        if (o.texture >= 0 || o.bump >= 0) h.features |= RT_FEAT_TEXTURE;
        if (!(m.reflectivity <= 0.0) || !(m.transparency <= 0.0)) h.spawning_materials = 1;   // also NaN
        if (!(m.transparency <= 0.0)) h.transparent_materials = 1;
    }
    h.lights.resize(d.n_lights);
    for (int i = 0; i < d.n_lights; ++i) {
        std::memcpy(h.lights[i].pos, d.lights[i].position, 24);
        std::memcpy(h.lights[i].intensity, d.lights[i].intensity, 24);
        h.lights[i].radius = d.lights[i].radius;
        if (d.lights[i].radius != 0.0) h.area_lights = 1;
    }
    if (d.env.present) h.features |= RT_FEAT_ENV;
    // camera.cpp:27-86
    const RtCamera& c = d.camera;
    DCamera& cam = h.cam;
    std::memcpy(cam.location, c.location, 24);
    V3 w = normalize(V3(c.gaze_direction));
    V3 u = normalize(cross(V3(c.up_vector), w));
    V3 v = cross(u, w);
    store(cam.u, u); store(cam.v, v); store(cam.w, w);
    cam.focal_length = c.focal_length;
    cam.sensor_width = c.sensor_width;
    double aspect = static_cast<double>(c.resolution_x) / static_cast<double>(c.resolution_y);
    cam.sensor_height = c.sensor_width / aspect;
    double focal_m = c.focal_length / 1000.0;
    cam.aperture_radius = (c.f_stop > 0) ? focal_m / (2.0 * c.f_stop) : 0.0;
    cam.focal_distance = c.focal_distance;
    cam.res_x = c.resolution_x;
    cam.res_y = c.resolution_y;
    return h;
}

// RT_BUILD_TIMING=1: the phases of scene creation on stderr.
struct BuildClock {
    bool on = std::getenv("RT_BUILD_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[rt build] %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

// Builder of the hierarchy the device traverses: rt_set_bvh_builder(), else RT_BVH_BUILDER = sah | lbvh | median, else SAH.
constexpr int kSmallScenePrimsHost = 16;     // == kSmallScenePrims of device_rt.h
constexpr int kDeviceOrderMinShapes = 65536;   // below this the host passes (order, SAH tree) take a few milliseconds
static int g_builder = -1;
static int chosen_builder(int n_shapes = 0) {
    if (g_builder >= 0) return g_builder;
    const char* e = std::getenv("RT_BVH_BUILDER");
    if (e && std::string(e) == "median") return RT_BVH_MEDIAN;
    if (e && std::string(e) == "lbvh") return RT_BVH_LBVH;
    if (e && std::string(e) == "sah_gpu") return RT_BVH_SAH_GPU;
    if (e && std::string(e) == "sah") return RT_BVH_SAH;
    // default: the binned-SAH tree -- built on the device for large scenes (the same tree, bvh_sah.hip), else on host threads
    return n_shapes >= kDeviceOrderMinShapes ? RT_BVH_SAH_GPU : RT_BVH_SAH;
}
static double max_camera_coordinate(const RtSceneDesc& desc) {
    double m = 0.0;
    for (int a = 0; a < 3; ++a) m = std::max(m, std::fabs(desc.camera.location[a]));
    return m;
}

// The hierarchy the device traverses, from the reference-topology tree that fixes the primitive order.
static HostBvh traversal_tree(const RtSceneDesc& desc, const HostBvh& ref) {
    if (chosen_builder() == RT_BVH_MEDIAN) return ref;   // traverse the reference topology itself (A/B runs)
    return build_traversal_bvh(traversal_boxes(desc, ref.order), kMaxBinaryDepth);
}
static Bvh4 device_image(const RtSceneDesc& desc, const HostBvh& trav) {
    return to_node4(trav, max_camera_coordinate(desc));
}

extern "C" {

const char* rt_version(void) { return "rt_mi355x 0.1 (gfx950)"; }
const char* rt_last_error(void) { return g_error.c_str(); }
void rt_free(void* p) { std::free(p); }

int rt_init(int device_id) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(RT_ERR_DEVICE, std::string("no HIP device available (") + hipGetErrorString(e) + "); this library has no CPU fallback");
    if (device_id >= count) return fail(RT_ERR_INVALID, "device_id out of range");
    if (device_id >= 0) HIP_TRY(hipSetDevice(device_id));
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (g_dev.ready && g_dev.device == dev) return RT_OK;
    if (g_dev.ready) rt_shutdown();
    g_dev.device = dev;
    HIP_TRY(hipStreamCreateWithFlags(&g_dev.stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&g_dev.copy_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&g_dev.ev0));
    HIP_TRY(hipEventCreate(&g_dev.ev1));
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipEventCreateWithFlags(&g_dev.rendered[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g_dev.copied[i], hipEventDisableTiming));
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g_dev.d_counters), RT_N_COUNTERS * sizeof(uint64_t)));
    HIP_TRY(wavefront_static_setup(dev));   // device address of the static scratch array of rt_render_device_noalloc
    g_dev.ready = true;
    return RT_OK;
}

int rt_release_workspaces(void) {
    if (!g_dev.ready) return RT_OK;
    HIP_TRY(hipSetDevice(g_dev.device));
    HIP_TRY(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lock(g_wf_mutex);
    for (auto& kv : g_wf) wavefront_workspace_free(kv.second);
    g_wf.clear();
    return RT_OK;
}

int rt_reserve(void* hip_stream, uint64_t rays_per_queue, int n_queues) {
    if (!g_dev.ready) return fail(RT_ERR_NOT_INIT, "rt_init() has not succeeded in this process");
    if (rays_per_queue == 0 || rays_per_queue > (uint64_t(1) << 31) || n_queues < 1 || n_queues > 64)
        return fail(RT_ERR_INVALID, "rt_reserve: rays_per_queue must be 1 .. 2^31 and n_queues 1 .. 64");
    HIP_TRY(hipSetDevice(g_dev.device));
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    hipError_t e = wavefront_reserve(*workspace_of(st, true), static_cast<size_t>(rays_per_queue), n_queues, st);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? RT_ERR_LIMIT : RT_ERR_DEVICE, std::string("rt_reserve: ") + hipGetErrorString(e));
    return RT_OK;
}

void rt_shutdown(void) {
    if (!g_dev.ready) return;
    (void)rt_release_workspaces();
    unpin_all();
    if (g_dev.stream) (void)hipStreamDestroy(g_dev.stream);
    if (g_dev.copy_stream) (void)hipStreamDestroy(g_dev.copy_stream);
    if (g_dev.ev0) (void)hipEventDestroy(g_dev.ev0);
    if (g_dev.ev1) (void)hipEventDestroy(g_dev.ev1);
    for (int i = 0; i < 2; ++i) {
        if (g_dev.rendered[i]) (void)hipEventDestroy(g_dev.rendered[i]);
        if (g_dev.copied[i]) (void)hipEventDestroy(g_dev.copied[i]);
        if (g_dev.staging[i]) (void)hipHostFree(g_dev.staging[i]);
    }
    if (g_dev.d_counters) (void)hipFree(g_dev.d_counters);
    g_dev = DeviceState();
}

int rt_bvh_build_host(const RtSceneDesc* desc, RtBvhInfo* info, int32_t* order_out) {
    int rc = validate_desc(desc);
    if (rc) return rc;
    HostBvh b = build_bvh(*desc);
    if (info) {
        info->n_nodes = static_cast<int32_t>(b.nodes.size());
        info->n_prims = desc->n_shapes;
        info->max_depth = b.max_depth;
        info->node_bytes = kNodeBytes;
        store(info->root_min, b.root_box.lo);
        store(info->root_max, b.root_box.hi);
    }
    if (order_out)
        for (int i = 0; i < desc->n_shapes; ++i) order_out[i] = b.order[i];
    return RT_OK;
}

int rt_bvh_device_image(const RtSceneDesc* desc, void* nodes_out, int32_t max_nodes, RtBvhInfo* info_out, int32_t* root_ref_out,
                        double* pad_out, double* prim_boxes_out) {
    int rc = validate_desc(desc);
    if (rc) return rc;
    BuildClock clock;
    HostBvh ref = build_bvh(*desc);
    clock.lap("reference-order tree");
    HostBvh trav = traversal_tree(*desc, ref);
    clock.lap("traversal tree");
    Bvh4 b4 = device_image(*desc, trav);
    clock.lap("4-wide image");
    if (info_out) {
        info_out->n_nodes = static_cast<int32_t>(b4.nodes.size());
        info_out->n_prims = desc->n_shapes;
        info_out->max_depth = b4.max_depth;
        info_out->node_bytes = kNodeBytes;
        for (int a = 0; a < 3; ++a) { info_out->root_min[a] = b4.root_lo[a]; info_out->root_max[a] = b4.root_hi[a]; }
    }
    if (root_ref_out) *root_ref_out = b4.root_ref;
    if (pad_out) *pad_out = b4.pad;
    if (nodes_out) {
        if (static_cast<size_t>(std::max(max_nodes, 0)) < b4.nodes.size()) return fail(RT_ERR_LIMIT, "nodes_out is too small");
        if (!b4.nodes.empty()) std::memcpy(nodes_out, b4.nodes.data(), b4.nodes.size() * sizeof(DNode4));
    }
    if (prim_boxes_out) {
        const std::vector<Box> boxes = traversal_boxes(*desc, ref.order);   // what the hierarchy was built over
        for (int k = 0; k < desc->n_shapes; ++k) {
            store(prim_boxes_out + 6 * k, boxes[k].lo);
            store(prim_boxes_out + 6 * k + 3, boxes[k].hi);
        }
    }
    return RT_OK;
}

int rt_scene_create(const RtSceneDesc* desc, rt_scene_t* out) {
    if (!out) return fail(RT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int rc = validate_desc(desc);
    if (rc) return rc;
    if (!g_dev.ready) return fail(RT_ERR_NOT_INIT, "rt_init() has not succeeded in this process");
    HIP_TRY(hipSetDevice(g_dev.device));

    auto sc = std::make_unique<rt_scene_s>();
    sc->n_shapes = desc->n_shapes;
    sc->width = desc->camera.resolution_x;
    sc->height = desc->camera.resolution_y;
    BuildClock clock;
    // The primitive order (the reference's visit order): for large scenes on the GPU (bvh_order.hip) unless equal box
    // centres meet at a split, else -- and always for the reference's own topology -- by the host pass.
    if (desc->n_shapes >= kDeviceOrderMinShapes && chosen_builder() != RT_BVH_MEDIAN && std::getenv("RT_ORDER_ON_HOST") == nullptr) {
        std::vector<Box> boxes = shape_boxes(*desc, nullptr);
        clock.lap("shape boxes");
        float ms = 0.f;
        hipError_t herr = hipSuccess;
        const OrderStatus st = reference_order_device(boxes, g_dev.stream, &sc->bvh.order, &ms, &herr);
        clock.lap("device order pass (with upload)");
        if (st == OrderStatus::device_error) return fail(RT_ERR_DEVICE, std::string("device order pass: ") + hipGetErrorString(herr));
        if (st == OrderStatus::ok) {
            sc->order_on_device = true;
            sc->order_ms = ms;
            sc->bvh.root_box = union_of(boxes);
            sc->bvh.shape_boxes = std::move(boxes);
            sc->bvh.root_ref = 0;
        }
    }
    if (!sc->order_on_device) {
        sc->bvh = build_bvh(*desc);
        clock.lap("reference-order tree");
    }
    // The hierarchy the device traverses: built on the GPU (RT_BVH_LBVH) or on host threads.
    DNode4* d_nodes = nullptr;
    Bvh4 b32;
    sc->builder = chosen_builder(desc->n_shapes);
    if ((sc->builder == RT_BVH_LBVH || sc->builder == RT_BVH_SAH_GPU) && desc->n_shapes >= 2) {
        const std::vector<Box> boxes = traversal_boxes(*desc, sc->bvh.order);
        const Box tight_root = union_of(boxes);
        clock.lap("primitive boxes");
        DeviceBvh4 dev4;
        hipError_t herr = hipSuccess;
        const LbvhStatus st = build_lbvh_device(boxes, tight_root, max_camera_coordinate(*desc), g_dev.stream, &dev4, &herr,
                                                sc->builder == RT_BVH_SAH_GPU);
        clock.lap("device hierarchy (with upload)");
        if (st == LbvhStatus::device_error) return fail(RT_ERR_DEVICE, std::string("device BVH build: ") + hipGetErrorString(herr));
        if (st == LbvhStatus::ok) {
            d_nodes = dev4.nodes;
            sc->allocations.push_back(d_nodes);
            b32.root_ref = dev4.root_ref;
            b32.max_depth = dev4.max_depth;
            b32.pad = dev4.pad;
            for (int a = 0; a < 3; ++a) { b32.root_lo[a] = dev4.root_lo[a]; b32.root_hi[a] = dev4.root_hi[a]; }
            sc->n_nodes4 = dev4.n_nodes;
            sc->build_ms = dev4.build_ms;
            sc->trav.root_box = tight_root;
            sc->trav.root_ref = 0;
        } else {
            sc->builder = RT_BVH_SAH;   // deeper than the traversal stack: the host builder bounds its depth
        }
    } else if (sc->builder == RT_BVH_LBVH || sc->builder == RT_BVH_SAH_GPU) {
        sc->builder = RT_BVH_SAH;       // fewer than two shapes: nothing to build
    }
    if (!d_nodes) {
        sc->trav = traversal_tree(*desc, sc->bvh);
        clock.lap("traversal tree");
        // A 4-wide node defers at most three children per level, and the collapsed tree is never deeper than the binary one.
        if (3 * sc->trav.max_depth > kTravStackDepth)
            return fail(RT_ERR_LIMIT, "BVH depth " + std::to_string(sc->trav.max_depth) + " exceeds the traversal stack (" + std::to_string(kTravStackDepth) + " entries)");
        b32 = device_image(*desc, sc->trav);
        clock.lap("4-wide image");
        sc->n_nodes4 = static_cast<int32_t>(b32.nodes.size());
    }
    sc->depth4 = b32.max_depth;
    sc->node_pad = b32.pad;
    // Small scenes are rendered without the hierarchy (device_rt.h: closest_hit_small): each primitive's own box, widened
    // like the node boxes, culls its FP64 test.
    std::vector<float> prim_boxes;
    if (desc->n_shapes > 0 && desc->n_shapes <= kSmallScenePrimsHost && sc->bvh.shape_boxes.size() == static_cast<size_t>(desc->n_shapes)) {
        const double pad = b32.pad > 0.0 ? b32.pad : node4_pad(sc->bvh.root_box, max_camera_coordinate(*desc));
        prim_boxes.resize(static_cast<size_t>(desc->n_shapes) * 6);
        for (int k = 0; k < desc->n_shapes; ++k) {
            const Box b = traversal_bounds(desc->shapes[sc->bvh.order[k]], desc->shutter_time);
            for (int a = 0; a < 3; ++a) {
                prim_boxes[6 * k + a] = round_down(b.lo[a] - pad);
                prim_boxes[6 * k + 3 + a] = round_up(b.hi[a] + pad);
            }
        }
    }
    std::vector<Box>().swap(sc->bvh.shape_boxes);   // 48 bytes per shape the scene no longer needs
    HostArrays h = derive_arrays(*desc, sc->bvh);
    clock.lap("primitive arrays");

    DScene& d = sc->d;
    auto track = [&](void* p) { if (p) sc->allocations.push_back(p); };
    auto cleanup_fail = [&](int code) {
        for (void* p : sc->allocations) (void)hipFree(p);
        return code;
    };
    DPrim* d_prims; int32_t* d_forder; DXform* d_x; DQuad* d_q; DMaterial* d_m; DLight* d_l;
    if (!b32.nodes.empty()) {
        if ((rc = upload(b32.nodes, &d_nodes))) return cleanup_fail(rc);
        track(d_nodes);
    }
    if ((rc = upload(h.prims, &d_prims))) return cleanup_fail(rc); track(d_prims);
    if ((rc = upload(h.file_order, &d_forder))) return cleanup_fail(rc); track(d_forder);
    if ((rc = upload(h.xforms, &d_x))) return cleanup_fail(rc); track(d_x);
    if ((rc = upload(h.quads, &d_q))) return cleanup_fail(rc); track(d_q);
    if ((rc = upload(h.materials, &d_m))) return cleanup_fail(rc); track(d_m);
    if ((rc = upload(h.lights, &d_l))) return cleanup_fail(rc); track(d_l);
    std::vector<DTexture> tex(desc->n_textures);
    for (int i = 0; i < desc->n_textures; ++i) {
        const RtTexture& t = desc->textures[i];
        size_t bytes = static_cast<size_t>(t.width) * t.height * 3;
        uint8_t* p = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), bytes);
        if (e == hipSuccess) { track(p); e = hipMemcpy(p, t.rgb, bytes, hipMemcpyHostToDevice); }
        if (e != hipSuccess) return cleanup_fail(fail(RT_ERR_DEVICE, std::string("texture upload: ") + hipGetErrorString(e)));
        tex[i] = DTexture{p, t.width, t.height};
    }
    DTexture* d_tex;
    if ((rc = upload(tex, &d_tex))) return cleanup_fail(rc); track(d_tex);
    float* d_pbox = nullptr;
    if ((rc = upload(prim_boxes, &d_pbox))) return cleanup_fail(rc); track(d_pbox);
    d.prim_boxes = d_pbox;
    float* d_env = nullptr;
    if (desc->env.present && desc->env.rgb && desc->env.width > 0) {
        size_t bytes = static_cast<size_t>(desc->env.width) * desc->env.height * 3 * sizeof(float);
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_env), bytes);
        if (e == hipSuccess) { track(d_env); e = hipMemcpy(d_env, desc->env.rgb, bytes, hipMemcpyHostToDevice); }
        if (e != hipSuccess) return cleanup_fail(fail(RT_ERR_DEVICE, std::string("environment upload: ") + hipGetErrorString(e)));
    }
    clock.lap("upload");
    d.nodes = d_nodes; d.prims = d_prims; d.file_order = d_forder; d.xforms = d_x; d.quads = d_q;
    d.materials = d_m; d.textures = d_tex; d.lights = d_l; d.env = d_env;
    d.env_present = desc->env.present ? 1 : 0;
    d.env_w = d_env ? desc->env.width : 0;
    d.env_h = d_env ? desc->env.height : 0;
    d.n_prims = desc->n_shapes;
    d.n_nodes = sc->n_nodes4;
    d.n_lights = desc->n_lights;
    d.n_xforms = static_cast<int32_t>(h.xforms.size());
    d.n_quads = static_cast<int32_t>(h.quads.size());
    d.n_materials = desc->n_materials;
    d.root_ref = b32.root_ref;
    d.features = h.features;
    d.spawning_materials = h.spawning_materials;
    d.transparent_materials = h.transparent_materials;
    d.area_lights = h.area_lights;
    for (int a = 0; a < 3; ++a) { d.root_lo[a] = b32.root_lo[a]; d.root_hi[a] = b32.root_hi[a]; }
    d.shutter_time = desc->shutter_time;
    d.cam = h.cam;
    {
        std::lock_guard<std::mutex> lock(g_wf_mutex);
        d.serial = ++g_scene_serial;
    }
    sc->on_device = true;
    *out = sc.release();
    return RT_OK;
}

void rt_scene_destroy(rt_scene_t scene) {
    if (!scene) return;
    if (scene->on_device && g_dev.ready) {
        (void)hipSetDevice(g_dev.device);
        for (void* p : scene->allocations) (void)hipFree(p);
        if (scene->d_frame) (void)hipFree(scene->d_frame);
        for (uint8_t* p : scene->d_batch) if (p) (void)hipFree(p);
    }
    delete scene;
}

int rt_set_bvh_builder(int builder) {
    if (builder != RT_BVH_DEFAULT && builder != RT_BVH_SAH && builder != RT_BVH_LBVH && builder != RT_BVH_MEDIAN && builder != RT_BVH_SAH_GPU)
        return fail(RT_ERR_INVALID, "unknown BVH builder");
    g_builder = builder;
    return RT_OK;
}

int rt_scene_bvh_nodes(rt_scene_t scene, void* nodes_out, int32_t max_nodes, RtBuildInfo* info_out) {
    if (!scene) return fail(RT_ERR_INVALID, "NULL argument");
    if (info_out) {
        info_out->root_ref = scene->d.root_ref;
        info_out->builder = scene->builder;
        info_out->order_on_device = scene->order_on_device ? 1 : 0;
        info_out->reserved = 0;
        info_out->pad = scene->node_pad;
        info_out->device_build_ms = scene->build_ms;
        info_out->device_order_ms = scene->order_ms;
    }
    if (nodes_out) {
        if (max_nodes < scene->n_nodes4) return fail(RT_ERR_LIMIT, "nodes_out is too small");
        if (scene->n_nodes4 > 0) {
            HIP_TRY(hipSetDevice(g_dev.device));
            HIP_TRY(hipMemcpy(nodes_out, scene->d.nodes, static_cast<size_t>(scene->n_nodes4) * sizeof(DNode4), hipMemcpyDeviceToHost));
        }
    }
    return RT_OK;
}

int rt_bvh_order_device(const RtSceneDesc* desc, int32_t* order_out, double* device_ms_out) {
    int rc = validate_desc(desc);
    if (rc) return rc;
    if (!order_out) return fail(RT_ERR_INVALID, "order_out is NULL");
    if (!g_dev.ready) return fail(RT_ERR_NOT_INIT, "rt_init() has not succeeded in this process");
    HIP_TRY(hipSetDevice(g_dev.device));
    if (desc->n_shapes < 2) {
        if (desc->n_shapes == 1) order_out[0] = 0;
        if (device_ms_out) *device_ms_out = 0.0;
        return RT_OK;
    }
    std::vector<int32_t> order;
    float ms = 0.f;
    hipError_t herr = hipSuccess;
    const OrderStatus st = reference_order_device(shape_boxes(*desc, nullptr), g_dev.stream, &order, &ms, &herr);
    if (st == OrderStatus::device_error) return fail(RT_ERR_DEVICE, std::string("device order pass: ") + hipGetErrorString(herr));
    if (st == OrderStatus::ties) return fail(RT_ERR_LIMIT, "equal box centres meet at a split: the reference's order is decided by its nth_element (use the host pass)");
    std::memcpy(order_out, order.data(), order.size() * sizeof(int32_t));
    if (device_ms_out) *device_ms_out = ms;
    return RT_OK;
}

int rt_scene_bvh_info(rt_scene_t scene, RtBvhInfo* out) {
    if (!scene || !out) return fail(RT_ERR_INVALID, "NULL argument");
    out->n_nodes = scene->n_nodes4;
    out->n_prims = scene->n_shapes;
    out->max_depth = scene->depth4;
    out->node_bytes = kNodeBytes;
    store(out->root_min, scene->trav.root_box.lo);
    store(out->root_max, scene->trav.root_box.hi);
    return RT_OK;
}

int rt_part_rows(int height, int tile_rows, int part_index, int part_count) {
    if (height <= 0 || tile_rows <= 0) return 0;
    if (part_count <= 1) return height;
    if (part_index < 0 || part_index >= part_count) return 0;
    int n_tiles = (height + tile_rows - 1) / tile_rows, rows = 0;
    for (int k = part_index; k < n_tiles; k += part_count) rows += std::min(tile_rows, height - k * tile_rows);
    return rows;
}

int rt_part_scatter(const uint8_t* slice, uint8_t* frame, int width, int height, int tile_rows, int part_index, int part_count) {
    if (!slice || !frame || width <= 0 || height <= 0 || tile_rows <= 0) return fail(RT_ERR_INVALID, "bad argument");
    if (part_count <= 1) { std::memcpy(frame, slice, static_cast<size_t>(width) * height * 3); return RT_OK; }
    if (part_index < 0 || part_index >= part_count) return fail(RT_ERR_INVALID, "part_index out of range");
    int n_tiles = (height + tile_rows - 1) / tile_rows;
    size_t row_bytes = static_cast<size_t>(width) * 3, src_row = 0;
    for (int k = part_index; k < n_tiles; k += part_count) {
        int y0 = k * tile_rows, rows = std::min(tile_rows, height - y0);
        std::memcpy(frame + y0 * row_bytes, slice + src_row * row_bytes, rows * row_bytes);
        src_row += rows;
    }
    return RT_OK;
}

static int make_dparams(rt_scene_t scene, const RtRenderParams* p, int collect, DParams* o) {
    if (!scene || !p) return fail(RT_ERR_INVALID, "NULL argument");
    if (p->samples_per_pixel <= 0) return fail(RT_ERR_INVALID, "samples_per_pixel must be >= 1");
    if (p->tonemap < 0 || p->tonemap > 3) return fail(RT_ERR_INVALID, "unknown tonemap mode");
    if (p->rng_mode != RT_RNG_D0 && p->rng_mode != RT_RNG_SEEDED) return fail(RT_ERR_INVALID, "unknown rng_mode");
    if (p->shadows && p->shadow_samples <= 0) return fail(RT_ERR_INVALID, "shadow_samples must be >= 1 when shadows are on");
    if (p->glossy_samples < 0) return fail(RT_ERR_INVALID, "glossy_samples must be >= 0");
    if (p->glossy_samples > 60) return fail(RT_ERR_LIMIT, "glossy_samples is limited to 60 (one bit per reflection ray of a hit in the shading kernel)");
    if (p->schedule != RT_SCHEDULE_WAVEFRONT && p->schedule != RT_SCHEDULE_MEGAKERNEL && p->schedule != RT_SCHEDULE_WAVEFRONT_FP32)
        return fail(RT_ERR_INVALID, "unknown schedule");
    // Megakernel pending-ray stack bound: the first hit may queue glossy_samples + 1 rays, every deeper level at most one more.
    if (p->schedule == RT_SCHEDULE_MEGAKERNEL && std::max(p->max_depth, 0) + p->glossy_samples + 2 > kRayStackDepth)
        return fail(RT_ERR_LIMIT, "max_depth + glossy_samples exceeds the pending-ray stack (" + std::to_string(kRayStackDepth) + ")");
    int parts = p->part_count <= 1 ? 1 : p->part_count;
    int tile = p->tile_rows > 0 ? p->tile_rows : 8;
    if (parts > 1 && (p->part_index < 0 || p->part_index >= parts)) return fail(RT_ERR_INVALID, "part_index out of range");
    DParams d{};
    d.width = scene->width; d.height = scene->height;
    d.spp = p->samples_per_pixel; d.max_depth = p->max_depth; d.max_bounces = p->max_bounces; d.tonemap = p->tonemap;
    d.shadows = p->shadows != 0; d.fresnel = p->fresnel != 0; d.normals = p->normals != 0; d.use_bvh = p->use_bvh != 0;
    d.glossy_samples = p->glossy_samples; d.shadow_samples = p->shadow_samples; d.march_steps = p->ray_march_steps; d.rng_mode = p->rng_mode;
    d.tile_rows = tile; d.part_index = parts > 1 ? p->part_index : 0; d.part_count = parts;
    d.part_rows = rt_part_rows(scene->height, tile, d.part_index, parts);
    d.count_tests = collect ? 1 : 0;
    d.fp32_tests = p->schedule == RT_SCHEDULE_WAVEFRONT_FP32 ? 1 : 0;
    d.frames = 1; d.slice_rows = d.part_rows;
    d.trace_coincident = std::getenv("RT_TRACE_COINCIDENT_RAYS") != nullptr ? 1 : 0;
    d.exposure = p->exposure; d.eps_scene = p->eps_scene; d.eps_triangle = p->eps_triangle; d.eps_march = p->eps_march;
    d.eps_shadow = p->eps_shadow; d.step_multiplier = p->step_multiplier;
    std::memcpy(d.background, p->background, 24);
    std::memcpy(d.ambient_light, p->ambient_light, 24);
    d.seed = p->seed;
    *o = d;
    return RT_OK;
}

// The device scene as the kernels get it (passed by value).  ensure_xforms32 may write scene->d.xforms32 while another thread
// renders the same scene on another stream, so every render entry point takes its copy under the lock that write holds.
static DScene scene_snapshot(rt_scene_t scene) {
    std::lock_guard<std::mutex> lock(g_wf_mutex);
    return scene->d;
}

// RT_SCHEDULE_WAVEFRONT_FP32 on a spheres-only scene: the FP32 copy of the transforms, made on the device by the first such
// frame (on its stream, ahead of its kernels; later frames on other streams wait for it once) and kept with the scene.
static int ensure_xforms32(rt_scene_t scene, const DParams& dp, hipStream_t st) {
    if (!dp.fp32_tests || scene->d.features != RT_FEAT_SPHERE || scene->d.n_xforms <= 0) return RT_OK;
    std::lock_guard<std::mutex> lock(g_wf_mutex);
    if (!scene->d.xforms32) {
        float* p = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), static_cast<size_t>(scene->d.n_xforms) * 16 * sizeof(float));
        if (e == hipSuccess) e = launch_make_xforms32(scene->d.xforms, scene->d.n_xforms, p, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);   // once per scene: every stream may use the copy from here on
        if (e != hipSuccess) {
            if (p) (void)hipFree(p);
            return fail(RT_ERR_DEVICE, std::string("FP32 transforms: ") + hipGetErrorString(e));
        }
        scene->allocations.push_back(p);
        scene->d.xforms32 = p;
    }
    return RT_OK;
}

// hip_stream is a hipStream_t with HIP's own meaning: NULL is the null stream (which is torch's default stream), so work the
// caller enqueues on that stream afterwards is ordered behind the render.
int rt_render_device(rt_scene_t scene, const RtRenderParams* params, void* d_rgb8, void* d_counters, int collect_counters, void* hip_stream) {
    if (!g_dev.ready) return fail(RT_ERR_NOT_INIT, "rt_init() has not succeeded in this process");
    if (!d_rgb8) return fail(RT_ERR_INVALID, "d_rgb8 is NULL");
    DParams dp;
    int rc = make_dparams(scene, params, collect_counters, &dp);
    if (rc) return rc;
    if (!scene->on_device) return fail(RT_ERR_DEVICE, "scene is not resident on the device");
    if (dp.part_rows == 0) return RT_OK;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if ((rc = ensure_xforms32(scene, dp, st))) return rc;
    hipError_t e;
    const DScene ds = scene_snapshot(scene);
    if (params->schedule == RT_SCHEDULE_MEGAKERNEL)
        e = launch_render_mega(ds, dp, static_cast<uint8_t*>(d_rgb8), static_cast<unsigned long long*>(d_counters), st);
    else
        e = launch_render_wavefront(ds, dp, static_cast<uint8_t*>(d_rgb8), static_cast<unsigned long long*>(d_counters), st,
                                    *workspace_of(st, true));
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, std::string("render kernel launch: ") + hipGetErrorString(e));
    return RT_OK;
}

// Several frames of one scene in one set of launches: a frame's share of the work on one of many GPUs is small, and a small
// launch of the persistent traversal kernel spends most of its time in the tail of its longest rays; F frames in one
// launch have one tail.  Frame f is rendered with seed + f (progressive passes, or the frames of a throughput benchmark).
int rt_render_frames_device(rt_scene_t scene, const RtRenderParams* params, int frames, int slice_rows, void* d_rgb8, void* d_counters,
                            void* hip_stream) {
    if (!g_dev.ready) return fail(RT_ERR_NOT_INIT, "rt_init() has not succeeded in this process");
    if (!d_rgb8) return fail(RT_ERR_INVALID, "d_rgb8 is NULL");
    if (frames < 1 || frames > 4096) return fail(RT_ERR_INVALID, "frames must be 1 .. 4096");
    DParams dp;
    int rc = make_dparams(scene, params, 0, &dp);
    if (rc) return rc;
    if (params->schedule == RT_SCHEDULE_MEGAKERNEL) return fail(RT_ERR_INVALID, "rt_render_frames_device renders with the wavefront schedule");
    if (!scene->on_device) return fail(RT_ERR_DEVICE, "scene is not resident on the device");
    if (slice_rows == 0) slice_rows = dp.part_rows;
    if (slice_rows < dp.part_rows) return fail(RT_ERR_INVALID, "slice_rows is smaller than the rows this part owns (rt_part_rows)");
    if (dp.part_rows == 0) return RT_OK;
    if (static_cast<long long>(frames) * slice_rows * dp.width >= (1ll << 31))
        return fail(RT_ERR_LIMIT, "frames x slice_rows x width must stay below 2^31 pixels per call");
    dp.frames = frames;
    dp.slice_rows = slice_rows;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if ((rc = ensure_xforms32(scene, dp, st))) return rc;
    hipError_t e = launch_render_wavefront(scene_snapshot(scene), dp, static_cast<uint8_t*>(d_rgb8), static_cast<unsigned long long*>(d_counters), st,
                                           *workspace_of(st, true));
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, std::string("render kernel launch: ") + hipGetErrorString(e));
    return RT_OK;
}

int rt_render_device_noalloc(rt_scene_t scene, const RtRenderParams* params, void* d_rgb8, void* hip_stream) {
    if (!g_dev.ready) return fail(RT_ERR_NOT_INIT, "rt_init() has not succeeded in this process");
    if (!d_rgb8) return fail(RT_ERR_INVALID, "d_rgb8 is NULL");
    DParams dp;
    int rc = make_dparams(scene, params, 0, &dp);
    if (rc) return rc;
    if (params->schedule != RT_SCHEDULE_WAVEFRONT) return fail(RT_ERR_INVALID, "rt_render_device_noalloc renders with the wavefront schedule");
    if (!scene->on_device) return fail(RT_ERR_DEVICE, "scene is not resident on the device");
    if (dp.part_rows == 0) return RT_OK;
    const long long npix = static_cast<long long>(dp.part_rows) * dp.width;
    if (npix > kStaticMaxPixels || dp.width > kStaticMaxWidth)
        return fail(RT_ERR_LIMIT, "rt_render_device_noalloc: the part has " + std::to_string(npix) + " pixels, " + std::to_string(dp.width) +
                                      " wide; the static workspace holds parts of at most " + std::to_string(kStaticMaxPixels) + " pixels, at most " +
                                      std::to_string(kStaticMaxWidth) + " wide (render the frame in parts, or use rt_render_device)");
    hipError_t e = launch_render_wavefront_static(scene_snapshot(scene), dp, static_cast<uint8_t*>(d_rgb8), static_cast<hipStream_t>(hip_stream), g_dev.device);
    if (e == hipErrorOutOfMemory)
        return fail(RT_ERR_LIMIT, "rt_render_device_noalloc: the ray tree of this frame needs more queues than the static workspace (" +
                                      std::to_string(kStaticScratchBytes >> 20) + " MiB) holds; use rt_render_device");
    if (e == hipErrorInvalidValue)   // launch_render_wavefront_static / launch_render_wavefront: a band of 8 rows x width does not fit a queue
        return fail(RT_ERR_LIMIT, "rt_render_device_noalloc: an 8-row band of this part (" + std::to_string(dp.width) + " pixels wide, " +
                                      std::to_string(dp.spp) + " samples) does not fit a ray queue of the static workspace; render narrower parts or use rt_render_device");
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, std::string("render kernel launch: ") + hipGetErrorString(e));
    return RT_OK;
}

int rt_render_linear_device(rt_scene_t scene, const RtRenderParams* params, void* d_rgb_f64, void* hip_stream) {
    if (!g_dev.ready) return fail(RT_ERR_NOT_INIT, "rt_init() has not succeeded in this process");
    if (!d_rgb_f64) return fail(RT_ERR_INVALID, "d_rgb_f64 is NULL");
    DParams dp;
    int rc = make_dparams(scene, params, 0, &dp);
    if (rc) return rc;
    if (params->schedule != RT_SCHEDULE_WAVEFRONT) return fail(RT_ERR_INVALID, "linear output needs the wavefront schedule");
    if (!scene->on_device) return fail(RT_ERR_DEVICE, "scene is not resident on the device");
    if (dp.part_rows == 0) return RT_OK;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    hipError_t e = launch_render_wavefront(scene_snapshot(scene), dp, nullptr, nullptr, st, *workspace_of(st, true), static_cast<double*>(d_rgb_f64));
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, std::string("render kernel launch: ") + hipGetErrorString(e));
    return RT_OK;
}

// Progressive accumulation with the running sum in HBM: pass k of the series is rendered with seed params->seed + k.
int rt_accumulate_device(rt_scene_t scene, const RtRenderParams* params, void* d_sum_f64, uint64_t first_pass, int passes, void* hip_stream) {
    if (!g_dev.ready) return fail(RT_ERR_NOT_INIT, "rt_init() has not succeeded in this process");
    if (!d_sum_f64) return fail(RT_ERR_INVALID, "d_sum_f64 is NULL");
    if (passes < 0) return fail(RT_ERR_INVALID, "passes must be >= 0");
    if (!params) return fail(RT_ERR_INVALID, "NULL argument");
    if (params->rng_mode != RT_RNG_SEEDED) return fail(RT_ERR_INVALID, "progressive accumulation needs RT_RNG_SEEDED: identical passes add nothing");
    if (params->schedule != RT_SCHEDULE_WAVEFRONT) return fail(RT_ERR_INVALID, "linear output needs the wavefront schedule");
    RtRenderParams p = *params;
    DParams dp;
    int rc = make_dparams(scene, &p, 0, &dp);
    if (rc) return rc;
    if (!scene->on_device) return fail(RT_ERR_DEVICE, "scene is not resident on the device");
    if (dp.part_rows == 0) return RT_OK;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    WavefrontWorkspace& ws = *workspace_of(st, true);
    double* d_sum = static_cast<double*>(d_sum_f64);
    // One pass after the other on the caller's stream, each added to the sum by the resolve kernel.  A pass of many samples is a
    // row of chunks, which launch_render_wavefront renders two at a time (a second set of queues, a stream and a host thread of
    // the library's own): that hides the small launches and the read-back of one chunk behind the other's large launches, which
    // is what running two PASSES at a time did before (4K x 16 spp with two area lights: 92 -> 82 ms per pass either way; two
    // passes at a time on top of chunk pairs would hold four sets of queues).
    hipError_t e = hipSuccess;
    const DScene ds = scene_snapshot(scene);
    int done = 0;
    for (; done < passes; ++done) {
        dp.seed = params->seed + first_pass + static_cast<uint64_t>(done);
        e = launch_render_wavefront(ds, dp, nullptr, nullptr, st, ws, d_sum, 1);
        if (e != hipSuccess) break;
    }
    // A pass is added to the sum by its last kernel, after its render has succeeded: the sum holds exactly the passes counted here.
    if (e != hipSuccess)
        return fail(RT_ERR_DEVICE, "rt_accumulate_device: pass " + std::to_string(done) + " of " + std::to_string(passes) + " failed (" + hipGetErrorString(e) +
                                       "); d_sum_f64 holds the " + std::to_string(done) + " passes before it (series positions " + std::to_string(first_pass) +
                                       " .. " + std::to_string(first_pass + static_cast<uint64_t>(done)) + " exclusive)");
    return RT_OK;
}

int rt_resolve_sum_device(const void* d_sum_f64, uint64_t passes, int width, int rows, int tonemap, void* d_rgb8, void* hip_stream) {
    if (!g_dev.ready) return fail(RT_ERR_NOT_INIT, "rt_init() has not succeeded in this process");
    if (!d_sum_f64 || !d_rgb8 || width <= 0 || rows < 0 || passes == 0) return fail(RT_ERR_INVALID, "rt_resolve_sum_device: bad argument");
    if (tonemap < 0 || tonemap > 3) return fail(RT_ERR_INVALID, "unknown tonemap mode");
    hipError_t e = launch_resolve_sum(static_cast<const double*>(d_sum_f64), passes, static_cast<size_t>(width) * rows, tonemap,
                                      static_cast<uint8_t*>(d_rgb8), static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, std::string("resolve kernel launch: ") + hipGetErrorString(e));
    return RT_OK;
}

int rt_render_linear(rt_scene_t scene, const RtRenderParams* params, double* rgb_out) {
    if (!g_dev.ready) return fail(RT_ERR_NOT_INIT, "rt_init() has not succeeded in this process");
    if (!scene || !params || !rgb_out) return fail(RT_ERR_INVALID, "NULL argument");
    if (params->part_count > 1) return fail(RT_ERR_INVALID, "rt_render_linear renders whole frames (use rt_render_linear_device for parts)");
    HIP_TRY(hipSetDevice(g_dev.device));
    const size_t bytes = static_cast<size_t>(scene->width) * scene->height * 3 * sizeof(double);
    double* d = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), bytes));
    int rc = rt_render_linear_device(scene, params, d, g_dev.stream);
    hipError_t e = hipSuccess;
    if (!rc) e = hipMemcpyAsync(rgb_out, d, bytes, hipMemcpyDeviceToHost, g_dev.stream);
    if (!rc && e == hipSuccess) e = hipStreamSynchronize(g_dev.stream);
    (void)hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, std::string("linear frame copy: ") + hipGetErrorString(e));
    return RT_OK;
}

int rt_traverse_time(rt_scene_t scene, void* hip_stream, double* ms_out, uint64_t* launches_out) {
    if (!scene || !ms_out || !launches_out) return fail(RT_ERR_INVALID, "NULL argument");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    WavefrontWorkspace* ws = workspace_of(st, false);
    *ms_out = 0.0;
    *launches_out = 0;
    if (!ws) return RT_OK;
    unsigned launches = 0;
    hipError_t e = wavefront_traverse_time(*ws, ms_out, &launches);
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, std::string("traversal events: ") + hipGetErrorString(e));
    *launches_out = launches;
    return RT_OK;
}

int rt_render(rt_scene_t scene, const RtRenderParams* params, uint8_t* rgb8_out, RtStats* stats) {
    if (!g_dev.ready) return fail(RT_ERR_NOT_INIT, "rt_init() has not succeeded in this process");
    std::lock_guard<std::recursive_mutex> host_lock(g_host_render_mutex);
    if (!scene || !params || !rgb8_out) return fail(RT_ERR_INVALID, "NULL argument");
    auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(g_dev.device));
    int parts = params->part_count <= 1 ? 1 : params->part_count;
    int tile = params->tile_rows > 0 ? params->tile_rows : 8;
    int rows = rt_part_rows(scene->height, tile, parts > 1 ? params->part_index : 0, parts);
    size_t bytes = static_cast<size_t>(rows) * scene->width * 3;
    if (scene->d_frame_bytes < bytes) {
        if (scene->d_frame) (void)hipFree(scene->d_frame);
        scene->d_frame = nullptr;
        scene->d_frame_bytes = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&scene->d_frame), bytes ? bytes : 1));
        scene->d_frame_bytes = bytes;
    }
    HIP_TRY(hipMemsetAsync(g_dev.d_counters, 0, RT_N_COUNTERS * sizeof(uint64_t), g_dev.stream));
    HIP_TRY(hipEventRecord(g_dev.ev0, g_dev.stream));
    int rc = rt_render_device(scene, params, scene->d_frame, g_dev.d_counters, stats != nullptr, g_dev.stream);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(g_dev.ev1, g_dev.stream));
    std::vector<uint8_t> slice;
    uint8_t* dst = rgb8_out;
    if (parts > 1) { slice.resize(bytes); dst = slice.data(); }
    // (Through the pinned staging buffer of rt_render_frames and one memcpy instead: measured the same, 1.50 against 1.52 ms for the
    // C2 frame -- the runtime's own staged copy into pageable memory already overlaps its DMA with its memcpy.)
    if (bytes) HIP_TRY(copy_out(dst, scene->d_frame, bytes, g_dev.stream));
    uint64_t counters[RT_N_COUNTERS] = {};
    if (stats) HIP_TRY(hipMemcpyAsync(counters, g_dev.d_counters, sizeof(counters), hipMemcpyDeviceToHost, g_dev.stream));
    HIP_TRY(hipStreamSynchronize(g_dev.stream));
    if (parts > 1 && bytes) {
        rc = rt_part_scatter(slice.data(), rgb8_out, scene->width, scene->height, tile, params->part_index, parts);
        if (rc) return rc;
    }
    if (stats) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g_dev.ev0, g_dev.ev1));
        stats->rays = counters[0];
        stats->nodes_visited = counters[1];
        stats->prims_tested = counters[2];
        stats->nodes_fetched = counters[4];
        stats->rays_coincident = counters[7];
        stats->kernel_ms = ms;
        stats->traverse_ms = static_cast<double>(counters[3]) * 1e-6;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return RT_OK;
}

// `frames` frames of one scene into host memory: the frames are rendered in batches (rt_render_frames_device: the rays of a
// batch share one set of launches and one tail of the traversal kernel) into two sets of slices that alternate, and the
// device -> host copy of a batch runs on a stream of its own while the next batch renders.
int rt_render_frames(rt_scene_t scene, const RtRenderParams* params, int frames, uint8_t* rgb8_out, RtStats* stats) {
    if (!g_dev.ready) return fail(RT_ERR_NOT_INIT, "rt_init() has not succeeded in this process");
    std::lock_guard<std::recursive_mutex> host_lock(g_host_render_mutex);
    if (!scene || !params || !rgb8_out) return fail(RT_ERR_INVALID, "NULL argument");
    if (frames < 1) return fail(RT_ERR_INVALID, "frames must be >= 1");
    if (params->part_count > 1) return fail(RT_ERR_INVALID, "rt_render_frames renders whole frames (use rt_render_frames_device for parts)");
    if (params->schedule == RT_SCHEDULE_MEGAKERNEL) return fail(RT_ERR_INVALID, "rt_render_frames renders with the wavefront schedule");
    auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(g_dev.device));
    const size_t frame_bytes = static_cast<size_t>(scene->height) * scene->width * 3;
    // batches of at most 16 frames (what one launch gains from more is small: DESIGN section 5), and at least two batches when
    // there are frames enough, so that a copy has a render to hide behind
    int batch = std::min(frames, 16);
    if (frames >= 4 && frames <= 16) batch = (frames + 1) / 2;
    const size_t need = frame_bytes * static_cast<size_t>(batch);
    if (scene->d_batch_bytes < need) {
        HIP_TRY(hipDeviceSynchronize());
        for (uint8_t*& p : scene->d_batch) { if (p) (void)hipFree(p); p = nullptr; }
        scene->d_batch_bytes = 0;
        for (uint8_t*& p : scene->d_batch) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p), need));
        scene->d_batch_bytes = need;
    }
    // Where the copies go: straight into the caller's buffer when it is (or can be) registered with the driver, else through
    // the pinned staging pair (this pool's driver refuses hipHostRegister: "invalid argument").
    const bool direct = pin_host(rgb8_out, frame_bytes * static_cast<size_t>(frames));
    if (!direct && g_dev.staging_bytes < need) {
        HIP_TRY(hipDeviceSynchronize());
        for (uint8_t*& p : g_dev.staging) { if (p) (void)hipHostFree(p); p = nullptr; }
        g_dev.staging_bytes = 0;
        for (uint8_t*& p : g_dev.staging) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p), need, hipHostMallocDefault));
        g_dev.staging_bytes = need;
    }
    HIP_TRY(hipMemsetAsync(g_dev.d_counters, 0, RT_N_COUNTERS * sizeof(uint64_t), g_dev.stream));
    HIP_TRY(hipEventRecord(g_dev.ev0, g_dev.stream));
    RtRenderParams q = *params;
    q.part_index = 0; q.part_count = 1;
    // what lies in staging set s and where it belongs: moved on by the host once the set's copy has arrived
    struct Pending { uint8_t* dst = nullptr; size_t bytes = 0; } pending[2];
    auto drain = [&](int set) -> hipError_t {
        if (!pending[set].dst) return hipSuccess;
        const hipError_t e = hipEventSynchronize(g_dev.copied[set]);
        if (e != hipSuccess) return e;
        host_copy(pending[set].dst, g_dev.staging[set], pending[set].bytes);
        pending[set].dst = nullptr;
        return hipSuccess;
    };
    int k = 0;
    for (int f = 0; f < frames; f += batch, ++k) {
        const int nb = std::min(batch, frames - f), set = k & 1;
        if (!direct) HIP_TRY(drain(set));                                            // staging set free again (batch k - 2)
        if (k >= 2) HIP_TRY(hipStreamWaitEvent(g_dev.stream, g_dev.copied[set], 0));   // the set's previous frames have left the device
        q.seed = params->seed + static_cast<uint64_t>(f);
        int rc = rt_render_frames_device(scene, &q, nb, 0, scene->d_batch[set], g_dev.d_counters, g_dev.stream);
        if (rc) { (void)hipDeviceSynchronize(); return rc; }
        HIP_TRY(hipEventRecord(g_dev.rendered[set], g_dev.stream));
        HIP_TRY(hipStreamWaitEvent(g_dev.copy_stream, g_dev.rendered[set], 0));
        uint8_t* dst = rgb8_out + static_cast<size_t>(f) * frame_bytes;
        const size_t bytes = frame_bytes * static_cast<size_t>(nb);
        if (direct) {
            HIP_TRY(copy_out(dst, scene->d_batch[set], bytes, g_dev.copy_stream));
        } else {
            HIP_TRY(hipMemcpyAsync(g_dev.staging[set], scene->d_batch[set], bytes, hipMemcpyDeviceToHost, g_dev.copy_stream));
            pending[set] = Pending{dst, bytes};
        }
        HIP_TRY(hipEventRecord(g_dev.copied[set], g_dev.copy_stream));
        if (!direct) HIP_TRY(drain(set ^ 1));   // the batch before this one: its copy ran while this one was enqueued, the GPU renders on
    }
    if (!direct) { HIP_TRY(drain(0)); HIP_TRY(drain(1)); }
    HIP_TRY(hipEventRecord(g_dev.ev1, g_dev.stream));
    uint64_t counters[RT_N_COUNTERS] = {};
    if (stats) HIP_TRY(hipMemcpyAsync(counters, g_dev.d_counters, sizeof(counters), hipMemcpyDeviceToHost, g_dev.stream));
    HIP_TRY(hipStreamSynchronize(g_dev.stream));
    HIP_TRY(hipStreamSynchronize(g_dev.copy_stream));
    if (stats) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g_dev.ev0, g_dev.ev1));
        *stats = RtStats{};
        stats->rays = counters[0];
        stats->rays_coincident = counters[7];
        stats->kernel_ms = ms;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return RT_OK;
}

// rt_render for every rank of the communicator (collective): each rank renders its interleaved row tiles, the slices are
// gathered over RCCL and rank 0 receives the assembled frame in host memory.
int rt_render_gathered(rt_scene_t scene, const RtRenderParams* params, uint8_t* rgb8_out, RtStats* stats) {
    if (!g_dev.ready) return fail(RT_ERR_NOT_INIT, "rt_init() has not succeeded in this process");
    std::lock_guard<std::recursive_mutex> host_lock(g_host_render_mutex);
    if (!scene || !params) return fail(RT_ERR_INVALID, "NULL argument");
    const int world = rt_comm_world(), rank = rt_comm_rank();
    const char* lb = std::getenv("RT_COMM_LOOPBACK");
    if (world == 1 && !(lb && lb[0] == '1')) {
        RtRenderParams q = *params;
        q.part_index = 0; q.part_count = 1;
        return rt_render(scene, &q, rgb8_out, stats);
    }
    if (rank == 0 && !rgb8_out) return fail(RT_ERR_INVALID, "rank 0 needs rgb8_out");
    auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(g_dev.device));
    RtRenderParams q = *params;
    q.part_index = rank; q.part_count = world;
    const int tile = q.tile_rows > 0 ? q.tile_rows : 8;
    q.tile_rows = tile;
    int max_rows = 0;
    for (int k = 0; k < world; ++k) max_rows = std::max(max_rows, rt_part_rows(scene->height, tile, k, world));
    const size_t slice_bytes = static_cast<size_t>(max_rows) * scene->width * 3;
    const size_t frame_bytes = static_cast<size_t>(scene->height) * scene->width * 3;
    const size_t bytes = slice_bytes + (rank == 0 ? frame_bytes : 0);   // [slice][assembled frame]
    if (scene->d_frame_bytes < bytes) {
        if (scene->d_frame) (void)hipFree(scene->d_frame);
        scene->d_frame = nullptr;
        scene->d_frame_bytes = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&scene->d_frame), bytes));
        scene->d_frame_bytes = bytes;
    }
    HIP_TRY(hipMemsetAsync(g_dev.d_counters, 0, RT_N_COUNTERS * sizeof(uint64_t), g_dev.stream));
    HIP_TRY(hipEventRecord(g_dev.ev0, g_dev.stream));
    int rc = rt_render_device(scene, &q, scene->d_frame, g_dev.d_counters, stats != nullptr, g_dev.stream);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(g_dev.ev1, g_dev.stream));
    rc = rt_gather_frames(scene->d_frame, 1, scene->width, scene->height, tile, rank == 0 ? scene->d_frame + slice_bytes : nullptr, g_dev.stream);
    if (rc) return rc;
    if (rank == 0) HIP_TRY(hipMemcpyAsync(rgb8_out, scene->d_frame + slice_bytes, frame_bytes, hipMemcpyDeviceToHost, g_dev.stream));
    uint64_t counters[RT_N_COUNTERS] = {};
    HIP_TRY(hipMemcpyAsync(counters, g_dev.d_counters, sizeof(counters), hipMemcpyDeviceToHost, g_dev.stream));
    HIP_TRY(hipStreamSynchronize(g_dev.stream));
    if (stats) {   // this rank's share
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g_dev.ev0, g_dev.ev1));
        stats->rays = counters[0];
        stats->nodes_visited = counters[1];
        stats->prims_tested = counters[2];
        stats->nodes_fetched = counters[4];
        stats->rays_coincident = counters[7];
        stats->kernel_ms = ms;
        stats->traverse_ms = static_cast<double>(counters[3]) * 1e-6;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return RT_OK;
}

// ---------------------------------------------------------------- config / scene files / images (host only)

int rt_config_load(const char* path, rt_config_t* out) {
    if (!path || !out) return fail(RT_ERR_INVALID, "NULL argument");
    auto c = new rt_config_s();
    *out = c;
    if (!c->cfg.load_file(path)) return fail(RT_ERR_IO, std::string("Warning: Could not open config file '") + path + "'. Using defaults.");
    return RT_OK;
}
int rt_config_parse(const char* text, rt_config_t* out) {
    if (!text || !out) return fail(RT_ERR_INVALID, "NULL argument");
    auto c = new rt_config_s();
    c->cfg.parse_text(text);
    *out = c;
    return RT_OK;
}
int rt_config_get_int(rt_config_t cfg, const char* key, int def) { return (cfg && key) ? cfg->cfg.get_int(key, def) : def; }
double rt_config_get_double(rt_config_t cfg, const char* key, double def) { return (cfg && key) ? cfg->cfg.get_double(key, def) : def; }
int rt_config_get_bool(rt_config_t cfg, const char* key, int def) { return (cfg && key) ? (cfg->cfg.get_bool(key, def != 0) ? 1 : 0) : def; }
void rt_config_destroy(rt_config_t cfg) { delete cfg; }

int rt_params_from_config(rt_config_t cfg, RtRenderParams* o) {
    if (!o) return fail(RT_ERR_INVALID, "NULL argument");
    ConfigData empty;
    const ConfigData& c = cfg ? cfg->cfg : empty;
    std::memset(o, 0, sizeof(*o));
    o->samples_per_pixel = c.get_int("settings.samples_per_pixel", 1);   // main.cpp:54
    o->max_depth = c.get_int("settings.max_bounces", 10);                // main.cpp:249
    o->max_bounces = c.get_int("settings.max_bounces", 5);               // scene.cpp:96
    o->tonemap = RT_TONEMAP_NONE;
    o->use_bvh = 1;
    o->glossy_samples = 0;                                               // main.cpp:59
    o->shadow_samples = c.get_int("render.shadow_samples", 4);           // scene.cpp:94
    o->ray_march_steps = c.get_int("advanced.ray_march_steps", 64);      // complex_*.cpp
    o->rng_mode = RT_RNG_SEEDED;
    o->exposure = c.get_double("image.exposure", 1.0);                   // main.cpp:56
    o->eps_scene = c.get_double("advanced.epsilon", 1e-4);               // scene.cpp:95
    o->eps_triangle = c.get_double("advanced.epsilon", 0.001);           // plane.cpp:83
    o->eps_march = c.get_double("advanced.epsilon", 0.001);              // complex_*.cpp
    o->eps_shadow = c.get_double("advanced.epsilon", 1e-4);              // shading.h:83
    o->step_multiplier = c.get_double("advanced.step_multiplier", 0.8);  // complex_plane.cpp:167
    o->background[0] = c.get_double("background.r", 0.5);                // tracer.h:290-292
    o->background[1] = c.get_double("background.g", 0.7);
    o->background[2] = c.get_double("background.b", 1.0);
    o->ambient_light[0] = c.get_double("lighting.r", 0.15);              // shading.h:142-144
    o->ambient_light[1] = c.get_double("lighting.g", 0.15);
    o->ambient_light[2] = c.get_double("lighting.b", 0.25);
    o->seed = 1;
    o->tile_rows = 8;
    o->part_index = 0;
    o->part_count = 1;
    return RT_OK;
}

static int scene_file_finish(std::unique_ptr<SceneFile> s, rt_scene_file_t* out) {
    auto f = new rt_scene_file_s();
    f->scene = std::move(s);
    *out = f;
    return RT_OK;
}
int rt_scene_file_load(const char* path, const char* texture_root, rt_config_t cfg, double shutter_time, rt_scene_file_t* out) {
    if (!path || !out) return fail(RT_ERR_INVALID, "NULL argument");
    *out = nullptr;
    ConfigData empty;
    try {
        return scene_file_finish(load_scene_file(path, texture_root ? texture_root : "", cfg ? cfg->cfg : empty, shutter_time), out);
    } catch (const std::exception& e) {
        return fail(RT_ERR_IO, e.what());
    }
}
int rt_scene_file_parse(const char* text, const char* texture_root, rt_config_t cfg, double shutter_time, rt_scene_file_t* out) {
    if (!text || !out) return fail(RT_ERR_INVALID, "NULL argument");
    *out = nullptr;
    ConfigData empty;
    try {
        return scene_file_finish(parse_scene_text(text, texture_root ? texture_root : "", cfg ? cfg->cfg : empty, shutter_time), out);
    } catch (const std::exception& e) {
        return fail(RT_ERR_IO, e.what());
    }
}
const RtSceneDesc* rt_scene_file_desc(rt_scene_file_t f) { return f ? &f->scene->desc : nullptr; }
void rt_scene_file_destroy(rt_scene_file_t f) { delete f; }

int rt_shapes_from_trs(int n, const double* trs, RtShape* shapes) {
    if (n < 0 || (n && (!trs || !shapes))) return fail(RT_ERR_INVALID, "bad argument");
    try {
        for (int i = 0; i < n; ++i) shape_set_trs(shapes[i], trs + i * 9, trs + i * 9 + 3, trs + i * 9 + 6);
    } catch (const std::exception& e) {
        return fail(RT_ERR_INVALID, e.what());
    }
    return RT_OK;
}

int rt_write_pfm(const char* path, int width, int height, const double* rgb) {
    if (!path || !rgb || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "bad argument");
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return fail(RT_ERR_IO, std::string("Could not open PFM file for writing: ") + path);
    std::fprintf(f, "PF\n%d %d\n-1.0\n", width, height);   // negative scale: little endian
    std::vector<float> row(static_cast<size_t>(width) * 3);
    bool ok = true;
    for (int y = height - 1; y >= 0 && ok; --y) {   // PFM rows run bottom to top (HDRImage.cpp:112-182 reads them that way)
        const double* src = rgb + static_cast<size_t>(y) * width * 3;
        for (size_t i = 0; i < row.size(); ++i) row[i] = static_cast<float>(src[i]);
        ok = std::fwrite(row.data(), sizeof(float), row.size(), f) == row.size();
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? RT_OK : fail(RT_ERR_IO, std::string("short write: ") + path);
}

int rt_write_ppm(const char* path, int width, int height, const uint8_t* rgb) {
    if (!path || !rgb || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "bad argument");
    try {
        write_ppm(path, width, height, rgb);
    } catch (const std::exception& e) {
        return fail(RT_ERR_IO, e.what());
    }
    return RT_OK;
}
int rt_read_ppm(const char* path, int* width, int* height, uint8_t** rgb_out) {
    if (!path || !width || !height || !rgb_out) return fail(RT_ERR_INVALID, "NULL argument");
    try {
        Image8 img = read_ppm(path);
        *width = img.width;
        *height = img.height;
        *rgb_out = static_cast<uint8_t*>(std::malloc(img.rgb.size()));
        if (!*rgb_out) return fail(RT_ERR_LIMIT, "out of memory");
        std::memcpy(*rgb_out, img.rgb.data(), img.rgb.size());
    } catch (const std::exception& e) {
        return fail(RT_ERR_IO, e.what());
    }
    return RT_OK;
}
int rt_read_pfm(const char* path, int* width, int* height, float** rgb_out) {
    if (!path || !width || !height || !rgb_out) return fail(RT_ERR_INVALID, "NULL argument");
    ImageF img;
    std::string why;
    if (!read_pfm(path, img, &why)) return fail(RT_ERR_IO, why);
    *width = img.width;
    *height = img.height;
    *rgb_out = static_cast<float*>(std::malloc(img.rgb.size() * sizeof(float)));
    if (!*rgb_out) return fail(RT_ERR_LIMIT, "out of memory");
    std::memcpy(*rgb_out, img.rgb.data(), img.rgb.size() * sizeof(float));
    return RT_OK;
}

}  // extern "C"
