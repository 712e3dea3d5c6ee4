// Wavefront schedule of the render path: the Whitted tree of every pixel is evaluated breadth-first as queues of rays
// in HBM, so that the two very different kinds of work run in kernels of their own:
//
//   k_raygen    camera rays (AA jitter, lens sample, ray time)                -> ray queue       camera.cpp:89-135, main.cpp:288-302
//   k_traverse  closest hit of every queued ray through the flattened BVH     -> hit records     bvh.cpp:101-114, aabb.cpp:4-38, shapes/*.cpp
//   k_shade     hit -> Blinn-Phong terms, shadow / reflection / refraction    -> next ray queue, tracer.h:147-295, shading.h:42-217
//               rays carrying their colour weight; local terms added to the      FP64 accumulators
//               pixel accumulator
//   k_resolve   average, tone map, 8-bit                                      -> RGB8 rows       main.cpp:308-320, tracer.h:23-54,297-308
//
// Why: on the 10^6-sphere frame a lane's traversal length varies by an order of magnitude inside one 8x8 tile, and a
// one-lane-per-pixel megakernel idles 85 % of its lanes (measured).  k_traverse is a persistent kernel:
// a wave reserves chunks of the queue (one share of it per XCD, so that an XCD's L2 sees neighbouring rays), lanes that
// finish their ray take the next one at once, and the wave votes on whether to spend the next step on interior nodes
// (a tight loop over 4-wide, 8-bit quantised nodes with a per-lane stack of (child, entry distance) in LDS) or on the
// much more expensive FP64 primitive tests of the leaves its lanes have parked meanwhile.
//
// Two properties of this chip shape the rest (MI355X_MICROARCH.md): float atomics execute at the memory side at a few
// G/s, and a returning atomic on one word serialises.  Hence contributions of neighbouring lanes to the same pixel
// are summed inside the wave before the FP64 atomics, a hit reserves the queue slots of all its shadow rays (and of
// all its kept glossy samples) with one atomic per wave, queues are planes of 16-byte pieces so that a wave's accesses
// are contiguous, and a frame with one ray per pixel writes its pixels straight from the shading kernel.
//
// The colour of a ray is linear in the colours of its child rays (tracer.h:268-279) and in the visibility of each
// light (shading.h:166-171, 206-211), so every queued ray carries the product of the weights above it and adds
// weight (x) result to its pixel when it terminates; no per-pixel recursion state is kept.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "device_rt.h"
#include "render_kernel.h"

namespace rt {

namespace {

constexpr int kBlock = 256;
constexpr unsigned kFetchCounterWords = 256;   // 8 XCD counters, 128 bytes apart (k_traverse)
constexpr int kKindRadiance = 0, kKindShadow = 1;

struct alignas(16) RayRec {
    double o[3], d[3], w[3];   // origin, direction, colour weight of whatever this ray returns
    double time;
    // One word for two things no ray needs both of: a radiance ray always searches up to 1e5 (tracer.h:156) and draws random
    // numbers at its hit; a shadow ray (a continuation through transmitting surfaces) searches up to the distance that is
    // left to its light sample (shading.h:45) and draws nothing.  96 bytes = six 16-byte pieces instead of seven.
    union {
        double tmax_shadow;            // kind == kKindShadow
        unsigned long long rng;        // kind == kKindRadiance: state of this ray's random stream
    };
    int pixel;                 // accumulator index inside this part
    short depth;               // remaining depth (tracer.h:149) for radiance rays
    short kind;
    __device__ __forceinline__ double tmax() const { return kind == kKindShadow ? tmax_shadow : 100000.0; }
};
// What lies in a queue: the HEAD, four 16-byte pieces, is all a traversal kernel reads -- origin, direction, time and the
// tmax-or-rng word, the ray's kind being the SIGN of the time (a time is never negative, so nothing is lost: -0.0 is a shadow
// ray at time 0) -- and all a primary ray has (the rest of its record is implied, see RayQ::primary); the TAIL, two pieces,
// is what only the shading kernel needs: colour weight, pixel, depth.
struct alignas(16) RayStore {
    double o[3], d[3];
    double time_kind;          // |.| = time; sign set = kKindShadow
    unsigned long long word;   // tmax_shadow or rng
    double w[3];
    int pixel;
    short depth, pad;
};
static_assert(sizeof(RayStore) == 96, "RayStore size");
constexpr int kRayPieces = sizeof(RayStore) / 16, kRayHeadPieces = 4;

struct alignas(16) HitRec {   // piece 0: what every hit needs; piece 1: the quad's barycentrics, which only texture / bump lookups read
    double t;
    int prim, tri;
    double bu, bv;
};
static_assert(sizeof(HitRec) == 32, "HitRec size");

// Queues in HBM are stored as planes of 16-byte pieces (piece k of record i at p[k * stride + i], stride = queue
// capacity): consecutive lanes work on consecutive records, so every load / store instruction of a wave covers 1 KB of
// contiguous memory instead of 64 pieces 112 bytes apart.
typedef unsigned rt_q4 __attribute__((ext_vector_type(4)));
struct Chunk {
    int row0, rows;     // rows [row0, row0 + rows) of the launch (compact rows of this part; frame after frame or band by band, k_raygen)
    int s0, ns;         // samples [s0, s0 + ns) of every pixel
};
// Primary rays (the queue k_raygen writes: level 0 of every ray tree) are stored SLIM: the head of a RayStore alone, 64 bytes
// instead of 96.  The rest of a primary ray's record is implied -- weight 1, depth max_depth -- or follows from its index in
// the queue (pixel: primary_coords).
struct RayQ {
    rt_q4* p;
    unsigned stride;
    unsigned primary;   // 1: slim primary records of chunk `ck`
    Chunk ck;
};
__device__ __host__ __forceinline__ RayQ rebase(const RayQ& q, unsigned origin) { return RayQ{q.p + origin, q.stride, q.primary, q.ck}; }
struct HitQ {
    rt_q4* p;
    unsigned stride;
};
// Queue records (rays, hit records, shadow bundles) are written by one kernel and read once by a later one, gigabytes apart: their
// loads and stores carry the streaming hint (nontemporal), so they do not displace the scene and the tails from the caches.
// C1 + shadows 1.35 -> 1.30 ms, C3 129.6 -> 126.0, C4 4.02 -> 3.92, C5's share 84.3 -> 83.0, C2 +0.3 % (RT_TEMPORAL_QUEUES: A/B).
#ifndef RT_TEMPORAL_QUEUES
#define RT_Q_LD(ptr) __builtin_nontemporal_load(ptr)
#define RT_Q_ST(ptr, val) __builtin_nontemporal_store(val, ptr)
#else
#define RT_Q_LD(ptr) (*(ptr))
#define RT_Q_ST(ptr, val) (*(ptr) = (val))
#endif
union RayBits {
    RayStore r;
    rt_q4 v[kRayPieces];
    __device__ RayBits() {}
};
__device__ __forceinline__ void unpack_head(const RayStore& t, RayRec& r) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.o[k] = t.o[k]; r.d[k] = t.d[k]; }
    r.time = fabs(t.time_kind);
    r.kind = (short)(__double_as_longlong(t.time_kind) < 0 ? kKindShadow : kKindRadiance);
    r.rng = t.word;
}
__device__ __forceinline__ void pack_head(const RayRec& r, RayStore& t) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { t.o[k] = r.o[k]; t.d[k] = r.d[k]; }
    t.time_kind = r.kind == kKindShadow ? -r.time : r.time;
    t.word = r.rng;
}
union HitBits {
    HitRec h;
    rt_q4 v[2];
    __device__ HitBits() {}
};
__device__ __forceinline__ RayRec ray_load_full(const RayQ& q, unsigned i) {
    RayBits b;
#pragma unroll
    for (int k = 0; k < kRayPieces; ++k) b.v[k] = RT_Q_LD(&q.p[(size_t)k * q.stride + i]);
    RayRec r;
    unpack_head(b.r, r);
    r.w[0] = b.r.w[0]; r.w[1] = b.r.w[1]; r.w[2] = b.r.w[2];
    r.pixel = b.r.pixel; r.depth = b.r.depth;
    return r;
}
// What a traversal kernel needs of a queued ray: origin, direction, time, kind, tmax (the head; a primary record is nothing else).
__device__ __forceinline__ RayRec ray_load_head(const RayQ& q, unsigned i) {
    RayBits b;
#pragma unroll
    for (int k = 0; k < kRayHeadPieces; ++k) b.v[k] = RT_Q_LD(&q.p[(size_t)k * q.stride + i]);
    RayRec r;
    unpack_head(b.r, r);
    r.w[0] = r.w[1] = r.w[2] = 1.0;
    r.pixel = 0; r.depth = 0;
    return r;
}
__device__ __forceinline__ void ray_store(const RayQ q, unsigned i, const RayRec& r) {
    RayBits b;
    pack_head(r, b.r);
    b.r.w[0] = r.w[0]; b.r.w[1] = r.w[1]; b.r.w[2] = r.w[2];
    b.r.pixel = r.pixel; b.r.depth = r.depth; b.r.pad = 0;
#pragma unroll
    for (int k = 0; k < kRayPieces; ++k) RT_Q_ST(&q.p[(size_t)k * q.stride + i], b.v[k]);
}
// uv: the scene has a texture or a bump map (DScene::features, the same for the kernel that writes and the one that reads); without,
// the second piece is neither written nor read: 16 bytes less each way per traced ray.
__device__ __forceinline__ bool hits_carry_uv(const DScene& S) { return (S.features & RT_FEAT_TEXTURE) != 0; }
__device__ __forceinline__ HitRec hit_load(const HitQ q, unsigned i, bool uv) {
    HitBits b;
    b.v[0] = RT_Q_LD(&q.p[i]);
    if (uv) b.v[1] = RT_Q_LD(&q.p[(size_t)q.stride + i]);
    else b.v[1] = rt_q4{0u, 0u, 0u, 0u};
    return b.h;
}
__device__ __forceinline__ void hit_store(const HitQ q, unsigned i, const HitRec& h, bool uv) {
    HitBits b;
    b.h = h;
    RT_Q_ST(&q.p[i], b.v[0]);
    if (uv) RT_Q_ST(&q.p[(size_t)q.stride + i], b.v[1]);
}

// Shadow rays do not travel through the ray queues.  A radiance hit writes ONE bundle per light -- what its
// 2 x shadow_samples shadow rays have in common (shading.h:79-101 evaluated for the diffuse and for the specular term,
// shading.h:154,192) -- and k_shadow traces the bundle's rays with one lane each: 128 bytes per (hit, light) instead of
// 2 x samples x (112-byte ray + 32-byte hit record), written once and read by neighbouring lanes.
struct alignas(16) Bundle {
    double p[3];       // the shaded point: directions and distances run from here (shading.h:86-90)
    double org[3];     // ray origin p + n * eps (shading.h:93)
    double wd[3];      // what one ray of the diffuse evaluation adds to the pixel when it reaches the light
    double ws[3];      // ... of the specular evaluation
    double time;
    unsigned long long rng;
    int pixel, light;
    int depth, pad;
};
static_assert(sizeof(Bundle) == 128, "Bundle size");
struct BundleQ {
    rt_q4* p;
    unsigned stride;
};
union BundleBits {
    Bundle b;
    rt_q4 v[8];
    __device__ BundleBits() {}
};
__device__ __forceinline__ Bundle bundle_load(const BundleQ q, unsigned i) {
    BundleBits b;
#pragma unroll
    for (int k = 0; k < 8; ++k) b.v[k] = RT_Q_LD(&q.p[(size_t)k * q.stride + i]);
    return b.b;
}
__device__ __forceinline__ void bundle_store(const BundleQ q, unsigned i, const Bundle& v) {
    BundleBits b;
    b.b = v;
#pragma unroll
    for (int k = 0; k < 8; ++k) RT_Q_ST(&q.p[(size_t)k * q.stride + i], b.v[k]);
}


// Sharded queues.  The tail of a queue is a word that every emitting wave adds to with a returning atomic, and one word
// takes ~88 such atomics per microsecond (MI355X_MICROARCH.md, dequeue row): a frame of small waves of work -- one hit,
// one bundle, a child or two per lane -- spends most of its shading time queueing for that word (measured: 532 -> 170 us
// for the 2 M hits of the C1 frame with the atomics spread over 32 words).  So a queue is cut into `shards` regions of
// `region` entries, each with a tail of its own (on a 128-byte line of its own), and a wave emits into the shard its
// input came from (or, reading a dense queue, into shard wave % shards): the children of a wave stay together, the load
// per shard is bounded (one shard's input times the fan-out), and 32 tails take 32 times the atomics.
#ifndef RT_SHARDS
#define RT_SHARDS 32
#endif
constexpr unsigned kShards = RT_SHARDS;          // shards of a queue when the scene is small (frames of many small levels)
constexpr unsigned kCountStride = 32;     // words between the tails of neighbouring shards
constexpr unsigned kCountArrays = 96;     // one array of tails per ray-tree level (reused modulo this)
constexpr size_t kShardCountBytes = (size_t)(kCountArrays + 1) * kShards * kCountStride * sizeof(unsigned);   // + 1: the bundles' tails
struct QIn {                              // what a consumer kernel works through
    const unsigned* counts;               // tails of the shards (word r * kCountStride), or null: `n` entries from index 0 (host count)
    unsigned n;
    unsigned shards, region;              // entries of shard r: [r * region + lo, r * region + min(hi, tail))
    unsigned lo, hi;
};
// A tail as it stands in L2, where the atomics that count into it are performed: k_tail reads tails its own workgroup has just
// counted into, and a line of this CU's L1 may be older than that.
__device__ __forceinline__ unsigned tail_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
struct BlockSpan {
    unsigned shard, origin;               // this workgroup's shard and the queue index of its entry 0
    unsigned begin, end, step;            // the workgroup's first entry (lane 0), the end of the shard's entries, entries between rounds
};
// Workgroup b of a grid that is a multiple of `shards` serves shard b % shards, striding over it with the others.
// `per`: lanes per entry (k_shadow: one lane per shadow ray of a bundle); begin / end / step are then in lanes.
__device__ __forceinline__ BlockSpan block_span(const QIn& q, unsigned per = 1u) {
    BlockSpan b;
    const unsigned k = q.counts ? q.shards : 1u;
    b.shard = blockIdx.x % k;
    const unsigned j = blockIdx.x / k, nb = gridDim.x / k;
    const unsigned tail = q.counts ? min(tail_load(&q.counts[b.shard * kCountStride]), q.region) : q.n;
    b.origin = q.counts ? b.shard * q.region : 0u;
    b.end = min(tail, q.hi) * per;
    b.begin = q.lo * per + j * kBlock;
    b.step = nb * kBlock;
    return b;
}
__device__ __forceinline__ unsigned queue_total(const QIn& q) {   // entries in all shards (one lane's loop)
    if (!q.counts) return q.n;
    unsigned t = 0;
    for (unsigned r = 0; r < q.shards; ++r) t += min(q.counts[r * kCountStride], q.region);
    return t;
}

// The accumulators are three planes (all reds, all greens, all blues of the part): float atomics execute at the memory
// side as 64-byte requests (MI355X_MICROARCH.md, global float atomics), so what an atomic instruction costs is the number
// of 64-byte segments its lanes touch; neighbouring lanes hold neighbouring pixels, and a plane puts their 8-byte sums
// side by side (8 pixels per request) where interleaved RGB spreads them over three times as many segments.
__device__ __forceinline__ void accum_add(double* accum, const DParams& P, int pixel, D3 c) {
    const size_t plane = (size_t)P.frames * P.slice_rows * P.width;
    unsafeAtomicAdd(&accum[pixel], c.x);
    unsafeAtomicAdd(&accum[plane + pixel], c.y);
    unsafeAtomicAdd(&accum[2 * plane + pixel], c.z);
}

// Reserve one output slot for every lane that calls this (wave-aggregated: one atomic per wave and call site).
__device__ __forceinline__ unsigned emit_slot(unsigned* count) {
    const unsigned long long m = __ballot(1);
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned)__popcll(m));
    base = __shfl(base, leader);
    return base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
}

__device__ __forceinline__ void emit_ray(const RayQ out, unsigned* count, unsigned cap, unsigned* overflow, D3 o, D3 d, D3 w, double time,
                                         double tmax, unsigned long long rng, int pixel, int depth, int kind) {
    unsigned slot = emit_slot(count);
    if (slot >= cap) { *overflow = 1u; return; }   // the host sizes batches so that this cannot happen; it checks the flag
    RayRec r;
    r.o[0] = o.x; r.o[1] = o.y; r.o[2] = o.z;
    r.d[0] = d.x; r.d[1] = d.y; r.d[2] = d.z;
    r.w[0] = w.x; r.w[1] = w.y; r.w[2] = w.z;
    r.time = time; r.pixel = pixel; r.depth = (short)depth; r.kind = (short)kind;
    if (kind == kKindShadow) r.tmax_shadow = tmax; else r.rng = rng;
    ray_store(out, slot, r);
}

// Several rays per lane with ONE atomic per wave (the queue's tail is a single word: returning atomics on it serialise
// in L2, and one atomic per emitted ray and wave made that word the bottleneck of shading-heavy frames).
// emit_reserve: every calling lane gets `per_lane` slots (the same number in all lanes); ray j of this lane goes to
// first + j * lanes, so that the lanes of a wave still write neighbouring slots.
__device__ __forceinline__ unsigned emit_reserve(unsigned* count, unsigned per_lane, unsigned& lanes) {
    const unsigned long long m = __ballot(1);
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    lanes = (unsigned)__popcll(m);
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(count, lanes * per_lane);
    base = __shfl(base, leader);
    return base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
}
__device__ __forceinline__ void emit_ray_at(const RayQ out, unsigned slot, unsigned cap, unsigned* overflow, D3 o, D3 d, D3 w, double time,
                                            double tmax, unsigned long long rng, int pixel, int depth, int kind) {
    if (slot >= cap) { *overflow = 1u; return; }
    RayRec r;
    r.o[0] = o.x; r.o[1] = o.y; r.o[2] = o.z;
    r.d[0] = d.x; r.d[1] = d.y; r.d[2] = d.z;
    r.w[0] = w.x; r.w[1] = w.y; r.w[2] = w.z;
    r.time = time; r.pixel = pixel; r.depth = (short)depth; r.kind = (short)kind;
    if (kind == kKindShadow) r.tmax_shadow = tmax; else r.rng = rng;
    ray_store(out, slot, r);
}

// ------------------------------------------------------------------------------------------ k_raygen
// Linear index -> pixel of the chunk, in 8-row bands of 8-wide tiles, so that 64 consecutive rays form an 8x8 tile.
// (All of it in unsigned 32-bit arithmetic: a chunk holds fewer than 2^32 rays.  The divisors are the same for every lane and
// mostly powers of two -- 8-wide tiles, 8-row bands and tiles, 1 or 16 frames, 1 sample -- which udiv_u settles with a shift behind
// a scalar branch; the compiler's 64-bit division by a run-time number is some ninety instructions, and this runs once per
// primary ray in k_raygen and once more in the kernel that shades it.)
__device__ __forceinline__ unsigned udiv_u(unsigned a, unsigned d) { return (d & (d - 1u)) == 0u ? a >> __builtin_ctz(d) : a / d; }
__device__ __forceinline__ void pixel_of(unsigned pix, int W, int rows, int& x, int& ly) {
    const int nb = (rows + 7) / 8;
    int b = (int)udiv_u(pix, (unsigned)W * 8u);
    if (b > nb - 1) b = nb - 1;
    const int hb = (b == nb - 1) ? rows - 8 * b : 8;
    const int j = (int)(pix - (unsigned)b * (unsigned)W * 8u);
    const int tiles_x = (W + 7) / 8;
    int tx = hb == 8 ? j >> 6 : j / (8 * hb);
    if (tx > tiles_x - 1) tx = tiles_x - 1;
    const int rem = j - tx * 8 * hb;
    const int tw = (8 * tx + 8 <= W) ? 8 : W - 8 * tx;
    const int q = tw == 8 ? rem >> 3 : rem / tw;
    ly = b * 8 + q;
    x = tx * 8 + (rem - q * tw);
}

// Index i of a chunk's primary rays -> pixel (x, y) of the frame, sample, frame of the launch, and the pixel's index in the
// output / the accumulators.  Row of the launch -> (frame, compact row of the part): with several frames the 8-row bands of
// the frames alternate (band b of every frame, then band b + 1 ...), so that a contiguous share of the queue -- what the
// waves of one XCD traverse -- covers the same region of the image in all frames; parts whose rows are no multiple of 8 go
// frame by frame.
struct PrimaryCoords { int x, y, s, frame, pixel; };
__device__ __forceinline__ PrimaryCoords primary_coords(const DParams& P, const Chunk& ck, unsigned i) {
    PrimaryCoords c;
    const unsigned pix = udiv_u(i, (unsigned)ck.ns);
    c.s = ck.s0 + (int)(i - pix * (unsigned)ck.ns);
    int ly;
    pixel_of(pix, P.width, ck.rows, c.x, ly);
    int lr = ck.row0 + ly;
    c.frame = 0;
    if (P.frames > 1) {
        if ((P.part_rows & 7) == 0) {
            const unsigned band = (unsigned)lr >> 3;
            const unsigned fb = udiv_u(band, (unsigned)P.frames);
            c.frame = (int)(band - fb * (unsigned)P.frames);
            lr = (int)fb * 8 + (lr & 7);
        } else {
            c.frame = (int)udiv_u((unsigned)lr, (unsigned)P.part_rows);
            lr -= c.frame * P.part_rows;
        }
    }
    const unsigned tile = udiv_u((unsigned)lr, (unsigned)P.tile_rows);
    c.y = ((int)tile * P.part_count + P.part_index) * P.tile_rows + (lr - (int)tile * P.tile_rows);
    c.pixel = (c.frame * P.slice_rows + lr) * P.width + c.x;
    return c;
}
// A queued ray: a full record, or a slim primary record completed by what is implied.
__device__ __forceinline__ RayRec ray_load(const RayQ& q, unsigned i, const DParams& P) {
    if (!q.primary) return ray_load_full(q, i);
    RayRec r = ray_load_head(q, i);   // weight 1, radiance
    r.pixel = primary_coords(P, q.ck, i).pixel;
    r.depth = (short)P.max_depth;
    return r;
}

__global__ void __launch_bounds__(kBlock) k_raygen(const DScene S, const DParams P, const RayQ out, long long n,
                                                   unsigned* __restrict__ fetch_counters) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    // the traversal launched next on this stream starts with its queue fetch counters at zero (saves a memset launch)
    if (blockIdx.x == 0 && threadIdx.x < kFetchCounterWords) fetch_counters[threadIdx.x] = 0u;
    if (i >= n) return;
    const PrimaryCoords c = primary_coords(P, out.ck, i);
    Rng rng;  // main.cpp:288-302
    rng.d0 = (P.rng_mode == RT_RNG_D0);
    rng.s = ((P.seed + (unsigned long long)c.frame) * 0x9E3779B97F4A7C15ull) ^
            (((unsigned long long)c.y * P.width + c.x) * 0xD1B54A32D192ED03ull + (unsigned long long)c.s * 0x8CB92BA72F3D8DD7ull);
    float ru = (float)rng.uniform(), rv = (float)rng.uniform();
    float px = ((float)c.x + ru) / (float)P.width;
    float py = ((float)c.y + rv) / (float)P.height;
    double time = rng.uniform() * S.shutter_time;
    Ray ray = generate_ray(S.cam, px, py, time, rng);
    RayBits b;   // a primary record is the head alone
    b.r.o[0] = ray.o.x; b.r.o[1] = ray.o.y; b.r.o[2] = ray.o.z;
    b.r.d[0] = ray.d.x; b.r.d[1] = ray.d.y; b.r.d[2] = ray.d.z;
    b.r.time_kind = time;
    b.r.word = rng.next();
#pragma unroll
    for (int k = 0; k < kRayHeadPieces; ++k) RT_Q_ST(&out.p[(size_t)k * out.stride + (unsigned)i], b.v[k]);
}

// ------------------------------------------------------------------------------------------ k_traverse
#ifndef RT_TRAVERSE_WAVES
#define RT_TRAVERSE_WAVES 4       // waves per SIMD asked of the register allocator
#endif
#ifndef RT_TRAVERSE_WAVES_SPHERES
#define RT_TRAVERSE_WAVES_SPHERES 5   // ... for the spheres-only variant, which fits 96 registers (its workgroups' LDS stacks
#endif                                // fit five to a CU with 14 entries per lane): 7 % faster than 4 on the launch of 16 C2 frames
#ifndef RT_TRAVERSE_WAVES_FP32
#define RT_TRAVERSE_WAVES_FP32 5      // ... for the spheres-only variant with the FP32 sphere test
#endif
template <int F> constexpr int traverse_waves() {
    return F == (RT_FEAT_SPHERE | RT_FEAT_FP32) ? RT_TRAVERSE_WAVES_FP32 : F == RT_FEAT_SPHERE ? RT_TRAVERSE_WAVES_SPHERES : RT_TRAVERSE_WAVES;
}
#ifndef RT_NODE_VOTE
#define RT_NODE_VOTE 24           // run an interior-node step while at least this many lanes want one
#endif
#ifndef RT_RAYS_PER_LANE
#define RT_RAYS_PER_LANE 2        // the traversal grid shrinks until every lane has this many rays to work through (default;
                                  // environment RT_RAYS_PER_LANE overrides: larger = smaller grids = more launches side by side)
#endif
#ifndef RT_INNER_VOTE
#define RT_INNER_VOTE 48          // stay in the node loop while at least this many lanes want another node
#endif
#ifndef RT_LEAF_VOTE
#define RT_LEAF_VOTE 8            // test the parked leaves once this many lanes hold one
#endif
#ifndef RT_LEAF_VOTE_FP32
#define RT_LEAF_VOTE_FP32 8       // ... in the variant with the FP32 sphere test
#endif
#ifndef RT_CHUNK
#define RT_CHUNK 64               // rays a wave reserves from the queue per atomic
#endif
#ifndef RT_REFILL_VOTE
#define RT_REFILL_VOTE 16         // fetch new rays once this many lanes are idle (8: 3.5 % slower, 24: 6 % slower on the C2 frame)
#endif

// The queue is cut into one contiguous share per XCD, each with its own fetch counter (on its own cache line): the
// waves of an XCD work through neighbouring rays, so that the part of the hierarchy they touch stays in that XCD's
// 4 MB L2, and take from the other shares only once their own is empty.
constexpr unsigned kXcds = 8;
constexpr unsigned kNextStride = kFetchCounterWords / kXcds;   // counters 128 bytes apart
constexpr unsigned kNextBase = 32;     // offset of the fetch counters in WavefrontWorkspace::counters
constexpr unsigned kLevelCountBase = 4;   // two queue-size words of the levels enqueued without a host round trip
constexpr unsigned kLevelTotal = 6;       // u64: rays of the levels whose size only the device knew (words 2 .. 7 are read back as one piece)
constexpr unsigned kLevelSizeBase = 8, kLevelSizeWords = 24;   // small scenes: entries of the queue of level l, l < 24 (words 2 .. 31 are read back as one piece)
__device__ __forceinline__ unsigned share_begin(unsigned n, unsigned q) {
    return q >= kXcds ? n : (unsigned)(((unsigned long long)n * q / kXcds) & ~63ull);
}

// History order of a queue's chunks.  A launch of the persistent kernel ends with a tail: the last rays to START include long
// ones (the longest ray of the 10^6-ellipsoid frame takes 361 node steps against a mean of 47), and while they finish most of
// the machine idles -- 0.4 of the 1.0 ms of a one-frame launch.  Which rays are long cannot be known in advance, but frames of a
// sequence resemble each other: k_traverse records per chunk (64 rays = one 8x8 tile) the most node steps one of its rays took,
// when that is well above the mean, and k_chunk_order turns the record into the order in which every XCD's share hands out
// its chunks next time -- longest first (counting sort into 32 buckets by step count; order inside a bucket as recorded).
// Scheduling only: every ray is traced exactly as before.  The record belongs to the workspace of a stream and is used only
// for a queue of the same scene, size and frame layout (Driver::chunk_history).
#ifndef RT_LONG_RAY_STEPS
#define RT_LONG_RAY_STEPS 64      // (48 .. 64 measure alike, 96 and 128 worse: one-frame launch 0.826 / 0.837 / 0.871 ms)
#endif
#ifndef RT_ORDER_BUCKET_SHIFT
#define RT_ORDER_BUCKET_SHIFT 3   // buckets of 8 steps
#endif
constexpr unsigned kLongRaySteps = RT_LONG_RAY_STEPS;
constexpr unsigned kOrderBuckets = 32;
__global__ void __launch_bounds__(kBlock) k_chunk_order(unsigned* __restrict__ cost, unsigned* __restrict__ order, const unsigned n_rays) {
    __shared__ unsigned hist[kOrderBuckets], offset[kOrderBuckets];
    const unsigned q = blockIdx.x;                                   // one workgroup per XCD share of the queue
    const unsigned lo = share_begin(n_rays, q) / (unsigned)RT_CHUNK, hi = share_begin(n_rays, q + 1u) / (unsigned)RT_CHUNK;
    if (threadIdx.x < kOrderBuckets) hist[threadIdx.x] = 0u;
    __syncthreads();
    // bucket 0 holds the longest (steps >= 64 + 30 * 8), the last bucket the chunks without a long ray
    auto bucket_of = [](unsigned c) { return c < kLongRaySteps ? kOrderBuckets - 1u : (kOrderBuckets - 2u) - min(kOrderBuckets - 2u, (c - kLongRaySteps) >> RT_ORDER_BUCKET_SHIFT); };
    for (unsigned i = lo + threadIdx.x; i < hi; i += kBlock) atomicAdd(&hist[bucket_of(cost[i])], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned run = 0;
        for (unsigned b = 0; b < kOrderBuckets; ++b) { offset[b] = run; run += hist[b]; }
    }
    __syncthreads();
    for (unsigned i = lo + threadIdx.x; i < hi; i += kBlock) {
        order[lo + atomicAdd(&offset[bucket_of(cost[i])], 1u)] = i;
        cost[i] = 0u;                                                // the next launch records anew
    }
}

// COUNT: also count box tests / node fetches / primitive tests (the statistics pass of bench.py and RtStats); the
// timed variant carries neither the registers nor the instructions for them.
template <int F, bool COUNT>
__global__ void __launch_bounds__(kBlock, traverse_waves<F>())
k_traverse(const DScene S, const DParams P, const RayQ rays, const HitQ hits, const unsigned n_host,
           const unsigned* __restrict__ n_device, const unsigned rays_per_block, unsigned* __restrict__ next,
           unsigned long long* __restrict__ counters, unsigned long long* __restrict__ level_total,
           const unsigned* __restrict__ chunk_order, unsigned* __restrict__ chunk_cost) {
    __shared__ unsigned long long stack_mem[kTravLdsEntries + 1][kBlock];   // + 1: the row branch-free pushes discard into
    // n_device != NULL: the queue was filled by the shading kernel launched before this one and the host has not read
    // its size (levels enqueued without a round trip); n_host is then the queue's capacity, which a count that
    // overflowed is cut to.  One lane adds the level's rays to the frame's total.
    const unsigned n = n_device ? min(*n_device, n_host) : n_host;
    if (n_device && blockIdx.x == 0 && threadIdx.x == 0 && n != 0u) atomicAdd(level_total, (unsigned long long)n);
    // The host sized the grid by the bound it knows; workgroups beyond what the actual count calls for (the host's own
    // rule: rays_per_block rays each) leave at once -- thousands of idle waves polling the eight fetch counters would
    // serialise on them for longer than a small level takes to trace.
    if (n == 0u || (n_device && blockIdx.x * (unsigned long long)rays_per_block >= n)) return;
    TravStack stack;
    stack.lds = &stack_mem[0][threadIdx.x];
    stack.stride = kBlock;
    const int lane = threadIdx.x & 63;
    constexpr int kDone = 0x7fffffff;
    constexpr unsigned kNone = 0xffffffffu;

    bool exhausted = false;
    unsigned my = kNone;                               // queue index of this lane's ray; kNone = idle lane
    unsigned chunk_next = 0, chunk_end = 0;            // this wave's reserved rays [chunk_next, chunk_end)
    unsigned steps = 0;                                // node steps of this lane's ray (what chunk_cost records)
    const unsigned xcd = blockIdx.x & (kXcds - 1u);     // workgroups are dealt to the XCDs round-robin
    unsigned victim = 0;                               // shares of the queue found empty so far
    constexpr bool kFp32 = (F & RT_FEAT_FP32) != 0;   // the throughput variant: FP32 sphere test (test_sphere32), the ray kept in FP32
    Ray r{mk(0, 0, 0), mk(0, 0, 1), 0.0};
    Ray32 r32{{0.f, 0.f, 0.f}, {0.f, 0.f, 1.f}, 0.f};
    RayF rf{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0u, 0u, 0u}};
    double tmin = 0.0;
    float tmin32 = 0.f, tmax32 = 0.f;   // tmax32 follows best.t (rounded up)
    Cand best{0.0, -1, 0, 0.0, 0.0};
    int best_order = 0, sp = 0, cur = kDone, pend = 0;   // pend: parked leaf reference (< 0) or 0; idle: cur == kDone, pend == 0
    bool best_incl = false;
    bool any_hit = false;   // shadow ray in a scene without transmitting materials: any occluder ends it (shading.h:52-55)
    Counters C{0u, 0u, 0u, 0u};
    unsigned ray_visits = 0, max_visits = 0, iters = 0;   // statistics (COUNT only)

    while (true) {
        if (COUNT) ++iters;
        // ---- refill idle lanes with the next rays of the queue.  A wave reserves RT_CHUNK consecutive rays with one
        //      atomic and hands them to its lanes as they fall idle (one atomic per refill on a single counter serialises
        //      in L2 and was the bottleneck of the whole kernel).
        const unsigned long long idle = __ballot(my == kNone);
        if (!exhausted && idle != 0ull && (__popcll(idle) >= RT_REFILL_VOTE || idle == ~0ull)) {
            while (chunk_next >= chunk_end) {   // wave-uniform: this XCD's share of the queue first, then the others'
                const unsigned qx = (xcd + victim) & (kXcds - 1u);
                const unsigned lo = share_begin(n, qx), hi = share_begin(n, qx + 1u);
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&next[qx * kNextStride], (unsigned)RT_CHUNK);
                base = __shfl(base, 0);
                if (base < hi - lo) {
                    chunk_next = lo + base;
                    chunk_end = (hi - lo - base < (unsigned)RT_CHUNK) ? hi : chunk_next + (unsigned)RT_CHUNK;
                    // History order (k_chunk_order): the c-th chunk a share hands out is the one that held the longest rays when
                    // this queue was last traced, so the rays started last are the short ones and the launch ends without
                    // a tail of long rays started late.  (Only whole chunks: the host passes an order when n is a multiple of RT_CHUNK.)
                    if (chunk_order) {
                        chunk_next = chunk_order[chunk_next / (unsigned)RT_CHUNK] * (unsigned)RT_CHUNK;
                        chunk_end = chunk_next + (unsigned)RT_CHUNK;
                    }
                } else if (++victim == kXcds) {
                    exhausted = true;
                    break;
                }
            }
            const unsigned avail = chunk_end - chunk_next;
            const unsigned rank = (unsigned)__popcll(idle & ((1ull << lane) - 1ull));
            const unsigned take = min((unsigned)__popcll(idle), avail);
            if (my == kNone && rank < take) {
                my = chunk_next + rank;
                const RayRec q = ray_load_head(rays, my);
                r.o = ld3(q.o);
                r.d = ld3(q.d);
                r.time = q.time;
                if constexpr (kFp32) r32 = make_ray32(r);
                const bool shadow = q.kind == kKindShadow;
                any_hit = shadow && S.transparent_materials == 0;
                tmin = shadow ? 0.001 : P.eps_scene;                       // shading.h:45 / tracer.h:152-156
                const double tmax = shadow ? q.tmax_shadow - 0.001 : 100000.0;
                best = Cand{tmax, -1, 0, 0.0, 0.0};
                best_order = 0;
                best_incl = false;
                sp = 0;
                cur = kDone;
                pend = 0;
                steps = 0;
                if (S.n_prims > 0) {
                    if (P.use_bvh) {
                        rf = make_rayf(r);
                        tmin32 = __double2float_rd(tmin);
                        float e;
                        if (COUNT) C.nodes++;
                        tmax32 = __double2float_ru(tmax);
                        if (box_test32(S.root_lo, S.root_hi, rf, tmin32, tmax32, e)) cur = S.root_ref;
                    } else {  // --no-bvh: the list in scene-file order (hittable_list.cpp:8-33)
                        if constexpr (kFp32) {
                            for (int i = 0; i < S.n_prims; ++i) test_sphere32<COUNT>(S, P, r32, S.file_order[i], (float)tmin, best, best_order, best_incl, C);
                        } else {
                            for (int i = 0; i < S.n_prims; ++i) test_prim<F, COUNT>(S, P, r, S.file_order[i], i, tmin, best, best_order, best_incl, C);
                        }
                    }
                }
            }
            chunk_next += take;
        }
        if (__ballot(my != kNone) == 0ull) {
            if (exhausted) break;
            continue;
        }
        // ---- a lane that reached a leaf parks it (one slot) and goes on with its next deferred node, so that node steps
        //      keep their lanes and the FP64 primitive tests run for many lanes at once.  The parked leaf may shorten
        //      the ray later than in a strict traversal, which only costs node visits, never a hit.
        if (cur < 0 && pend == 0) {
            pend = cur;
            cur = stack.pop(sp, tmax32);
        }
        bool is_node = static_cast<unsigned>(cur) < static_cast<unsigned>(kDone);
        int nn = __popcll(__ballot(is_node));
        const int nl = __popcll(__ballot(pend != 0));
        if (nn != 0 && nl < (kFp32 ? RT_LEAF_VOTE_FP32 : RT_LEAF_VOTE) && (nn >= RT_NODE_VOTE || nn >= nl)) {
            // ---- interior nodes; while most of the wave still wants one, stay here (no scheduling work per step)
            do {
#ifdef RT_LANE_STATS
                if (COUNT) {   // experiment build (scratch/lane_stats.py): lanes of a node step by state -> counters 5 (idle), 6 (no node work), 7 (steps)
                    const unsigned long long b_idle = __ballot(my == kNone), b_leaf = __ballot(my != kNone && !is_node);
                    if (lane == 0) {
                        atomicAdd(&counters[5], (unsigned long long)__popcll(b_idle));
                        atomicAdd(&counters[6], (unsigned long long)__popcll(b_leaf));
                        atomicAdd(&counters[7], 1ull);
                    }
                }
#endif
                if (is_node) {
                    const NodeRegs nd = load_node(&S.nodes[cur]);
                    ++steps;
                    if (COUNT) { C.nodes += node4_children(nd); C.visits++; ray_visits++; }
                    cur = node4_step<false>(nd, rf, tmin32, tmax32, stack, sp);
                    // one pop loop per step for both kinds of lanes that need their next deferred node: no child entered, or
                    // the nearest child is a leaf, which is parked
                    bool need = cur == kNoChild;
                    if (cur < 0 && pend == 0) { pend = cur; need = true; }
                    if (need) cur = stack.pop_parking(sp, tmax32, pend);
                }
                is_node = static_cast<unsigned>(cur) < static_cast<unsigned>(kDone);
                nn = __popcll(__ballot(is_node));
                if (COUNT) ++iters;
            } while (nn >= RT_INNER_VOTE);
        } else if (nl != 0) {
            // ---- the parked leaves
            if (pend != 0) {
                const int code = ~pend;
                const int first = code >> 1;
                if constexpr (kFp32) {
                    test_sphere32<COUNT>(S, P, r32, first, (float)tmin, best, best_order, best_incl, C);
                    if (code & 1) test_sphere32<COUNT>(S, P, r32, first + 1, (float)tmin, best, best_order, best_incl, C);
                } else {
                    test_prim<F, COUNT>(S, P, r, first, first, tmin, best, best_order, best_incl, C);
                    if (code & 1) test_prim<F, COUNT>(S, P, r, first + 1, first + 1, tmin, best, best_order, best_incl, C);
                }
                tmax32 = __double2float_ru(best.t);
                pend = 0;
                // The shading of an occluded shadow ray only asks whether its occluder transmits: with opaque materials
                // only, the first occluder found answers for the closest one, and the ray is done.
                if (any_hit && best.prim >= 0) { cur = kDone; sp = 0; }
            }
        }
        // ---- retire finished rays
        if (my != kNone && cur == kDone && pend == 0) {
            HitRec h;
            h.t = best.t; h.bu = best.bu; h.bv = best.bv; h.prim = best.prim; h.tri = best.tri;
            hit_store(hits, my, h, hits_carry_uv(S));
            // a ray that took more than the mean number of node steps (~47 on the 10^6-ellipsoid frame) marks its chunk as a long one
            if (chunk_cost && steps >= kLongRaySteps) atomicMax(&chunk_cost[my / (unsigned)RT_CHUNK], steps);
            my = kNone;
            if (COUNT) { max_visits = max(max_visits, ray_visits); ray_visits = 0; }
        }
    }
    if (COUNT && counters) {
        unsigned nodes = C.nodes, prims = C.prims, visits = C.visits;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            nodes += __shfl_down(nodes, off);
            prims += __shfl_down(prims, off);
            visits += __shfl_down(visits, off);
            max_visits = max(max_visits, (unsigned)__shfl_down(max_visits, off));
        }
        if (lane == 0) {
#ifndef RT_LANE_STATS
            atomicMax(&counters[5], (unsigned long long)max_visits);
            atomicMax(&counters[6], (unsigned long long)iters);
#endif
            atomicAdd(&counters[4], (unsigned long long)visits);
            atomicAdd(&counters[1], (unsigned long long)nodes);
            atomicAdd(&counters[2], (unsigned long long)prims);
        }
    }
}

// ------------------------------------------------------------------------------------------ k_shade
struct EmitCtx {
    RayQ out;
    unsigned* count;
    unsigned* overflow;
    unsigned cap;
    uint8_t* direct;   // non-null: one ray per pixel and no children (1 spp, nothing spawns rays): the colour of a ray is
                       // the pixel, written tone-mapped straight into the frame instead of summed and resolved later
    BundleQ bundles;   // shadow bundles of this launch's hits (k_shadow traces them), their count, the buffer's capacity
    unsigned* bundle_count;
    unsigned bundle_cap;
    unsigned shards;   // `count` and `bundle_count` are arrays of this many tails; `cap` / `bundle_cap` are one shard's entries
};
// The emission context of one wave: out queue, bundle buffer and their tails narrowed to shard `r`.
__device__ __forceinline__ EmitCtx emit_view(const EmitCtx& E, unsigned r) {
    EmitCtx V = E;
    r %= E.shards;
    V.out.p = E.out.p ? E.out.p + (size_t)r * E.cap : nullptr;
    V.count = E.count + r * kCountStride;
    V.bundles.p = E.bundles.p ? E.bundles.p + (size_t)r * E.bundle_cap : nullptr;
    V.bundle_count = E.bundle_count + r * kCountStride;
    return V;
}

// Contribution c of one ray to its pixel.
__device__ __forceinline__ void deposit(const EmitCtx& E, const DParams& P, double* accum, int pixel, D3 c) {
    if (E.direct) {
        const D3 m = tonemap(c, P.tonemap);   // == k_resolve with spp 1 (main.cpp:308-316)
        uint8_t* o = E.direct + (size_t)pixel * 3;
        o[0] = quantise(m.x);
        o[1] = quantise(m.y);
        o[2] = quantise(m.z);
    } else {
        accum_add(accum, P, pixel, c);
    }
}

// Shades one traced ray: returns whether it contributes `contrib` to its pixel, and queues its child rays.
// EMIT = false: the frame cannot queue a ray (no material reflects or transmits and shadows are off, or --normals):
// none of the code that does is compiled in, which halves the registers of the kernel that shades such frames.
template <int F, bool EMIT>
__device__ __forceinline__ bool shade_ray(const DScene& S, const DParams& P, const RayRec& q, const HitRec& hr, const EmitCtx& E,
                                          D3& contrib) {
    const Ray r{ld3(q.o), ld3(q.d), q.time};
    const D3 w = ld3(q.w);
    Rng rng{q.rng, P.rng_mode == RT_RNG_D0};
    const Cand c{hr.t, hr.prim, hr.tri, hr.bu, hr.bv};

    if (EMIT && q.kind == kKindShadow) {  // shading.h:42-75, one segment of the transmission chain
        if (c.prim < 0) {  // reached the light sample: this sample's share of the light term
            contrib = w;
            return true;
        }
        const DPrim pr = S.prims[c.prim];
        const DMaterial& M = S.materials[pr.material];
        if (!(M.transparency > 0.0)) return false;  // opaque occluder: transmission 0
        Hit h;
        fill_hit<F>(S, P, r, c, h);
        double n1 = h.front ? 1.0 : M.ior, n2 = h.front ? M.ior : 1.0;
        double eta = n1 / n2;
        double cos_i = -dot(r.d, h.n);
        double sin2 = eta * eta * (1.0 - cos_i * cos_i);
        if (sin2 > 1.0) return false;
        double r0 = ((n1 - n2) / (n1 + n2));
        r0 = r0 * r0;
        double refl = r0 + (1.0 - r0) * pow((1.0 - cos_i), 5.0);  // shadow_schlick, shading.h:24-28
        D3 local = cmul(mk(1.0, 1.0, 1.0), ld3(M.diffuse)) * (1.0 - refl);
        if (length(local) < 0.001) return false;
        emit_ray(E.out, E.count, E.cap, E.overflow, h.p + r.d * 0.001, r.d, cmul(w, local), q.time, q.tmax_shadow - h.t, 0ull, q.pixel,
                 q.depth, kKindShadow);
        return false;
    }

    // ---- radiance ray: tracer.h:147-295
    if (c.prim < 0) {
        D3 bg = (Feat<F>::env && S.env_present) ? env_sample(S, r.d) : ld3(P.background);  // tracer.h:282-293
        contrib = cmul(w, bg);
        return true;
    }
    Hit h;
    fill_hit<F>(S, P, r, c, h);
    if (P.normals) {  // tracer.h:157-161
        contrib = cmul(w, (h.n + mk(1, 1, 1)) * 0.5);
        return true;
    }
    const DMaterial& M = S.materials[h.material];
    const bool transparent = M.transparency > 0;
    bool has_refl = M.reflectivity > 0;
    if (transparent && P.fresnel) has_refl = true;
    const double eps = P.eps_scene;

    // refraction geometry first: the weights of the local terms depend on it (tracer.h:226-266, 100-143)
    double reflect_prob = M.reflectivity, transmit_prob = M.transparency;
    bool do_refract = false, tir_mirror = false;
    D3 refr_dir = mk(0, 0, 0), Vin = mk(0, 0, 0), Nh = mk(0, 0, 0);
    if (transparent) {
        Vin = normalize(r.d);
        Nh = normalize(h.n);
        double n1 = h.front ? 1.0 : M.ior, n2 = h.front ? M.ior : 1.0;
        double eta = n1 / n2;
        double cos_i = -dot(Vin, Nh);
        double sin2 = eta * eta * (1.0 - cos_i * cos_i);
        if (sin2 > 1.0) {
            transmit_prob = 0.0;
            reflect_prob = 1.0;
            tir_mirror = !has_refl;
        } else {
            double cos_t = sqrt(1.0 - sin2);
            refr_dir = (eta * Vin) + (eta * cos_i - cos_t) * Nh;
            do_refract = true;
            if (P.fresnel) {
                double r0 = ((n1 - n2) / (n1 + n2)) * ((n1 - n2) / (n1 + n2));  // schlick, tracer.h:88-97
                double x = 1.0 - cos_i;
                reflect_prob = r0 + (1.0 - r0) * (x * x * x * x * x);
                transmit_prob = 1.0 - reflect_prob;
            }
        }
    }
    // weight of the ambient+diffuse term and of the reflected colour in the final mix (tracer.h:268-279)
    const double ad_weight = transparent ? 0.0 : (1.0 - M.reflectivity);
    const double refl_weight = transparent ? reflect_prob : M.reflectivity;

    // ---- secondary rays.  They are queued BEFORE the local terms are computed: what the two parts share is the hit, the
    // weight and a few scalars, and the vectors of the refraction / reflection geometry die here instead of staying in
    // registers across the light loop below.  The random stream is the one of the old order: the light loop draws one
    // number per light for its bundles, so the emission starts that many draws further on (splitmix64's state is a counter).
    const int child_depth = q.depth - 1;
    if (EMIT && child_depth > 0) {  // (children of depth 0 would return black without tracing, tracer.h:149)
    Rng rng_e = rng;
    if (P.shadows && S.n_lights > 0) rng_e.s += (unsigned long long)S.n_lights * 0x9E3779B97F4A7C15ull;
    // reflection (tracer.h:176-219): samples below the surface are dropped (tracer.h:196-204), so the kept ones are found
    // first with a copy of the random stream.  All children of the wave's hits then get their slots with one atomic,
    // SAMPLE-major: the lanes that keep sample k take consecutive slots, so every store instruction of the wave writes
    // consecutive 16-byte pieces (a lane's own rays side by side would put the lanes of an instruction 128 bytes and more
    // apart, and the L2 passes such pieces on to memory one by one: measured, 6 x the bytes).
    int samples = 0;
    double rough = 0.0;
    D3 perfect = mk(0, 0, 0), wr = mk(0, 0, 0), org = mk(0, 0, 0);
    unsigned long long keep = 0ull;   // bit k < 62: reflection ray k of this hit; bit 62: TIR mirror ray; bit 63: refraction ray
    if (has_refl) {
        samples = P.glossy_samples;
        if (q.depth < P.max_bounces) samples = 1;
        rough = 1.0 / sqrt(M.shininess);
        D3 Vn = normalize(r.d);
        perfect = normalize(reflect(Vn, h.n));
        wr = w * refl_weight;
        if (M.is_metal) wr = cmul(wr, ld3(M.diffuse));
        org = h.p + h.n * eps;
        if (samples > 0) {
            wr = wr * (1.0 / samples);
            Rng dry = rng_e;
            for (int k = 0; k < samples; ++k) {
                // (tracer.h:197-199 tests the NORMALISED direction; its sign is the unnormalised one's unless the vector is shorter
                // than vector3.h's 1e-6.  The deterministic mode keeps the reference's form to the letter.)
                const D3 v = perfect + dry.in_unit_sphere() * rough;
                const bool above = dry.d0 ? dot(normalize(v), h.n) > 0 : (dot(v, h.n) > 0 && dot(v, v) > 1e-12);
                if (above) { keep |= 1ull << k; (void)dry.next(); }
            }
        } else {
            keep = 1ull;
        }
    }
    if (tir_mirror) keep |= 1ull << 62;
    if (do_refract) keep |= 1ull << 63;
    // how many rays the lanes of the wave that got here emit (ballots count only them), and one atomic for all
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int kmax = has_refl ? (samples > 0 ? samples : 1) : 0;   // reflection slots of this lane
    unsigned total = (unsigned)__popcll(__ballot(tir_mirror)) + (unsigned)__popcll(__ballot(do_refract));
    for (int k = 0; __ballot(k < kmax) != 0ull; ++k) total += (unsigned)__popcll(__ballot((keep >> k) & 1ull));
    const int leader = __ffsll((long long)__ballot(1)) - 1;
    unsigned slot0 = 0;
    if (lane == leader && total != 0u) slot0 = atomicAdd(E.count, total);
    slot0 = __shfl(slot0, leader);
    for (int k = 0; __ballot(k < kmax) != 0ull; ++k) {
        const unsigned long long mk_ = __ballot((keep >> k) & 1ull);
        if (k < kmax) {
            D3 target = perfect;
            const bool mine_k = (keep >> k) & 1ull;   // as the dry pass decided
            if (samples > 0) target = perfect + rng_e.in_unit_sphere() * rough;
            if (mine_k)
                emit_ray_at(E.out, slot0 + (unsigned)__popcll(mk_ & below), E.cap, E.overflow, org, samples > 0 ? normalize(target) : target, wr, q.time,
                            100000.0, rng_e.next(), q.pixel, child_depth, kKindRadiance);
        }
        slot0 += (unsigned)__popcll(mk_);
    }
    {
        const unsigned long long mt = __ballot(tir_mirror);
        if (tir_mirror)  // tracer.h:258-264 (not tinted for metals)
            emit_ray_at(E.out, slot0 + (unsigned)__popcll(mt & below), E.cap, E.overflow, h.p + Nh * eps, normalize(reflect(Vin, Nh)), w * reflect_prob, q.time,
                        100000.0, rng_e.next(), q.pixel, child_depth, kKindRadiance);
        slot0 += (unsigned)__popcll(mt);
        const unsigned long long mr = __ballot(do_refract);
        if (do_refract)  // tracer.h:239-245: starts at the hit point, tinted by diffuse
            emit_ray_at(E.out, slot0 + (unsigned)__popcll(mr & below), E.cap, E.overflow, h.p, normalize(refr_dir), cmul(w * transmit_prob, ld3(M.diffuse)), q.time,
                        100000.0, rng_e.next(), q.pixel, child_depth, kKindRadiance);
    }
    }

    // ---- local shading: shading.h:104-217
    D3 base;
    if (Feat<F>::tex && M.texture >= 0) {
        const DTexture& T = S.textures[M.texture];
        int x = to_int(h.u * (T.width - 1)), y = to_int((1.0 - h.v) * (T.height - 1));
        x = imin(imax(x, 0), T.width - 1);
        y = imin(imax(y, 0), T.height - 1);
        Px p = tex_get(T, x, y);
        base = mk(p.r / 255.0, p.g / 255.0, p.b / 255.0);
    } else {
        base = ld3(M.diffuse);
    }
    const D3 Pt = h.p, N = normalize(h.n);
    const D3 V = normalize(r.o - Pt);
    D3 ad = cmul(cmul(ld3(M.ambient), ld3(P.ambient_light)), base);
    D3 spec = mk(0, 0, 0);
    // With shadows every light's terms reach the pixel through its shadow rays: one bundle per (hit, light), slots for all
    // lights of the hit reserved with one atomic per wave.
    unsigned bundle_lanes = 0, bundle_first = 0;
    if (EMIT && P.shadows && S.n_lights > 0) bundle_first = emit_reserve(E.bundle_count, (unsigned)S.n_lights, bundle_lanes);
    for (int li = 0; li < S.n_lights; ++li) {
        const DLight& L = S.lights[li];
        D3 Lr = ld3(L.pos) - Pt;
        double falloff = 1.0 / dot(Lr, Lr);
        D3 Ld = normalize(Lr);
        D3 I = ld3(L.intensity) * falloff * P.exposure;
        double ldn = smax(0.0, dot(Ld, N));
        D3 diff = cmul(base, I) * ldn;
        diff = diff * (1.0 - M.transparency);                                   // shading.h:166-167
        D3 Hv = normalize(Ld + V);
        double hdn = smax(0.0, dot(Hv, N));
        // shading.h:206, fast_pow :177.  A material without a specular colour multiplies the power by zero: where the power is
        // certain to be a finite number (0 <= hdn <= 1 + a few ulp, 0 < shininess < 1e12: the exponent is -inf or a finite number below 1) any finite factor gives
        // the same product, bit for bit, and the two transcendental functions are left out (matte floors are most of the hits).
        const bool matte = M.specular[0] == 0.0 && M.specular[1] == 0.0 && M.specular[2] == 0.0 && M.shininess > 0.0 && M.shininess < 1e12 && hdn >= 0.0;
        const double power = matte ? 1.0 : exp(M.shininess * log(hdn));
        D3 sp = cmul(ld3(M.specular), I) * power;
        if (!EMIT || !P.shadows) {  // visibility (1,1,1), shading.h:80
            ad = ad + cmul(diff, mk(1.0, 1.0, 1.0));
            spec = spec + cmul(sp, mk(1.0, 1.0, 1.0));
        } else {
            // compute_light_visibility (shading.h:79-101) is evaluated once for the diffuse and once for the specular
            // term (shading.h:154,192), shadow_samples rays each; a ray that reaches the light adds weight/samples.
            const double inv_s = 1.0 / P.shadow_samples;
            const unsigned slot = bundle_first + (unsigned)li * bundle_lanes;
            if (slot >= E.bundle_cap) { *E.overflow = 1u; continue; }   // the host sizes batches so that this cannot happen
            Bundle b;
            const D3 org = Pt + N * P.eps_shadow;
            const D3 wdv = cmul(w, diff * ad_weight) * inv_s, wsv = cmul(w, sp) * inv_s;
            b.p[0] = Pt.x; b.p[1] = Pt.y; b.p[2] = Pt.z;
            b.org[0] = org.x; b.org[1] = org.y; b.org[2] = org.z;
            b.wd[0] = wdv.x; b.wd[1] = wdv.y; b.wd[2] = wdv.z;
            b.ws[0] = wsv.x; b.ws[1] = wsv.y; b.ws[2] = wsv.z;
            b.time = q.time; b.rng = rng.next(); b.pixel = q.pixel; b.light = li; b.depth = q.depth; b.pad = 0;
            bundle_store(E.bundles, slot, b);
        }
    }
    contrib = cmul(w, ad * ad_weight + spec);
    return true;
}

// One lane per traced ray.  Contributions go to the FP64 accumulators with atomics, which on this chip execute at the
// memory side at a fraction of the store rate (MI355X_MICROARCH.md, global float atomics), so lanes that follow each
// other with the same pixel -- the samples of one pixel and their children stay neighbours in the queues -- are summed
// inside the wave first and only the first lane of each run issues the three atomics.
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 3          // waves per SIMD asked of the register allocator for the shading kernel that queues rays, analytic
#endif                            // shapes: 200 registers without a bound (two waves); held to 168 it spills a little and C3 / C1 gain 1-2 %
#ifndef RT_SHADE_TEX_WAVES
#define RT_SHADE_TEX_WAVES 2      // ... of its variants with textures, the ray marcher or the environment: 256 / 272 registers without
#endif                            // a bound -- the full variant then runs ONE wave per SIMD (C4 3.53 ms; held to 256: 3.21 ms); 3 spills badly (C5 + 8 %)
#ifndef RT_SHADE_LEAN_WAVES
#define RT_SHADE_LEAN_WAVES 4     // ... for the variant that cannot queue a ray (no emission code: 136 registers without a bound)
#endif
// (Tracing the rays of a small scene inside this kernel instead of in k_traverse_small -- no hit records, one launch per
// level less -- was built and measured: the kernel grows from 223 to 255 registers and C3 at 16 spp takes 47.6 instead of
// 40.3 ms, C1 0.67 instead of 0.66 ms: the traversal loop runs better at its own 4 waves per SIMD than inside this kernel's 2.)
// The work of one workgroup of k_shade on its span of a queue (a function of its own so that k_tail can run it level after level).
template <int F, bool EMIT>
__device__ __forceinline__ void shade_body(const DScene& S, const DParams& P, const RayQ& rays_all, const HitQ& hits_all, const QIn& in,
                                           const BlockSpan& span, const EmitCtx& E, double* __restrict__ accum) {
    const int lane = threadIdx.x & 63;
    const RayQ rays = rebase(rays_all, span.origin);
    const HitQ hits{hits_all.p + span.origin, hits_all.stride};
    // The host sizes the grid by the count when it knows it, and by the machine when the count is on the device (a level
    // enqueued without a round trip: its bound is often orders of magnitude above what is really there, and a hundred
    // thousand workgroups that only find out that they have nothing to do cost more than a small level's work).
    for (unsigned base = span.begin; base < span.end; base += span.step) {
        const unsigned i = base + threadIdx.x;
        // children and bundles of this wave go to the shard its rays came from (a dense queue deals its waves round)
        const EmitCtx Ew = EMIT ? emit_view(E, in.counts ? span.shard : (base + threadIdx.x) / 64u) : E;
        D3 c = mk(0.0, 0.0, 0.0);
        int pixel = -1 - lane;   // lanes without a contribution: a run of their own
        if (i < span.end) {
            const RayRec q = ray_load(rays, i, P);
            const HitRec hr = hit_load(hits, i, hits_carry_uv(S));
            D3 contrib = mk(0.0, 0.0, 0.0);
            if (shade_ray<F, EMIT>(S, P, q, hr, Ew, contrib)) {
                c = contrib;
                pixel = q.pixel;
            }
        }
        if (E.direct) {
            if (pixel >= 0) deposit(E, P, accum, pixel, c);
            continue;
        }
        // runs of equal pixel: [first lane whose left neighbour differs, lane before the next such lane]
        const int left = __shfl_up(pixel, 1);
        const unsigned long long heads = __ballot(lane == 0 || left != pixel);
        const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
        const int run_end = above ? lane + __ffsll((long long)above) - 1 : 63;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double ox = __shfl_down(c.x, d), oy = __shfl_down(c.y, d), oz = __shfl_down(c.z, d);
            if (lane + d <= run_end) { c.x += ox; c.y += oy; c.z += oz; }
        }
        if (pixel >= 0 && ((heads >> lane) & 1ull)) accum_add(accum, P, pixel, c);
    }
}

template <int F, bool EMIT, bool SMALL>
__global__ void __launch_bounds__(kBlock, (F & (RT_FEAT_COMPLEX | RT_FEAT_TEXTURE | RT_FEAT_ENV)) ? RT_SHADE_TEX_WAVES : EMIT ? RT_SHADE_WAVES : RT_SHADE_LEAN_WAVES) k_shade(const DScene S0, const DParams P, const RayQ rays_all,
                                                  const HitQ hits_all, const QIn in, const EmitCtx E, double* __restrict__ accum) {
    __shared__ SceneStage<SMALL> staged;
    const BlockSpan span = block_span(in);
    if (span.begin >= span.end) return;
    const DScene S = staged.stage(S0);
    shade_body<F, EMIT>(S, P, rays_all, hits_all, in, span, E, accum);
}

// ------------------------------------------------------------------------------------------ k_shadow
// One lane per shadow ray: lane i traces ray i % (2 x samples) of bundle i / (2 x samples) -- the first `samples` rays
// belong to the diffuse evaluation of the light's visibility, the others to the specular one (shading.h:154,192) -- from
// the light sample (shading.h:31-36) through every transmitting occluder (shading.h:42-75) to the light, inside this
// kernel: no queue entry, no hit record, no second kernel.  The rays of a bundle are neighbouring lanes with the same
// pixel, bundles of one hit and of neighbouring samples of a pixel follow each other, so the wave sums runs of equal
// pixel before the FP64 atomics (as k_shade does).  A chain longer than kInlineSegments segments goes on as an
// ordinary shadow ray of the next level's queue (shade_ray continues it there).
constexpr int kInlineSegments = 6;
#ifndef RT_SHADOW_WAVES
#define RT_SHADOW_WAVES 4         // waves per SIMD asked of the register allocator: 3 measured 5-8 % faster than 2 on the analytic
#endif                            // variants and 4 another 2-3.5 % on the 64-spp frames (5: spills, x 1.2 - 1.4 slower); the variant with
                                  // the ray marcher and the texture paths spills already at 3 (x 1.6) and stays at 2
#ifndef RT_SHADOW_COMPLEX_WAVES
#define RT_SHADOW_COMPLEX_WAVES 2 // ... of the variant with the ray marcher and the environment (3 and 4 spill: C4 7.5 -> 8.9 / 8.0 ms)
#endif
#ifndef RT_SHADOW_TEX_WAVES
#define RT_SHADOW_TEX_WAVES 2     // ... of the analytic + textures variant
#endif
// The work of one workgroup of k_shadow on its share of the bundles (stack_lds: this lane's column of the traversal stack in LDS).
template <int F, bool SMALL>
__device__ __forceinline__ void shadow_body(const DScene& S, const DParams& P, const BundleQ& bundles_all, const QIn& in, const EmitCtx& E0,
                                            double* __restrict__ accum, unsigned long long* __restrict__ stats, unsigned long long* stack_lds) {
    // The 2 x samples rays of a bundle towards a light of radius 0 are one and the same ray (shading.h:31-36 only moves the
    // target when the light has a size): it is traced ONCE, carries the sum of the weights and counts as the 2 x samples
    // closest-hit queries the reference makes.  With point lights only, a bundle is one lane; otherwise the other lanes of such
    // a bundle sit out.
    const unsigned per_ref = 2u * (unsigned)P.shadow_samples;
    bool point_lights_only = true;
    for (int l = 0; l < S.n_lights; ++l) point_lights_only = point_lights_only && S.lights[l].radius == 0.0;
    const bool merge = P.trace_coincident == 0;
    point_lights_only = point_lights_only && merge;
    const unsigned per = point_lights_only ? 1u : per_ref;
    const int per_shift = (per & (per - 1u)) == 0u ? __builtin_ctz(per) : -1;
    // this workgroup's shard of the bundles (block_span in 64-bit lanes: a bundle is `per` lanes)
    const unsigned shard = blockIdx.x % in.shards;
    const unsigned long long n = (unsigned long long)min(tail_load(&in.counts[shard * kCountStride]), in.region) * per;
    const unsigned long long first = (unsigned long long)(blockIdx.x / in.shards) * kBlock, stride = (unsigned long long)(gridDim.x / in.shards) * kBlock;
    const BundleQ bundles{bundles_all.p + (size_t)shard * in.region, bundles_all.stride};
    const EmitCtx E = emit_view(E0, shard);   // a chain that goes on as a queued ray stays in the shard
    const int lane = threadIdx.x & 63;
    TravStack stack;
    stack.lds = stack_lds;
    stack.stride = kBlock;
    Counters C{0u, 0u, 0u, 0u};
    unsigned coincident = 0u;   // queries of the reference that were not traced because they coincide with one that was
    const bool any_hit = S.transparent_materials == 0;
    // the grid is sized by the machine, not by the bound on the bundles (most levels of a ray tree hold far fewer than
    // the bound): every workgroup strides over the rays that are really there
    for (unsigned long long base = first; base < n; base += stride) {
        const unsigned long long i = base + threadIdx.x;
        D3 c = mk(0.0, 0.0, 0.0);
        int pixel = -1 - lane;
        bool mine = i < n;
        Bundle b;
        Ray sr{mk(0, 0, 0), mk(0, 0, 1), 0.0};
        D3 wt = mk(0, 0, 0);
        double dist = 0.0;
        Rng rng{0ull, true};
        unsigned copies = 1u;   // rays of the reference this lane's ray stands for
        if (mine) {
            // (lane -> bundle and sample: `per` is 2 x shadow_samples, a power of two with the reference's defaults -- a 64-bit
            // division by a run-time number is some ninety instructions, a sixth of this kernel's per ray when it was written so)
            unsigned bi, k;
            if (per_shift >= 0) { bi = (unsigned)(i >> per_shift); k = (unsigned)i & (per - 1u); }
            else if (n <= 0xffffffffull) { bi = (unsigned)i / per; k = (unsigned)i - bi * per; }
            else { bi = (unsigned)(i / per); k = (unsigned)(i % per); }
            b = bundle_load(bundles, bi);
            const DLight& L = S.lights[b.light];
            const bool specular = k >= (unsigned)P.shadow_samples;
            if (merge && L.radius == 0.0) {
                copies = per_ref;
                if (k != 0u) mine = false;
            }
            // this ray's own random stream (the reference draws the samples of a hit one after the other from one
            // generator; the estimator only needs them independent)
            rng = Rng{b.rng ^ ((unsigned long long)(k + 1u) * 0xD1B54A32D192ED03ull), P.rng_mode == RT_RNG_D0};
            (void)rng.next();
            const D3 Pt = ld3(b.p);
            D3 lp = ld3(L.pos);
            if (L.radius != 0.0) lp = lp + rng.on_unit_sphere() * L.radius;  // shading.h:31-36
            D3 d = lp - Pt;
            dist = length(d);
            d = normalize(d);
            sr = Ray{ld3(b.org), d, b.time};
            wt = copies > 1u ? (ld3(b.wd) + ld3(b.ws)) * (double)P.shadow_samples : specular ? ld3(b.ws) : ld3(b.wd);
        }
        bool going = mine, reached = false;
        int seg = 0;
        const unsigned rays_before = C.rays;
        for (; seg < kInlineSegments; ++seg) {
            if (SMALL) { if (__ballot(going) == 0ull) break; }
            else if (!going) break;
            Cand cand;
            const bool hit = SMALL ? closest_hit_small<F>(S, P, sr, 0.001, dist - 0.001, cand, C, any_hit, going)
                                   : closest_hit<F>(S, P, sr, 0.001, dist - 0.001, cand, C, stack, any_hit);
            if (!going) continue;
            if (!hit) { reached = true; going = false; continue; }
            const DPrim pr = S.prims[cand.prim];
            const DMaterial& M = S.materials[pr.material];
            if (!(M.transparency > 0.0)) { going = false; continue; }  // opaque occluder: transmission 0
            Hit h;
            fill_hit<F>(S, P, sr, cand, h);
            double n1 = h.front ? 1.0 : M.ior, n2 = h.front ? M.ior : 1.0;
            double eta = n1 / n2;
            double cos_i = -dot(sr.d, h.n);
            double sin2 = eta * eta * (1.0 - cos_i * cos_i);
            if (sin2 > 1.0) { going = false; continue; }
            double r0 = ((n1 - n2) / (n1 + n2));
            r0 = r0 * r0;
            double refl = r0 + (1.0 - r0) * pow((1.0 - cos_i), 5.0);  // shadow_schlick, shading.h:24-28
            D3 local = cmul(mk(1.0, 1.0, 1.0), ld3(M.diffuse)) * (1.0 - refl);
            if (length(local) < 0.001) { going = false; continue; }
            wt = cmul(wt, local);
            sr.o = h.p + sr.d * 0.001;
            dist = dist - h.t;
        }
        coincident += (C.rays - rays_before) * (copies - 1u);
        C.rays += (C.rays - rays_before) * (copies - 1u);
        if (reached) {
            c = wt;
            pixel = b.pixel;
        } else if (going) {   // still inside a stack of transmitting surfaces: the next level's queue takes over (a ray that stands
                              // for several goes on as that many, so that the queries of the queued part are counted as the reference's)
            const D3 share = copies > 1u ? wt * (1.0 / (double)copies) : wt;
            for (unsigned k = 0; k < copies; ++k)
                emit_ray(E.out, E.count, E.cap, E.overflow, sr.o, sr.d, share, b.time, dist, rng.next(), b.pixel, b.depth, kKindShadow);
        }
        const int left = __shfl_up(pixel, 1);
        const unsigned long long heads = __ballot(lane == 0 || left != pixel);
        const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
        const int run_end = above ? lane + __ffsll((long long)above) - 1 : 63;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const double ox = __shfl_down(c.x, dlt), oy = __shfl_down(c.y, dlt), oz = __shfl_down(c.z, dlt);
            if (lane + dlt <= run_end) { c.x += ox; c.y += oy; c.z += oz; }
        }
        if (pixel >= 0 && ((heads >> lane) & 1ull)) accum_add(accum, P, pixel, c);
    }
    // closest-hit queries of this wave (== the reference's world.intersect calls of its shadow rays)
    unsigned rays = C.rays, nodes = C.nodes, prims = C.prims, visits = C.visits;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        rays += __shfl_down(rays, off);
        coincident += __shfl_down(coincident, off);
        if (P.count_tests) {
            nodes += __shfl_down(nodes, off);
            prims += __shfl_down(prims, off);
            visits += __shfl_down(visits, off);
        }
    }
    if (lane == 0 && rays != 0u) {
        atomicAdd(&stats[7], (unsigned long long)rays);
        if (coincident != 0u) atomicAdd(&stats[3], (unsigned long long)coincident);   // (word 3 is free on the device: the traversal time is the host's)
        if (P.count_tests) {
            atomicAdd(&stats[1], (unsigned long long)nodes);
            atomicAdd(&stats[2], (unsigned long long)prims);
            atomicAdd(&stats[4], (unsigned long long)visits);
        }
    }
}

template <int F, bool SMALL>
__global__ void __launch_bounds__(kBlock, (F & (RT_FEAT_COMPLEX | RT_FEAT_ENV)) ? RT_SHADOW_COMPLEX_WAVES : (F & RT_FEAT_TEXTURE) ? RT_SHADOW_TEX_WAVES : RT_SHADOW_WAVES)
k_shadow(const DScene S0, const DParams P, const BundleQ bundles_all, const QIn in, const EmitCtx E0, double* __restrict__ accum,
         unsigned long long* __restrict__ stats) {
    __shared__ unsigned long long stack_mem[SMALL ? 1 : kTravLdsEntries + 1][kBlock];
    __shared__ SceneStage<SMALL> staged;
    const DScene S = staged.stage(S0);
    shadow_body<F, SMALL>(S, P, bundles_all, in, E0, accum, stats, &stack_mem[0][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------ k_traverse_small
// k_traverse for scenes of at most kSmallScenePrims primitives: one lane per queued ray, every primitive tested in step
// by the whole wave (closest_hit_small).  The persistent scheduling of k_traverse -- refill, votes, parked leaves, a
// stack -- buys nothing when there is no hierarchy to speak of, and costs most of the time of such frames.
// The work of one workgroup of k_traverse_small on its span of a queue.
template <int F>
__device__ __forceinline__ void traverse_small_body(const DScene& S, const DParams& P, const RayQ& rays_all, const HitQ& hits_all, const BlockSpan& span,
                                                    unsigned long long* __restrict__ counters) {
    const RayQ rays = rebase(rays_all, span.origin);
    const HitQ hits{hits_all.p + span.origin, hits_all.stride};
    const unsigned n = span.end;
    const int lane = threadIdx.x & 63;
    Counters C{0u, 0u, 0u, 0u};
    for (unsigned base = span.begin; base < n; base += span.step) {
        const unsigned i = base + threadIdx.x;
        const bool mine = i < n;
        Ray r{mk(0, 0, 0), mk(0, 0, 1), 0.0};
        double tmin = 0.0, tmax = 0.0;
        bool shadow = false;
        if (mine) {
            const RayRec q = ray_load_head(rays, i);
            r.o = ld3(q.o);
            r.d = ld3(q.d);
            r.time = q.time;
            shadow = q.kind == kKindShadow;
            tmin = shadow ? 0.001 : P.eps_scene;                 // shading.h:45 / tracer.h:152-156
            tmax = shadow ? q.tmax_shadow - 0.001 : 100000.0;
        }
        Cand best;
        // queued shadow rays are continuations through transmitting surfaces: they need their closest occluder
        (void)closest_hit_small<F>(S, P, r, tmin, tmax, best, C, false, mine);
        if (mine) {
            HitRec h;
            h.t = best.t; h.bu = best.bu; h.bv = best.bv; h.prim = best.prim; h.tri = best.tri;
            hit_store(hits, i, h, hits_carry_uv(S));
        }
    }
    if (P.count_tests && counters) {
        unsigned prims = C.prims;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) prims += __shfl_down(prims, off);
        if (lane == 0 && prims != 0u) atomicAdd(&counters[2], (unsigned long long)prims);   // (idle waves of a nearly empty level stay off the counter)
    }
}

#ifndef RT_SMALL_WAVES
#define RT_SMALL_WAVES 4
#endif
template <int F>
__global__ void __launch_bounds__(kBlock, RT_SMALL_WAVES)
k_traverse_small(const DScene S0, const DParams P, const RayQ rays_all, const HitQ hits_all, const QIn in,
                 unsigned long long* __restrict__ counters, unsigned long long* __restrict__ level_total, unsigned* __restrict__ level_size,
                 unsigned* __restrict__ zero_a, unsigned* __restrict__ zero_b, const unsigned zero_shards) {
    __shared__ SmallSceneLds small_lds;
    // the tails this level's shading kernel counts into (next queue's shards, shadow bundles) start at zero: stores here
    // instead of a fill launch each (the kernels of a stream run one after the other)
    if (blockIdx.x == 0 && threadIdx.x < zero_shards) {
        if (zero_a) zero_a[threadIdx.x * kCountStride] = 0u;
        if (zero_b) zero_b[threadIdx.x * kCountStride] = 0u;
    }
    // a level whose size only the device knows joins the frame's ray total, and its size the record of the frame's levels
    if (in.counts && blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned t = queue_total(in);
        if (t != 0u) atomicAdd(level_total, (unsigned long long)t);
        if (level_size) *level_size = t;
    }
    const BlockSpan span = block_span(in);
    if (span.begin >= span.end) return;   // (whole workgroup: before the barrier of the staging)
    const DScene S = stage_small_scene(S0, small_lds);
    traverse_small_body<F>(S, P, rays_all, hits_all, span, counters);
}

// ------------------------------------------------------------------------------------------ k_tail
// The deep levels of a small scene's ray tree in ONE launch.  A 1-spp frame of the default scene is a dozen levels of which all
// but the first two hold a few thousand rays at most: three launches each, ~30 us per level in launch gaps and in grids of two
// thousand workgroups that find nothing to do -- 330 of the C1 frame's 600 us.  The queues of a small scene are cut into K shards
// and a wave emits into the shard its input came from, so the ray tree below a shard's entries never leaves the shard: workgroup
// r of this kernel (grid = K) takes shard r through traversal, shading and shadow rays of level after level on its own, with
// workgroup barriers between the phases and no synchronisation across workgroups at all; it stops when its shard runs empty.
// The phases are the bodies of k_traverse_small, k_shade and k_shadow, the queues alternate between the two the host hands
// over, and what the per-level kernels leave for the host (tails of the last level, ray total, level sizes, overflow flag) is
// left the same way.  Whether and from which level a frame ends this way is decided by the host from the level sizes of the
// previous frame of the same shape (Driver::tail_from): a guess that can only cost time, never a ray.
struct TailArgs {
    RayQ q[2];                  // level `first` reads q[0] and queues into q[1], the next level the other way round, ...
    HitQ h[2];
    BundleQ bundles;
    unsigned* shard_counts;     // tails of level l: shard_counts + (l % kCountArrays) * kShards * kCountStride
    unsigned* bundle_tails;
    unsigned* overflow;
    unsigned long long* level_total;
    unsigned* level_sizes;      // [kLevelSizeWords]
    uint8_t* direct;
    unsigned region, bundle_region, shards;
    int first, levels, shadows;
};
template <int F>
__global__ void __launch_bounds__(kBlock, 2)
k_tail(const DScene S0, const DParams P, const TailArgs A, double* __restrict__ accum, unsigned long long* __restrict__ stats) {
    __shared__ SmallSceneLds lds;
    __shared__ unsigned long long stack_mem[1][kBlock];
    const DScene S = stage_small_scene(S0, lds);
    const unsigned shard = blockIdx.x;   // gridDim.x == A.shards
    unsigned* const my_bundle_tail = A.bundle_tails + shard * kCountStride;
    int level = A.first;
    for (; level < A.levels; ++level) {
        unsigned* const tails_in = A.shard_counts + (size_t)(level % (int)kCountArrays) * kShards * kCountStride;
        unsigned* const tails_out = A.shard_counts + (size_t)((level + 1) % (int)kCountArrays) * kShards * kCountStride;
        const unsigned n_in = min(tail_load(&tails_in[shard * kCountStride]), A.region);   // the same number in every lane
        if (threadIdx.x == 0) {
            tails_out[shard * kCountStride] = 0u;
            *my_bundle_tail = 0u;
            if (n_in != 0u) {
                atomicAdd(A.level_total, (unsigned long long)n_in);
                if (level < (int)kLevelSizeWords) atomicAdd(&A.level_sizes[level], n_in);
            }
        }
        if (n_in == 0u) break;
        __syncthreads();   // the zeroes stand before anything is counted
        const int k = (level - A.first) & 1;
        const QIn in{tails_in, 0u, A.shards, A.region, 0u, 0xffffffffu};
        const BlockSpan span = block_span(in);
        traverse_small_body<F>(S, P, A.q[k], A.h[k], span, stats);
        __syncthreads();   // hit records written, visible to the workgroup that wrote them
        const EmitCtx E{A.q[k ^ 1], tails_out, A.overflow, A.region, A.direct, A.bundles, A.bundle_tails, A.bundle_region, A.shards};
        shade_body<F, true>(S, P, A.q[k], A.h[k], in, span, E, accum);
        __syncthreads();   // children and bundles queued and counted
        if (A.shadows) {
            const QIn bin{A.bundle_tails, 0u, A.shards, A.bundle_region, 0u, 0xffffffffu};
            shadow_body<F, true>(S, P, A.bundles, bin, E, accum, stats, &stack_mem[0][threadIdx.x]);
            __syncthreads();
        }
    }
    // the host sums the tails of level `levels` and reads the level sizes: a shard that ran empty early leaves zeroes there
    if (threadIdx.x == 0 && level < A.levels)
        A.shard_counts[(size_t)(A.levels % (int)kCountArrays) * kShards * kCountStride + shard * kCountStride] = 0u;
}

// ------------------------------------------------------------------------------------------ k_resolve
// (pixel counts reach 2^31 - 1 -- rt_render_frames_device: 4096 frames of a rank's share -- so every index is a size_t)
__global__ void __launch_bounds__(kBlock) k_resolve(const double* __restrict__ accum, uint8_t* __restrict__ out, const size_t npix,
                                                    const int spp, const int mode) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= npix) return;
    D3 c = mk(accum[i], accum[npix + i], accum[2 * npix + i]);
    D3 avg = tonemap(c * (1.0 / spp), mode);  // main.cpp:308-316
    out[i * 3 + 0] = quantise(avg.x);
    out[i * 3 + 1] = quantise(avg.y);
    out[i * 3 + 2] = quantise(avg.z);
}

__global__ void k_add_counters(unsigned long long* c, unsigned long long rays, unsigned long long boxes, unsigned long long prims,
                               unsigned long long traverse_ns, unsigned long long fetches, unsigned long long max_fetches,
                               unsigned long long max_iters, const unsigned long long* __restrict__ shadow_rays,
                               const unsigned long long* __restrict__ coincident) {
    if (threadIdx.x == 0) {
        c[0] += rays + (shadow_rays ? *shadow_rays : 0ull);   // queue entries traversed (host count) + rays traced inside k_shadow
        c[7] += coincident ? *coincident : 0ull;              // of which: counted as the reference counts them, traced as one (k_shadow)
        c[1] += boxes; c[2] += prims; c[3] += traverse_ns; c[4] += fetches;
        if (max_fetches > c[5]) c[5] = max_fetches;
        if (max_iters > c[6]) c[6] = max_iters;
    }
}

// Entries in all shards of a queue -> one word the host reads back (with the overflow flag and the ray total next to it).
__global__ void k_sum_counts(const unsigned* __restrict__ counts, const unsigned shards, const unsigned region, unsigned* __restrict__ out) {
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (unsigned r = 0; r < shards; ++r) t += min(counts[r * kCountStride], region);
        *out = t;
    }
}

// DScene::xforms32 from DScene::xforms (round to nearest).
__global__ void __launch_bounds__(kBlock) k_make_xforms32(const DXform* __restrict__ x, const int n, float* __restrict__ out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;   // one float per thread
    if (i >= n * 16) return;
    const DXform& X = x[i >> 4];
    const int k = i & 15;
    out[i] = (float)(k < 12 ? X.inv[k] : k < 15 ? X.vel[k - 12] : X.inv33);
}

// Linear radiance (the average before tone mapping) for float output and progressive accumulation.
__global__ void __launch_bounds__(kBlock) k_resolve_linear(const double* __restrict__ accum, double* __restrict__ out, const size_t n3,
                                                           const int spp, const int add) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;   // interleaved RGB out, planes in
    if (i >= n3) return;
    const double v = accum[(i % 3) * (n3 / 3) + i / 3] * (1.0 / spp);
    out[i] = add ? out[i] + v : v;   // add: a pass of a progressive render joins the running sum in HBM
}

// Progressive render: the sum of `passes` pass averages -> the picture so far (average, tone map, truncate: main.cpp:308-320).
__global__ void __launch_bounds__(kBlock) k_resolve_sum(const double* __restrict__ sum, uint8_t* __restrict__ out, const size_t npix,
                                                        const double inv_passes, const int mode) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= npix) return;
    const D3 avg = tonemap(mk(sum[i * 3 + 0], sum[i * 3 + 1], sum[i * 3 + 2]) * inv_passes, mode);
    out[i * 3 + 0] = quantise(avg.x);
    out[i * 3 + 1] = quantise(avg.y);
    out[i * 3 + 2] = quantise(avg.z);
}

// ------------------------------------------------------------------------------------------ host driver
constexpr int kSpheres = RT_FEAT_SPHERE;
constexpr int kAnalytic = RT_FEAT_SPHERE | RT_FEAT_QUAD | RT_FEAT_CUBE;
constexpr int kFull = kAnalytic | RT_FEAT_COMPLEX | RT_FEAT_TEXTURE | RT_FEAT_ENV;
constexpr int kTextured = kAnalytic | RT_FEAT_TEXTURE;   // analytic shapes with textures / bump maps (k_shade; k_shadow where transmitting occluders may carry bump maps)

constexpr size_t kMaxTimedEvents = 8192;   // start/stop events kept per workspace until they are collected
constexpr size_t kMinQueueRays = size_t(1) << 16;
constexpr size_t kDefaultMaxQueueRays = size_t(1) << 26;   // 9.7 GB per queue (rays + hit records)
constexpr size_t kGrownMaxQueueRays = size_t(1) << 28;     // after a frame's ray tree outgrew the queues (one-chunk frames only)

struct Driver {
    const DScene& S;
    const DParams& P;
    WavefrontWorkspace& ws;
    hipStream_t stream;
    int variant;
    // Traversal only looks at geometry (textures are sampled by the ray marcher of the displaced shapes alone), and the shadow
    // kernel at geometry plus the bump maps of transmitting occluders: a textured scene without displaced shapes is traversed by
    // the analytic variants, which need half the registers of the full one.
    int trav_variant = 2, shadow_variant = 2;   // 0 spheres, 1 analytic, 2 full, 3 (shadow) analytic + textures
    int traverse_blocks;
    unsigned long long rays_traced = 0;
    bool may_emit = true;   // false: no launch of k_shade can queue a ray
    size_t max_capacity = 0;   // RT_QUEUE_RAYS
    uint8_t* direct = nullptr;   // see EmitCtx::direct
    double* accum = nullptr;     // the frame's accumulators (the workspace's own, or the partner's frame's: see WavefrontWorkspace::partner)
    hipError_t err = hipSuccess;

    bool ok(hipError_t e) {
        if (e != hipSuccess && err == hipSuccess) err = e;
        return err == hipSuccess;
    }
    Chunk chunk{0, 0, 0, 1};   // the chunk of primary rays being rendered
    // queue q as the kernels see it: level 0 holds the slim primary records of `chunk`, deeper levels full records
    RayQ rays(int q, int level = 1) { return RayQ{static_cast<rt_q4*>(ws.queues[q]), (unsigned)ws.capacity, level == 0 ? 1u : 0u, chunk}; }
    HitQ hits(int q) { return HitQ{static_cast<rt_q4*>(ws.hit_buffers[q]), (unsigned)ws.capacity}; }

    int fail_acquire_in = -1;   // tests (RT_TEST_PARTNER_FAIL=acquire): this many calls from now, acquire() reports the device full
    int acquire() {
        if (fail_acquire_in >= 0 && fail_acquire_in-- == 0) { ok(hipErrorOutOfMemory); return -1; }
        for (size_t i = 0; i < ws.queues.size(); ++i)
            if (!ws.in_use[i]) { ws.in_use[i] = 1; return (int)i; }
        if (ws.fixed) { ok(hipErrorOutOfMemory); return -1; }   // the ray tree needs more queues than the caller's memory holds
        void *q = nullptr, *h = nullptr;
        if (!ok(hipMalloc(&q, ws.capacity * sizeof(RayRec)))) return -1;
        if (!ok(hipMalloc(&h, ws.capacity * sizeof(HitRec)))) { (void)hipFree(q); return -1; }
        ws.queues.push_back(q);
        ws.hit_buffers.push_back(h);
        ws.in_use.push_back(1);
        return (int)ws.queues.size() - 1;
    }
    void release(int q) { ws.in_use[q] = 0; }

    // worst-case number of rays one queued ray can emit at `level` (radiance rays there have depth max_depth - level)
    size_t fanout(int level) const {
        int depth = P.max_depth - level;
        size_t refl = (depth < P.max_bounces) ? 1 : (size_t)std::max(P.glossy_samples, 1);
        size_t shadow = P.shadows ? 2ull * (size_t)S.n_lights * (size_t)P.shadow_samples : 0ull;
        return std::max<size_t>(1, refl + 1 + shadow);
    }

    hipEvent_t next_event() {
        if (ws.events_used >= kMaxTimedEvents) return nullptr;
        if (ws.events_used == ws.events.size()) {
            hipEvent_t e = nullptr;
            if (!ok(hipEventCreate(&e))) return nullptr;
            ws.events.push_back(e);
        }
        return ws.events[ws.events_used++];
    }
    // ---- sharded queues (see QIn): level 0 is the dense queue k_raygen wrote; the queue of level l >= 1 has K shards whose
    //      tails are the array tails(l)
    unsigned K = 1;
    unsigned region() const { return (unsigned)(ws.capacity / K); }
    unsigned* tails(int level) { return ws.shard_counts + (size_t)(level % (int)kCountArrays) * kShards * kCountStride; }
    unsigned* bundle_tails() { return ws.shard_counts + (size_t)kCountArrays * kShards * kCountStride; }
    size_t tails_bytes() const { return K == 1 ? sizeof(unsigned) : (size_t)kShards * kCountStride * sizeof(unsigned); }
    QIn dense(unsigned n, unsigned lo = 0, unsigned hi = 0xffffffffu) { return QIn{nullptr, n, 1u, (unsigned)ws.capacity, lo, std::min(hi, n)}; }
    QIn sharded(int level, unsigned lo = 0, unsigned hi = 0xffffffffu) { return QIn{tails(level), 0u, K, region(), lo, hi}; }
    // grid of a kernel that strides over `in`: by the count where the host knows it, else by the machine; a multiple of the shards
    unsigned grid_for(const QIn& in, unsigned long long bound_lanes, int per_cu = 0) {
        unsigned long long blocks = (bound_lanes + kBlock - 1) / kBlock;
        if (!in.counts) return (unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>(blocks, 0x7fffffffull));
        blocks = std::min<unsigned long long>(blocks, (unsigned long long)ws.compute_units * (unsigned)(per_cu > 0 ? per_cu : grid_blocks_per_cu));
        const unsigned k = in.shards;
        return (unsigned)std::max<unsigned long long>(k, (blocks + k - 1) / k * k);
    }

    template <int F> void launch_traverse(int q, const QIn& in, unsigned bound, int level) {
        const bool timed = !ws.fixed && ws.events_used + 2 <= kMaxTimedEvents;
        hipEvent_t e0 = timed ? next_event() : nullptr, e1 = e0 ? next_event() : nullptr;
        if (e1) (void)hipEventRecord(e0, stream);
        launch_traverse_kernel<F>(q, in, bound, level);
        if (e1) (void)hipEventRecord(e1, stream);
        else if (e0) --ws.events_used;   // keep start/stop pairs whole
    }
    template <int F> void launch_traverse_kernel(int q, const QIn& in, unsigned bound, int level) {
        unsigned long long* level_total = reinterpret_cast<unsigned long long*>(ws.counters + kLevelTotal);
        if (small_scene()) {
            const unsigned blocks = in.counts ? grid_for(in, bound) : std::max(1u, std::min((bound + kBlock - 1) / kBlock, (unsigned)ws.compute_units * (unsigned)grid_blocks_per_cu));
            hipLaunchKernelGGL((k_traverse_small<F>), dim3(blocks), dim3(kBlock), 0, stream, S, P, rays(q, level), hits(q), in, ws.stats, level_total,
                               level < (int)kLevelSizeWords ? ws.counters + kLevelSizeBase + level : nullptr, zero_a, zero_b, K);
            zero_a = zero_b = nullptr;
            return;
        }
        // (scenes with a hierarchy have unsharded queues: K == 1, and the tail of a level is word 0 of its array)
        const unsigned n = bound;
        const unsigned* n_device = in.counts;
        // Persistent grid sized by the work: at least RT_RAYS_PER_LANE rays per lane, so that lanes can be refilled and
        // a small queue (a part of a frame, a deep level of the ray tree) leaves the other CUs to concurrent streams.
        const unsigned per_block = kBlock * (unsigned)std::max(1, ws.rays_per_lane);
        const unsigned want = (n + per_block - 1) / per_block;
        const unsigned blocks = std::max(1u, std::min((unsigned)traverse_blocks, want));
        const unsigned *order = nullptr;
        unsigned* cost = nullptr;
        const bool history = chunk_history(n, n_device != nullptr, level, &order, &cost);
        if (P.count_tests)
            hipLaunchKernelGGL((k_traverse<F, true>), dim3(blocks), dim3(kBlock), 0, stream, S, P, rays(q, level), hits(q), n, n_device, per_block,
                               ws.counters + kNextBase, ws.stats, level_total, order, cost);
        else
            hipLaunchKernelGGL((k_traverse<F, false>), dim3(blocks), dim3(kBlock), 0, stream, S, P, rays(q, level), hits(q), n, n_device, per_block,
                               ws.counters + kNextBase, ws.stats, level_total, order, cost);
        if (history) {   // this launch's record -> the order of the next launch over such a queue
            hipLaunchKernelGGL(k_chunk_order, dim3(kXcds), dim3(kBlock), 0, stream, ws.chunk_cost, ws.chunk_order, n);
            ws.chunk_order_valid = true;
        }
    }
    // The chunk history of this stream (see k_chunk_order): kept for primary-ray queues of whole frames -- the level-0 queue of a
    // launch that is one chunk -- of one scene, size and frame layout at a time.  *order: the order recorded by the previous such
    // launch, or null (none yet, another queue since, RT_NO_CHUNK_ORDER); *cost: where this launch records.
    bool chunk_history(unsigned n, bool size_on_device, int level, const unsigned** order, unsigned** cost) {
        static const bool off = getenv("RT_NO_CHUNK_ORDER") != nullptr;
        // (launches of many frames have their tail amortised already, and sorting half a million chunks costs them 125 us)
        if (off || ws.fixed || level != 0 || size_on_device || n < 65536u || n > (1u << 22) || n % (unsigned)RT_CHUNK != 0u) return false;
        if ((long long)chunk.rows * P.width * chunk.ns != (long long)n || chunk.rows != P.frames * P.part_rows || chunk.ns != P.spp) return false;   // not the whole launch
        const unsigned chunks = n / (unsigned)RT_CHUNK;
        const unsigned long long key[4] = {S.serial, ((unsigned long long)n << 32) | (unsigned)P.frames, ((unsigned long long)P.part_index << 32) | (unsigned)P.part_count,
                                           ((unsigned long long)P.width << 32) | (unsigned)P.part_rows};
        if (ws.chunk_capacity < chunks) {
            if (ws.chunk_cost) { if (!ok(hipStreamSynchronize(stream))) return false; (void)hipFree(ws.chunk_cost); (void)hipFree(ws.chunk_order); }
            ws.chunk_cost = ws.chunk_order = nullptr;
            ws.chunk_capacity = 0;
            if (hipMalloc(reinterpret_cast<void**>(&ws.chunk_cost), chunks * sizeof(unsigned)) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&ws.chunk_order), chunks * sizeof(unsigned)) != hipSuccess) {
                (void)hipGetLastError();
                if (ws.chunk_cost) (void)hipFree(ws.chunk_cost);
                ws.chunk_cost = ws.chunk_order = nullptr;
                return false;   // no history: the launch runs in queue order
            }
            ws.chunk_capacity = chunks;
            ws.chunk_order_valid = false;
        }
        if (!ws.chunk_order_valid || memcmp(key, ws.chunk_key, sizeof(key)) != 0) {
            if (!ok(hipMemsetAsync(ws.chunk_cost, 0, chunks * sizeof(unsigned), stream))) return false;
            memcpy(ws.chunk_key, key, sizeof(key));
            ws.chunk_order_valid = false;
        }
        *order = ws.chunk_order_valid ? ws.chunk_order : nullptr;
        *cost = ws.chunk_cost;
        return true;
    }
    unsigned *zero_a = nullptr, *zero_b = nullptr;   // tails the next small-scene traversal launch zeroes (run_levels)
    bool allow_grow = true;   // a frame of one chunk whose ray tree outgrows the queues gets larger queues next time (a frame of
                              // several chunks sizes its chunks by the queues: larger queues, larger chunks, the same overflow)
    bool with_shadows() const { return may_emit && P.shadows && S.n_lights > 0; }
    // (the environment is looked at once per frame, not once per launch: a 1-spp frame of a small scene is some forty launches)
    const bool small_env_ok = getenv("RT_NO_SMALL_SCENE") == nullptr;
    // workgroups per CU of the grids that stride over a queue whose size only the device knows: 8 (16: C1 + shadows 10 % slower,
    // the 64-spp frames 1-2 %; 4: the same or worse but for C1 + shadows; 2: 20-30 % slower.  RT_GRID_BLOCKS_PER_CU: A/B runs)
    const int grid_blocks_per_cu = getenv("RT_GRID_BLOCKS_PER_CU") ? std::max(1, atoi(getenv("RT_GRID_BLOCKS_PER_CU"))) : 8;
    bool small_scene() const { return S.n_prims <= kSmallScenePrims && small_env_ok; }
    // (A/B runs) workgroups per CU of the grids of the shading kernel that queues rays -- 2 are resident at its 220 registers: 1-2 %
    // faster than 8 on every config -- and of k_shadow (default: what is resident, see launch_shadow)
    const int shade_grid_per_cu = getenv("RT_SHADE_GRID_PER_CU") ? atoi(getenv("RT_SHADE_GRID_PER_CU")) : 0;   // 0: what is resident
    const int shadow_grid_per_cu = getenv("RT_SHADOW_GRID_PER_CU") ? atoi(getenv("RT_SHADOW_GRID_PER_CU")) : 0;
    // children of a launch go to queue qo (the queue of level `out_level`), its bundles to the bundle buffer
    EmitCtx emit_ctx(int qo, int out_level) {
        return EmitCtx{qo >= 0 ? rays(qo) : RayQ{nullptr, 0u, 0u, chunk}, tails(out_level), ws.counters + 2, qo >= 0 ? region() : 0u, direct,
                       BundleQ{static_cast<rt_q4*>(ws.bundles), (unsigned)ws.capacity}, bundle_tails(), ws.bundles ? region() : 0u, K};
    }
    // The shadow rays of the hits a shading launch of at most n rays has just bundled (at most n x lights bundles).
    template <int F> void launch_shadow(unsigned n, int qo, int out_level) {
        const EmitCtx E = emit_ctx(qo, out_level);
        const QIn in{bundle_tails(), 0u, K, region(), 0u, 0xffffffffu};
        const unsigned long long bound = std::min<unsigned long long>((unsigned long long)n * (unsigned)S.n_lights, ws.capacity) * 2ull * (unsigned)P.shadow_samples;
        // The variants with the ray marcher or the texture paths hold two workgroups per CU (a workgroup is one wave per SIMD): a grid of
        // exactly those -- more only queue up behind them and stage the scene again (C4 3.91 -> 3.55 ms).  The analytic variant holds four and keeps the 8 per CU of the other striding grids: 4 is 3 % faster on C1 + shadows
        // and 2 % slower on C3 and on the 3900-quad scene.
        constexpr int kResident = (F & (RT_FEAT_COMPLEX | RT_FEAT_ENV)) ? RT_SHADOW_COMPLEX_WAVES : (F & RT_FEAT_TEXTURE) ? RT_SHADOW_TEX_WAVES : 0;
        const unsigned blocks = grid_for(in, bound, shadow_grid_per_cu > 0 ? shadow_grid_per_cu : kResident);
        if (small_scene())
            hipLaunchKernelGGL((k_shadow<F, true>), dim3(blocks), dim3(kBlock), 0, stream, S, P, E.bundles, in, E, accum, ws.stats);
        else
            hipLaunchKernelGGL((k_shadow<F, false>), dim3(blocks), dim3(kBlock), 0, stream, S, P, E.bundles, in, E, accum, ws.stats);
    }
    void shadow(unsigned n, int qo, int out_level) {
        if (!with_shadows()) return;
        if (shadow_variant == 0) launch_shadow<kSpheres>(n, qo, out_level);
        else if (shadow_variant == 1) launch_shadow<kAnalytic>(n, qo, out_level);
        else if (shadow_variant == 3) launch_shadow<kTextured>(n, qo, out_level);
        else launch_shadow<kFull>(n, qo, out_level);
    }
    template <int F> void launch_shade(int q, const QIn& in, unsigned bound, int qo, int out_level) {
        const int level = out_level - 1;
        const EmitCtx E = emit_ctx(qo, out_level);
        constexpr int kResident = (F & (RT_FEAT_COMPLEX | RT_FEAT_TEXTURE | RT_FEAT_ENV)) ? RT_SHADE_TEX_WAVES : RT_SHADE_WAVES;
        const dim3 grid(grid_for(in, in.counts ? bound : in.hi - std::min(in.lo, in.hi), may_emit ? (shade_grid_per_cu > 0 ? shade_grid_per_cu : kResident) : 0));
        const QIn& v = in;
        if (may_emit && small_scene())
            hipLaunchKernelGGL((k_shade<F, true, true>), grid, dim3(kBlock), 0, stream, S, P, rays(q, level), hits(q), v, E, accum);
        else if (may_emit)
            hipLaunchKernelGGL((k_shade<F, true, false>), grid, dim3(kBlock), 0, stream, S, P, rays(q, level), hits(q), v, E, accum);
        else if (small_scene())
            hipLaunchKernelGGL((k_shade<F, false, true>), grid, dim3(kBlock), 0, stream, S, P, rays(q, level), hits(q), v, E, accum);
        else
            hipLaunchKernelGGL((k_shade<F, false, false>), grid, dim3(kBlock), 0, stream, S, P, rays(q, level), hits(q), v, E, accum);
    }
    void traverse(int q, const QIn& in, unsigned bound, int level) {
        if (trav_variant == 0 && P.fp32_tests && S.xforms32 && !small_scene()) launch_traverse<kSpheres | RT_FEAT_FP32>(q, in, bound, level);
        else if (trav_variant == 0) launch_traverse<kSpheres>(q, in, bound, level);
        else if (trav_variant == 1) launch_traverse<kAnalytic>(q, in, bound, level);
        else launch_traverse<kFull>(q, in, bound, level);
    }
    void shade(int q, const QIn& in, unsigned bound, int qo, int out_level) {
        if (variant == 0) launch_shade<kSpheres>(q, in, bound, qo, out_level);
        else if (variant == 1) launch_shade<kAnalytic>(q, in, bound, qo, out_level);
        else if (variant == 3) launch_shade<kTextured>(q, in, bound, qo, out_level);
        else launch_shade<kFull>(q, in, bound, qo, out_level);
    }
    // entries in all shards of the queue of `level` -> the word next to the overflow flag (read back together)
    void sum_tails(int level) {
        hipLaunchKernelGGL(k_sum_counts, dim3(1), dim3(64), 0, stream, tails(level), K, region(), ws.counters + kLevelCountBase);
    }

    // k_tail: from which level on (0: not at all) the rest of this chunk's ray tree goes into one launch.  By the sizes the
    // levels of the previous frame of this shape had: the first level from which every level would keep the K workgroups of
    // k_tail busy for two rounds at most (8192 lanes a round; a shadow ray is a lane like a queued ray).  RT_NO_TAIL: A/B runs.
    int tail_from(int levels, const unsigned long long key[4]) {
        static const bool off = getenv("RT_NO_TAIL") != nullptr;
        if (off || ws.fixed || !small_scene() || K <= 1 || !may_emit || !allow_grow || levels > (int)kLevelSizeWords || levels < 3) return 0;
        if (!ws.level_sizes_valid || memcmp(key, ws.level_key, sizeof(ws.level_key)) != 0) return 0;
        const unsigned long long per = (S.area_lights || P.trace_coincident) ? 2ull * (unsigned)P.shadow_samples : 1ull;   // lanes of a bundle in k_shadow
        const unsigned long long lanes_per_ray = 1ull + (with_shadows() ? (unsigned long long)S.n_lights * per : 0ull);
        int from = levels;
        for (int l = levels - 1; l >= 1; --l) {
            if ((unsigned long long)ws.level_sizes[l] * lanes_per_ray > 2ull * K * kBlock) break;
            from = l;
        }
        return levels - from >= 2 ? from : 0;   // (one level is as well served by its own three launches)
    }
    template <int F> void launch_tail_variant(const TailArgs& A) {
        hipLaunchKernelGGL((k_tail<F>), dim3(K), dim3(kBlock), 0, stream, S, P, A, accum, ws.stats);
    }
    void launch_tail(int q, int qo, int first, int levels) {
        const TailArgs A{{rays(q), rays(qo)}, {hits(q), hits(qo)}, BundleQ{static_cast<rt_q4*>(ws.bundles), (unsigned)ws.capacity}, ws.shard_counts, bundle_tails(),
                         ws.counters + 2, reinterpret_cast<unsigned long long*>(ws.counters + kLevelTotal), ws.counters + kLevelSizeBase, direct,
                         region(), ws.bundles ? region() : 0u, K, first, levels, with_shadows() ? 1 : 0};
        // one variant for the three phases: the shading kernel's, which covers what the traversal and shadow variants cover
        if (variant == 0) launch_tail_variant<kSpheres>(A);
        else if (variant == 1) launch_tail_variant<kAnalytic>(A);
        else if (variant == 3) launch_tail_variant<kTextured>(A);
        else launch_tail_variant<kFull>(A);
    }
    // A whole ray tree without a host round trip per level: every level's queue size stays on the device (the kernels
    // of level L + 1 read the tails the shading kernel of level L left), grids are sized by what the host knows -- the
    // level cannot hold more than the previous level's bound times its fan-out, nor more than a queue -- and the queue
    // overflow flag, the size of the last queue and the ray total are read back ONCE at the end.  Optimistic: a level
    // that outgrows its queue (the kernels never write past a shard) makes the caller render the frame again level by
    // level with batches (process()).  Returns 0 = done, 1 = overflow; rays still queued after `levels` levels (chains
    // of shadow segments through transmitting surfaces) are finished by process().
    int run_levels(int q, unsigned n0, int levels) {
        // overflow flag, the word the last queue's size goes to, the ray total and the level sizes start at zero (words 2 .. 31)
        if (!ok(hipMemsetAsync(ws.counters + 2, 0, (kLevelSizeBase + kLevelSizeWords - 2) * sizeof(unsigned), stream))) { release(q); return 0; }
        size_t bound = n0;
        int level = 0;
        const unsigned long long key[4] = {S.serial, ((unsigned long long)n0 << 32) | (unsigned)levels,
                                           ((unsigned long long)P.width << 32) | (unsigned)(P.frames * P.part_rows),
                                           (unsigned long long)P.shadows | (unsigned long long)P.fresnel << 1 | (unsigned long long)P.normals << 2 | (unsigned long long)P.use_bvh << 3 |
                                               (unsigned long long)P.rng_mode << 4 | (unsigned long long)P.glossy_samples << 8 | (unsigned long long)P.shadow_samples << 16 |
                                               (unsigned long long)P.max_bounces << 24 | (unsigned long long)P.spp << 32};
        const int tail = tail_from(levels, key);
        for (; level < (tail > 0 ? tail : levels) && err == hipSuccess; ++level) {
            const QIn in = level == 0 ? dense(n0) : sharded(level);
            const unsigned n = (unsigned)bound;
            const bool small = small_scene();
            if (small) { zero_a = tails(level + 1); zero_b = bundle_tails(); }   // k_traverse_small zeroes them itself
            else if (level > 0 && !ok(hipMemsetAsync(ws.counters + kNextBase, 0, kFetchCounterWords * sizeof(unsigned), stream))) break;
            traverse(q, in, n, level);
            const int qo = acquire();
            if (qo < 0) break;
            if (!small && !ok(hipMemsetAsync(tails(level + 1), 0, tails_bytes(), stream))) { release(qo); break; }
            if (!small && with_shadows() && !ok(hipMemsetAsync(bundle_tails(), 0, tails_bytes(), stream))) { release(qo); break; }
            shade(q, in, n, qo, level + 1);
            shadow(n, qo, level + 1);
            release(q);   // stream order: whatever reuses these buffers runs after this level's kernels
            q = qo;
            bound = std::min<size_t>(ws.capacity, bound * fanout(level));
        }
        if (tail > 0 && level == tail && err == hipSuccess) {   // the levels from `tail` on in one launch (k_tail)
            const int qo = acquire();
            if (qo >= 0) {
                launch_tail(q, qo, tail, levels);
                if (getenv("RT_DEBUG_LEVELS")) fprintf(stderr, "[rt levels] levels %d .. %d of %u primary rays in one launch (k_tail)\n", tail, levels - 1, n0);
                if ((levels - tail) & 1) { release(q); q = qo; } else release(qo);   // the queue the last level queued into
                level = levels;
            }
        }
        // (looking every third level whether anything is still queued was measured: a round trip costs ~70 us with the
        // launches behind it, more than the empty levels of a C1 frame save)
        constexpr unsigned kReadWords = kLevelSizeBase + kLevelSizeWords - 2;
        unsigned local[kReadWords] = {};
        unsigned* res = (ws.host_readback && ws.fixed) ? ws.host_readback : local;   // words 2 .. 31 of the counters, read back as one piece
        const unsigned read_words = ws.fixed ? 6u : kReadWords;                      // (the fixed workspace's pinned words hold the first six)
        unsigned queued = 0, overflowed = 0;
        unsigned long long total = 0;
        if (err == hipSuccess) {
            sum_tails(level);
            if (ok(hipMemcpyAsync(res, ws.counters + 2, read_words * sizeof(unsigned), hipMemcpyDeviceToHost, stream)) && ok(hipStreamSynchronize(stream))) {
                overflowed = res[0];
                queued = res[kLevelCountBase - 2];
                memcpy(&total, &res[kLevelTotal - 2], sizeof(total));
                if (!ws.fixed && !overflowed && small_scene() && K > 1 && allow_grow && levels <= (int)kLevelSizeWords) {   // the record for the next frame of this shape
                    memcpy(ws.level_sizes, &res[kLevelSizeBase - 2], sizeof(ws.level_sizes));
                    memcpy(ws.level_key, key, sizeof(key));
                    ws.level_sizes_valid = true;
                    if (getenv("RT_DEBUG_LEVELS")) {
                        fprintf(stderr, "[rt levels] sizes:");
                        for (int l = 1; l < levels; ++l) fprintf(stderr, " %u", ws.level_sizes[l]);
                        fprintf(stderr, "\n");
                    }
                }
            }
        }
        if (err != hipSuccess) { release(q); return 0; }
        if (overflowed) {
            // A level outgrew its queue.  At most twice per workspace the queues are doubled for the frames to come
            // (deep ray trees of glass scenes peak well above their first level) -- up to RT_QUEUE_RAYS when that is set,
            // else up to 2^28 rays, and only while the device has room for eight such queues; if that was not enough
            // either, this stream's frames stay on the level-by-level path.
            release(q);
            size_t free_bytes = 0, total_bytes = 0;
            if (!ws.fixed) (void)hipMemGetInfo(&free_bytes, &total_bytes);
            const size_t bigger = std::min(ws.capacity * 2, max_capacity);
            if (!ws.fixed && allow_grow && ws.grow <= 2 && bigger > ws.capacity && free_bytes >= 8 * bigger * (sizeof(RayRec) + sizeof(HitRec))) {
                ws.grow *= 2;
                ws.grown_capacity = bigger;
            } else {
                ws.levels_overflowed = true;
                ws.overflowed_scene = S.serial;
            }
            return 1;
        }
        rays_traced += n0 + total;   // level 0 was launched with a host count
        if (queued != 0u) process(q, queued, level);
        else release(q);
        return 0;
    }

    // Trace and shade the n rays of queue q (the queue of `level`; owned by this call, released when fully consumed), then
    // their children.  The host knows n; with sharded queues (K > 1, level > 0) the kernels still find the entries through
    // the tails on the device.
    void process(int q, unsigned n, int level) {
        if (err != hipSuccess || n == 0) { release(q); return; }
        const bool is_dense = level == 0 || K == 1;
        // level 0: k_raygen has zeroed the fetch counters
        if (level > 0 && !small_scene() && !ok(hipMemsetAsync(ws.counters + kNextBase, 0, kFetchCounterWords * sizeof(unsigned), stream))) { release(q); return; }
        traverse(q, is_dense ? dense(n) : sharded(level), n, level);
        rays_traced += n;
        if (!may_emit) {   // no material spawns rays and shadows are off: shade and return, nothing to read back (asynchronous)
            shade(q, dense(n), n, -1, level + 1);
            release(q);
            return;
        }
        // Batches sized so that the children of a batch always fit: of a dense queue by entries, of a sharded one by the
        // same range of every shard (a wave's children stay in its shard: a shard's batch times the fan-out fits a shard).
        const size_t fo = fanout(level);
        unsigned span = n, batch = (unsigned)std::max<size_t>(1, std::min<size_t>(n, ws.capacity / fo));
        if (!is_dense) {
            span = batch = region();
            if ((size_t)n * fo > ws.capacity) {   // more than one batch: how far the fullest shard reaches
                std::vector<unsigned> t((size_t)kShards * kCountStride);
                if (!ok(hipMemcpyAsync(t.data(), tails(level), t.size() * sizeof(unsigned), hipMemcpyDeviceToHost, stream)) || !ok(hipStreamSynchronize(stream))) {
                    release(q);
                    return;
                }
                span = 0;
                for (unsigned r = 0; r < K; ++r) span = std::max(span, std::min(t[(size_t)r * kCountStride], region()));
                batch = (unsigned)std::max<size_t>(1, (ws.capacity / fo) / K);
            }
        }
        for (unsigned start = 0; start < span && err == hipSuccess; start += batch) {
            const unsigned m = std::min(batch, span - start);
            const unsigned m_total = is_dense ? m : (unsigned)std::min<size_t>(n, (size_t)m * K);
            const int qo = acquire();
            if (qo < 0) break;
            // overflow flag, the tails of the children's queue and of the bundles
            if (!ok(hipMemsetAsync(ws.counters + 2, 0, sizeof(unsigned), stream)) || !ok(hipMemsetAsync(tails(level + 1), 0, tails_bytes(), stream)) ||
                !ok(hipMemsetAsync(bundle_tails(), 0, tails_bytes(), stream))) { release(qo); break; }
            shade(q, is_dense ? dense(n, start, start + m) : sharded(level, start, start + m), m_total, qo, level + 1);
            shadow(m_total, qo, level + 1);
            sum_tails(level + 1);
            unsigned local[3] = {0, 0, 0};
            unsigned* res = ws.host_readback ? ws.host_readback : local;   // words 2 .. 4: overflow flag, -, entries queued
            if (!ok(hipMemcpyAsync(res, ws.counters + 2, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, stream)) || !ok(hipStreamSynchronize(stream))) {
                release(qo);
                break;
            }
            if (res[0]) { ok(hipErrorOutOfMemory); release(qo); break; }  // queue overflow: cannot happen with the batching above
            if (start + m >= span) { release(q); q = -1; }                 // input fully consumed: its buffers may be reused below
            const unsigned n_children = res[2];
            process(qo, n_children, level + 1);
        }
        if (q >= 0) release(q);
    }
};

}  // namespace

void wavefront_workspace_free(WavefrontWorkspace& ws) {
    if (ws.partner) {
        wavefront_workspace_free(*ws.partner);
        delete ws.partner;
        ws.partner = nullptr;
    }
    if (ws.partner_stream) (void)hipStreamDestroy(ws.partner_stream);
    ws.partner_stream = nullptr;
    if (!ws.fixed) {
        for (void* p : ws.queues) (void)hipFree(p);
        for (void* p : ws.hit_buffers) (void)hipFree(p);
        if (ws.bundles) (void)hipFree(ws.bundles);
        if (ws.accum) (void)hipFree(ws.accum);
        if (ws.counters) (void)hipFree(ws.counters);
        if (ws.shard_counts) (void)hipFree(ws.shard_counts);
        if (ws.stats) (void)hipFree(ws.stats);
        if (ws.chunk_cost) (void)hipFree(ws.chunk_cost);
        if (ws.chunk_order) (void)hipFree(ws.chunk_order);
    }
    for (hipEvent_t e : ws.events) (void)hipEventDestroy(e);
    ws = WavefrontWorkspace();
}

namespace {

bool frame_may_emit(const DScene& scene, const DParams& params) {
    return !params.normals && ((params.shadows && scene.n_lights > 0) || scene.spawning_materials != 0);
}
// worst-case children of one primary ray
size_t first_level_fanout(const DScene& scene, const DParams& params) {
    const size_t refl = (params.max_depth < params.max_bounces) ? 1 : (size_t)std::max(params.glossy_samples, 1);
    return refl + 1 + (params.shadows ? 2ull * (size_t)scene.n_lights * (size_t)params.shadow_samples : 0ull);
}
// levels of the ray tree: radiance rays of depth max_depth .. 1, the shadow rays of the last of them, and a few
// segments of shadow rays that continue through transmitting surfaces (longer chains are finished level by level)
int tree_levels(const DScene& scene, const DParams& params) {
    return std::max(1, params.max_depth) + (params.shadows ? 1 : 0) + ((params.shadows && scene.transparent_materials) ? 4 : 0);
}

hipError_t drop_queues(WavefrontWorkspace& ws, hipStream_t stream) {
    hipError_t e = hipStreamSynchronize(stream);   // nothing of this stream's workspace is in flight afterwards
    if (e != hipSuccess) return e;
    for (void* p : ws.queues) (void)hipFree(p);
    for (void* p : ws.hit_buffers) (void)hipFree(p);
    if (ws.bundles) (void)hipFree(ws.bundles);
    ws.bundles = nullptr;
    ws.queues.clear();
    ws.hit_buffers.clear();
    ws.in_use.clear();
    return hipSuccess;
}

hipError_t ensure_small_buffers(WavefrontWorkspace& ws) {
    if (ws.counters) return hipSuccess;
    hipError_t e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&ws.counters), (kNextBase + kXcds * kNextStride) * sizeof(unsigned))) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&ws.stats), RT_N_COUNTERS * sizeof(unsigned long long))) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&ws.shard_counts), kShardCountBytes)) != hipSuccess) return e;
    if ((e = hipMemset(ws.shard_counts, 0, kShardCountBytes)) != hipSuccess) return e;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    ws.compute_units = cus;
    const char* rpl = getenv("RT_RAYS_PER_LANE");
    ws.rays_per_lane = rpl ? std::max(1, atoi(rpl)) : RT_RAYS_PER_LANE;
    return hipSuccess;
}

// Sizes and allocates what a frame needs of an owned (not fixed) workspace.
hipError_t prepare_workspace(const DScene& scene, const DParams& params, hipStream_t stream, WavefrontWorkspace& ws, size_t npix,
                             bool may_emit, size_t* max_queue_rays, bool need_accum = true) {
    hipError_t e;
    // Queue capacity: what one chunk of primary rays and its children need, up to RT_QUEUE_RAYS (default 2^26 rays =
    // 9.7 GB per queue; a frame that needs more is rendered in chunks of rows x samples and batches of rays).  Large
    // queues mean few batches, i.e. few launches and few host round trips per level: HBM is there to be used.
    const char* env = getenv("RT_QUEUE_RAYS");
    const size_t env_cap = env ? strtoull(env, nullptr, 10) : 0;
    // Default ceiling: 2^26 rays per queue; 2^27 where the frame can use it and the device has the room (a workspace of such
    // queues is ~70 GB: three queues with their hit records and the bundles) -- a frame of many samples is cut into chunks whose
    // every ray-tree level is a few launches and whose end is a read-back, so twice the queue is half of those (4K x 16 spp
    // with two area lights: 101 -> 92 ms).
    size_t default_cap = kDefaultMaxQueueRays;
    if (env_cap < 4096 && ws.capacity < 2 * kDefaultMaxQueueRays && npix * (size_t)params.spp * (may_emit ? first_level_fanout(scene, params) : 1) > kDefaultMaxQueueRays) {
        size_t free_bytes = 0, total_bytes = 0;
        if (hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess && free_bytes >= (size_t(160) << 30)) default_cap = 2 * kDefaultMaxQueueRays;
    } else if (ws.capacity >= 2 * kDefaultMaxQueueRays) {
        default_cap = 2 * kDefaultMaxQueueRays;   // this stream already holds such queues
    }
    const size_t max_cap = std::min(env_cap >= 4096 ? env_cap : std::max(default_cap, ws.grown_capacity), size_t(1) << 31);   // slots are 32-bit
    const size_t fan = may_emit ? first_level_fanout(scene, params) : 1;
    size_t want = kMinQueueRays;
    const char* head = getenv("RT_QUEUE_HEADROOM");   // queues this many times larger than the first level needs (deep ray trees)
    const size_t headroom = std::max<size_t>(1, head ? strtoull(head, nullptr, 10) : 1) * ws.grow;
    while (want < max_cap && want < npix * (size_t)params.spp * fan * headroom) want <<= 1;
    want = std::min(want, max_cap);
    *max_queue_rays = env_cap >= 4096 ? max_cap : kGrownMaxQueueRays;   // how far run_levels may grow the queues
    // a larger frame than before, or RT_QUEUE_RAYS now asks for smaller queues than this stream holds (tests): drop the queues
    if (ws.capacity != 0 && (ws.capacity < want || ws.capacity > max_cap)) {
        if ((e = drop_queues(ws, stream)) != hipSuccess) return e;
        ws.capacity = want;
    }
    if (ws.capacity == 0) ws.capacity = want;
    if ((e = ensure_small_buffers(ws)) != hipSuccess) return e;
    if (may_emit && params.shadows && scene.n_lights > 0 && !ws.bundles &&
        (e = hipMalloc(&ws.bundles, ws.capacity * sizeof(Bundle))) != hipSuccess) return e;
    if (need_accum && ws.accum_pixels < npix) {
        if (ws.accum) (void)hipFree(ws.accum);
        ws.accum = nullptr;
        ws.accum_pixels = 0;
        if ((e = hipMalloc(reinterpret_cast<void**>(&ws.accum), npix * 3 * sizeof(double))) != hipSuccess) return e;
        ws.accum_pixels = npix;
    }
    return hipSuccess;
}

}  // namespace

hipError_t wavefront_reserve(WavefrontWorkspace& ws, size_t rays_per_queue, int n_queues, hipStream_t stream) {
    if (ws.fixed) return hipErrorInvalidValue;
    hipError_t e;
    size_t want = kMinQueueRays;
    while (want < rays_per_queue && want < (size_t(1) << 31)) want <<= 1;
    if (ws.capacity < want) {
        if ((e = drop_queues(ws, stream)) != hipSuccess) return e;
        ws.capacity = want;
        ws.grown_capacity = std::max(ws.grown_capacity, want);   // a later frame keeps this size even above the default maximum
    }
    if ((e = ensure_small_buffers(ws)) != hipSuccess) return e;
    while ((int)ws.queues.size() < n_queues) {
        void *q = nullptr, *h = nullptr;
        if ((e = hipMalloc(&q, ws.capacity * sizeof(RayRec))) != hipSuccess) return e;
        if ((e = hipMalloc(&h, ws.capacity * sizeof(HitRec))) != hipSuccess) { (void)hipFree(q); return e; }
        ws.queues.push_back(q);
        ws.hit_buffers.push_back(h);
        ws.in_use.push_back(0);
    }
    if (!ws.bundles && (e = hipMalloc(&ws.bundles, ws.capacity * sizeof(Bundle))) != hipSuccess) return e;
    // first touch of every page here, not inside the first frame
    if ((e = hipMemsetAsync(ws.bundles, 0, ws.capacity * sizeof(Bundle), stream)) != hipSuccess) return e;
    for (size_t i = 0; i < ws.queues.size(); ++i) {
        if ((e = hipMemsetAsync(ws.queues[i], 0, ws.capacity * sizeof(RayRec), stream)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(ws.hit_buffers[i], 0, ws.capacity * sizeof(HitRec), stream)) != hipSuccess) return e;
    }
    return hipStreamSynchronize(stream);
}

hipError_t launch_resolve_sum(const double* d_sum, unsigned long long passes, size_t npix, int tonemap_mode, uint8_t* d_rgb8, hipStream_t stream) {
    if (npix == 0) return hipSuccess;
    hipLaunchKernelGGL(k_resolve_sum, dim3((unsigned)((npix + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, d_sum, d_rgb8, npix,
                       1.0 / (double)(passes ? passes : 1ull), tonemap_mode);
    return hipGetLastError();
}

hipError_t launch_make_xforms32(const DXform* xforms, int n, float* out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_make_xforms32, dim3((unsigned)(((size_t)n * 16 + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, xforms, n, out);
    return hipGetLastError();
}

hipError_t launch_render_wavefront(const DScene& scene, const DParams& params, uint8_t* d_rgb8, unsigned long long* d_counters,
                                   hipStream_t stream, WavefrontWorkspace& ws, double* d_linear, int add_linear) {
    // pixels of the output: `frames` slices of slice_rows rows each (one frame of part_rows rows unless rt_render_frames_device calls)
    const size_t npix = (size_t)params.frames * params.slice_rows * params.width;
    const int launch_rows = params.frames * params.part_rows;   // rows of primary rays
    if (npix == 0 || launch_rows == 0) return hipSuccess;
    hipError_t e;
    const bool may_emit = frame_may_emit(scene, params);
    size_t max_queue_rays = ws.capacity;
    if (!ws.fixed && (e = prepare_workspace(scene, params, stream, ws, npix, may_emit, &max_queue_rays)) != hipSuccess) return e;
    if (ws.levels_overflowed && ws.overflowed_scene != scene.serial) ws.levels_overflowed = false;   // another scene: try again
    // the statistics words: all of them for a counting pass; else words 7 and 3 -- rays traced inside k_shadow, and those it counted
    // without tracing (4 .. 6 lie between them and belong to the counting pass)
    if ((e = params.count_tests ? hipMemsetAsync(ws.stats, 0, RT_N_COUNTERS * sizeof(unsigned long long), stream)
                                : hipMemsetAsync(ws.stats + 3, 0, 5 * sizeof(unsigned long long), stream)) != hipSuccess) return e;

    if (ws.events_used + 64 > kMaxTimedEvents) ws.events_used = 0;   // nobody collected them (wavefront_traverse_time): start over
    const size_t first_event = ws.events_used;
    Driver D{scene, params, ws, stream, 0, 0};
    D.may_emit = may_emit;
    D.max_capacity = max_queue_rays;
    D.accum = ws.accum;
    if (!D.may_emit && params.spp == 1 && !d_linear) D.direct = d_rgb8;
    else if ((e = hipMemsetAsync(ws.accum, 0, npix * 3 * sizeof(double), stream)) != hipSuccess) return e;
    const int f = scene.features;
    // shading: spheres only / analytic shapes / analytic shapes with textures or bump maps (no ray marcher, no environment: the floor
    // texture of config 5 must not cost it the 278 registers -- one wave per SIMD -- of the full variant) / everything
    D.variant = ((f & ~kSpheres) == 0) ? 0 : ((f & ~kAnalytic) == 0) ? 1 : ((f & ~kTextured) == 0) ? 3 : 2;
    D.trav_variant = (f & RT_FEAT_COMPLEX) ? 2 : (f & (RT_FEAT_QUAD | RT_FEAT_CUBE)) ? 1 : 0;
    D.shadow_variant = (f & RT_FEAT_COMPLEX) ? 2 : ((f & RT_FEAT_TEXTURE) && scene.transparent_materials != 0) ? 3 : D.trav_variant;
    if (getenv("RT_NO_GEOMETRY_VARIANTS")) D.trav_variant = D.shadow_variant = D.variant;   // (A/B runs)
    // persistent traversal grid: as many workgroups per CU as the variant's registers and LDS stacks (30 KB per 256-thread block) allow
    D.traverse_blocks = ws.compute_units * (D.trav_variant == 0 ? RT_TRAVERSE_WAVES_SPHERES : RT_TRAVERSE_WAVES);
    if (const char* tb = getenv("RT_TRAVERSE_BLOCKS_PER_CU")) D.traverse_blocks = ws.compute_units * std::max(1, atoi(tb));
    // RT_TRAVERSE_SPARE_BLOCKS: workgroups left out of the persistent grid, so that kernels of other streams -- the RCCL
    // send / receive kernels of an exchange that overlaps the render -- find a free slot before the launch has drained
    if (const char* sb = getenv("RT_TRAVERSE_SPARE_BLOCKS")) D.traverse_blocks = std::max(ws.compute_units, D.traverse_blocks - std::max(0, atoi(sb)));

    // Primary rays are generated in chunks of whole 8-row bands x a batch of samples, sized so that the first
    // level's children always fit one queue.
    const size_t fo0 = may_emit ? D.fanout(0) : 1;   // nothing is queued behind the primary rays of such a frame
    // Frames of a small scene are many small levels whose cost is the queue tails: their queues are sharded (QIn).  The
    // waves of the dense first level are dealt to the shards in turn: a shard gets one wave more than its share at most.
    D.K = (may_emit && D.small_scene() && ws.capacity >= (size_t(1) << 16) && getenv("RT_NO_SHARDS") == nullptr) ? kShards : 1u;
    const size_t slack = D.K > 1 ? (size_t)128 * D.K : 0;
    const size_t max_primary = std::max<size_t>(64, ws.capacity / fo0 > slack + 64 ? ws.capacity / fo0 - slack : 64);
    const int W = params.width;
    int ns = (int)std::min<size_t>((size_t)params.spp, std::max<size_t>(1, max_primary / ((size_t)W * 8)));
    int rows_per_chunk = (int)std::min<size_t>((size_t)launch_rows, std::max<size_t>(8, (max_primary / ((size_t)W * ns)) / 8 * 8));
    if ((size_t)rows_per_chunk * W * ns > ws.capacity) return hipErrorInvalidValue;  // frame wider than the queue: raise RT_QUEUE_RAYS
    // A frame is first tried without host round trips between the levels of its ray tree: every chunk of primary rays
    // enqueues all its levels and is read back once (Driver::run_levels).  A level that outgrows a queue shard makes the
    // FRAME start over level by level with batches (what the aborted attempt added to the accumulators, the test
    // counters and the traversal timing is dropped), and the stream's later frames of that scene go that way at once.
    // RT_SYNC_LEVELS=1 keeps the level-by-level path (A/B runs, tests).
    const int levels = tree_levels(scene, params);
    D.allow_grow = rows_per_chunk >= launch_rows && ns >= params.spp;
    std::vector<Chunk> chunks;
    for (int row0 = 0; row0 < launch_rows; row0 += rows_per_chunk)
        for (int s0 = 0; s0 < params.spp; s0 += ns)
            chunks.push_back(Chunk{row0, std::min(rows_per_chunk, launch_rows - row0), s0, std::min(ns, params.spp - s0)});
    // one chunk through driver X: primary rays, then its ray tree.  0 = done, 1 = a level outgrew its queue, -1 = error
    auto run_chunk = [&](Driver& X, const Chunk& ck, bool optimistic) -> int {
        const long long n = (long long)ck.rows * W * ck.ns;
        const int q = X.acquire();
        if (q < 0) return -1;
        X.chunk = ck;
        hipLaunchKernelGGL(k_raygen, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, X.stream, scene, params, X.rays(q, 0), n,
                           X.ws.counters + kNextBase);
        if (!optimistic) {
            X.process(q, (unsigned)n, 0);
            return X.err == hipSuccess ? 0 : -1;
        }
        const int lr = X.run_levels(q, (unsigned)n, levels);
        if (getenv("RT_DEBUG_LEVELS")) fprintf(stderr, "[rt levels] capacity %zu grow %zu -> %s\n", X.ws.capacity, X.ws.grow, lr == 0 ? "done" : "overflow");
        return X.err != hipSuccess ? -1 : lr;
    };
    // A frame of many chunks renders them TWO AT A TIME: a chunk is a dozen levels of launches, the deep ones a few microseconds
    // of work each, and a read-back the host waits for, ~0.4 ms whatever the chunk's size; the odd chunks go through a second
    // set of queues (ws.partner) on a stream of its own, enqueued by a second host thread, and add into the same accumulators,
    // so that one chunk's small launches and round trip hide behind the other's large launches (RT_ONE_CHUNK_AT_A_TIME: A/B).
    unsigned long long partner_rays = 0;
    bool partner_used = false;
    bool partner_failed = false;   // the second set of queues could not be had (or ran out of memory mid-frame): this frame renders its chunks one by one
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool optimistic = attempt == 0 && may_emit && !ws.levels_overflowed && getenv("RT_SYNC_LEVELS") == nullptr;
        bool overflowed = false;
        bool pairs = optimistic && !partner_failed && chunks.size() >= 4 && !ws.fixed && !params.count_tests && getenv("RT_ONE_CHUNK_AT_A_TIME") == nullptr;
        if (pairs && !ws.partner) {
            size_t free_bytes = 0, total_bytes = 0;
            const size_t workspace_bytes = ws.capacity * (4 * (sizeof(RayStore) + sizeof(HitRec)) + sizeof(Bundle));
            if (hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess || free_bytes < workspace_bytes + (size_t(24) << 30)) pairs = false;
            else {
                ws.partner = new WavefrontWorkspace();
                if (hipStreamCreateWithFlags(&ws.partner_stream, hipStreamNonBlocking) != hipSuccess) { delete ws.partner; ws.partner = nullptr; pairs = false; }
            }
        }
        if (pairs) {
            WavefrontWorkspace& w2 = *ws.partner;
            size_t mq2 = 0;
            w2.grow = ws.grow;
            w2.grown_capacity = ws.grown_capacity;
            // The headroom was checked when the partner was created; frames, grown queues or the caller's own allocations may
            // have used the device up since.  Without its partner a frame renders one chunk at a time: never an error.
            const char* test_fail = getenv("RT_TEST_PARTNER_FAIL");   // tests: "prepare" = the partner cannot be had, "acquire" = it runs out mid-frame
            if ((test_fail && strcmp(test_fail, "prepare") == 0) ||
                prepare_workspace(scene, params, ws.partner_stream, w2, npix, may_emit, &mq2, false) != hipSuccess) {
                (void)hipGetLastError();
                (void)drop_queues(w2, ws.partner_stream);
                w2.capacity = 0;
                pairs = false;
            } else if (w2.capacity != ws.capacity) {
                pairs = false;   // (cannot happen: same frame, same rules)
            }
        }
        if (pairs) {
            WavefrontWorkspace& w2 = *ws.partner;
            hipStream_t s2 = ws.partner_stream;
            Driver D2{scene, params, w2, s2, 0, 0};
            D2.may_emit = D.may_emit; D2.max_capacity = D.max_capacity; D2.accum = ws.accum; D2.direct = D.direct;
            D2.variant = D.variant; D2.trav_variant = D.trav_variant; D2.shadow_variant = D.shadow_variant;
            D2.traverse_blocks = D.traverse_blocks; D2.K = D.K; D2.allow_grow = false;
            if (const char* tf = getenv("RT_TEST_PARTNER_FAIL")) if (strcmp(tf, "acquire") == 0) D2.fail_acquire_in = 3;
            if (w2.events_used + 64 > kMaxTimedEvents) w2.events_used = 0;
            hipEvent_t ev0 = nullptr, ev1 = nullptr;
            if ((e = hipEventCreateWithFlags(&ev0, hipEventDisableTiming)) != hipSuccess) return e;
            if ((e = hipEventCreateWithFlags(&ev1, hipEventDisableTiming)) != hipSuccess) { (void)hipEventDestroy(ev0); return e; }
            (void)hipEventRecord(ev0, stream);            // the accumulators are cleared before either stream adds to them
            (void)hipStreamWaitEvent(s2, ev0, 0);
            (void)hipMemsetAsync(w2.stats + 3, 0, 5 * sizeof(unsigned long long), s2);
            int device = 0;
            (void)hipGetDevice(&device);
            std::atomic<bool> stop{false};
            int result2 = 0;
            std::thread odd;
            bool threaded = true;
            try {
                odd = std::thread([&]() {
                    if (hipSetDevice(device) != hipSuccess) { result2 = -1; stop = true; return; }
                    for (size_t i = 1; i < chunks.size() && !stop.load(); i += 2) {
                        result2 = run_chunk(D2, chunks[i], true);
                        if (result2 != 0) stop = true;
                    }
                });
            } catch (...) {   // no thread to be had: this one renders every chunk (nothing may leave through the C ABI)
                threaded = false;
            }
            int result1 = 0;
            for (size_t i = 0; i < chunks.size() && !stop.load(); i += threaded ? 2 : 1) {
                result1 = run_chunk(D, chunks[i], true);
                if (result1 != 0) stop = true;
            }
            if (threaded) odd.join();
            (void)hipEventRecord(ev1, s2);
            (void)hipStreamWaitEvent(stream, ev1, 0);      // whatever follows on the caller's stream sees both halves
            (void)hipEventDestroy(ev0);
            (void)hipEventDestroy(ev1);
            if (result2 < 0 && result1 >= 0 && D2.err == hipErrorOutOfMemory && !partner_failed) {
                // The partner's lazily allocated queues did not fit the device any more.  Whatever both halves have added so far is
                // dropped (as after a queue overflow) and the frame starts over on the caller's stream alone.
                (void)hipGetLastError();
                if ((e = hipStreamSynchronize(s2)) != hipSuccess || (e = hipStreamSynchronize(stream)) != hipSuccess) return e;
                (void)drop_queues(w2, s2);
                w2.capacity = 0;
                partner_failed = true;
                for (char& u : ws.in_use) u = 0;
                if ((e = hipMemsetAsync(ws.accum, 0, npix * 3 * sizeof(double), stream)) != hipSuccess) return e;
                if ((e = hipMemsetAsync(ws.stats + 3, 0, 5 * sizeof(unsigned long long), stream)) != hipSuccess) return e;
                ws.events_used = first_event;
                D.rays_traced = 0;
                D.err = hipSuccess;
                --attempt;
                continue;
            }
            if (result1 < 0 || result2 < 0) return D.err != hipSuccess ? D.err : (D2.err != hipSuccess ? D2.err : hipErrorUnknown);
            overflowed = result1 == 1 || result2 == 1;
            if (w2.levels_overflowed) { ws.levels_overflowed = true; ws.overflowed_scene = scene.serial; w2.levels_overflowed = false; }
            partner_rays = D2.rays_traced;
            partner_used = !overflowed;
        } else {
            for (size_t i = 0; i < chunks.size() && D.err == hipSuccess && !overflowed; ++i) overflowed = run_chunk(D, chunks[i], optimistic) == 1;
        }
        if (!overflowed || D.err != hipSuccess) break;
        if ((e = hipMemsetAsync(ws.accum, 0, npix * 3 * sizeof(double), stream)) != hipSuccess) return e;
        if ((e = params.count_tests ? hipMemsetAsync(ws.stats, 0, RT_N_COUNTERS * sizeof(unsigned long long), stream)
                                    : hipMemsetAsync(ws.stats + 3, 0, 5 * sizeof(unsigned long long), stream)) != hipSuccess) return e;
        ws.events_used = first_event;
        D.rays_traced = 0;
        partner_rays = 0;
    }
    if (D.err != hipSuccess) return D.err;
    if (d_linear)
        hipLaunchKernelGGL(k_resolve_linear, dim3((unsigned)((npix * 3 + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, ws.accum, d_linear,
                           npix * 3, params.spp, add_linear);
    else if (!D.direct)
        hipLaunchKernelGGL(k_resolve, dim3((unsigned)((npix + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, ws.accum, d_rgb8, npix,
                           params.spp, params.tonemap);
    // Rays are counted on the host (every traversed queue entry is one closest-hit query).  With count_tests the call
    // also waits for the frame, reads the test counters and the traversal time of this frame's launches; without, it
    // stays asynchronous and the traversal time is collected later (wavefront_traverse_time).
    if (d_counters && params.count_tests) {
        ws.host_counts[0] = D.rays_traced;
        if ((e = hipMemcpyAsync(ws.host_counts + 1, ws.stats + 1, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
        double ms = 0.0;
        for (size_t i = first_event; i + 1 < ws.events_used; i += 2) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, ws.events[i], ws.events[i + 1]) == hipSuccess) ms += t;
        }
        ws.events_used = first_event;
        hipLaunchKernelGGL(k_add_counters, dim3(1), dim3(64), 0, stream, d_counters, ws.host_counts[0], ws.host_counts[1], ws.host_counts[2],
                           (unsigned long long)(ms * 1e6), ws.host_counts[4], ws.host_counts[5], ws.host_counts[6], ws.stats + 7, ws.stats + 3);
    } else if (d_counters) {
        hipLaunchKernelGGL(k_add_counters, dim3(1), dim3(64), 0, stream, d_counters, D.rays_traced, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull,
                           ws.stats + 7, ws.stats + 3);
        if (partner_used)
            hipLaunchKernelGGL(k_add_counters, dim3(1), dim3(64), 0, stream, d_counters, partner_rays, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull,
                               ws.partner->stats + 7, ws.partner->stats + 3);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ no-allocation variant
// Callers that cannot call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid it) get
// the same wavefront render through a workspace carved out of this array: queue counters, statistics words, FP64
// accumulators and as many ray queues as fit.  Every GPU the library runs on holds its own copy of the array at its
// own address until the process ends; `scratch` itself is only the host-side stand-in of the symbol, so host code
// resolves the address once per device (wavefront_static_setup, from rt_init) and uses that pointer everywhere.
// All calls on a GPU share the array, which keeps the last call's contents: the driver clears what must start at zero
// (counters, accumulators) on the caller's stream, and calls that overlap on one GPU race.
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {
constexpr int kMaxDevices = 64;
unsigned char* g_scratch_ptr[kMaxDevices] = {};
unsigned* g_scratch_readback[kMaxDevices] = {};   // pinned host words for the queue-size read-backs of this variant
constexpr size_t kStaticCountersOffset = 0;                       // (kNextBase + 256) unsigned words
constexpr size_t kStaticStatsOffset = 4096;                       // RT_N_COUNTERS u64
constexpr size_t kStaticShardsOffset = 8192;                      // tails of the queue shards (kShardCountBytes)
constexpr size_t kStaticAccumOffset = kStaticShardsOffset + (kShardCountBytes + 4095) / 4096 * 4096;   // kStaticMaxPixels x 3 doubles
constexpr size_t kStaticQueuesOffset = kStaticAccumOffset + (size_t)kStaticMaxPixels * 3 * sizeof(double);
static_assert((kNextBase + kXcds * kNextStride) * sizeof(unsigned) <= kStaticStatsOffset, "counters overlap the statistics words");
static_assert(kStaticQueuesOffset % 256 == 0 && kStaticQueuesOffset < kStaticScratchBytes, "static scratch layout");
}  // namespace

hipError_t wavefront_static_setup(int device) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    if (g_scratch_ptr[device]) return hipSuccess;
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));   // the current device's copy of the array
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    void* h = nullptr;
    if ((e = hipHostMalloc(&h, 16 * sizeof(unsigned), hipHostMallocDefault)) != hipSuccess) return e;
    g_scratch_readback[device] = static_cast<unsigned*>(h);
    g_scratch_ptr[device] = static_cast<unsigned char*>(p);
    return hipSuccess;
}

hipError_t launch_render_wavefront_static(const DScene& scene, const DParams& params, uint8_t* d_rgb8, hipStream_t stream, int device) {
    if (device < 0 || device >= kMaxDevices || !g_scratch_ptr[device]) return hipErrorNotInitialized;
    const size_t npix = (size_t)params.part_rows * params.width;
    if (npix == 0) return hipSuccess;
    if (npix > (size_t)kStaticMaxPixels || params.width > kStaticMaxWidth) return hipErrorInvalidValue;
    unsigned char* base = g_scratch_ptr[device];   // never `scratch`: that name is not a device address in host code
    // As many queues as the ray tree is deep (one per level in flight, one being filled, one spare), as large as fits.
    const size_t need = (size_t)tree_levels(scene, params) + 2;
    const size_t room = kStaticScratchBytes - kStaticQueuesOffset;
    size_t capacity = size_t(1) << 20;
    while (capacity > 4096 && capacity * ((sizeof(RayRec) + sizeof(HitRec)) * need + sizeof(Bundle)) > room) capacity >>= 1;
    if ((size_t)params.width * 8 > capacity) return hipErrorInvalidValue;
    const size_t n_queues = (room - capacity * sizeof(Bundle)) / (capacity * (sizeof(RayRec) + sizeof(HitRec)));
    WavefrontWorkspace ws;
    ws.fixed = true;
    ws.capacity = capacity;
    ws.counters = reinterpret_cast<unsigned*>(base + kStaticCountersOffset);
    ws.stats = reinterpret_cast<unsigned long long*>(base + kStaticStatsOffset);
    ws.shard_counts = reinterpret_cast<unsigned*>(base + kStaticShardsOffset);
    ws.accum = reinterpret_cast<double*>(base + kStaticAccumOffset);
    ws.accum_pixels = (size_t)kStaticMaxPixels;
    ws.host_readback = g_scratch_readback[device];
    ws.bundles = base + kStaticQueuesOffset;
    unsigned char* q = base + kStaticQueuesOffset + capacity * sizeof(Bundle);
    for (size_t i = 0; i < n_queues; ++i) {
        ws.queues.push_back(q);
        ws.hit_buffers.push_back(q + capacity * sizeof(RayRec));
        ws.in_use.push_back(0);
        q += capacity * (sizeof(RayRec) + sizeof(HitRec));
    }
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    ws.compute_units = cus;
    ws.rays_per_lane = RT_RAYS_PER_LANE;
    // the array keeps the previous call's contents: everything the kernels expect at zero is cleared on this stream
    hipError_t e = hipMemsetAsync(base, 0, kStaticAccumOffset, stream);
    if (e != hipSuccess) return e;
    e = launch_render_wavefront(scene, params, d_rgb8, nullptr, stream, ws, nullptr);
    // the workspace object dies here; its memory is the array's.  The frame's last kernels may still be running on `stream`.
    return e;
}

hipError_t wavefront_traverse_time(WavefrontWorkspace& ws, double* ms_out, unsigned* launches_out) {
    double ms = 0.0;
    unsigned launches = 0;
    for (size_t i = 0; i + 1 < ws.events_used; i += 2) {
        float t = 0.f;
        hipError_t e = hipEventElapsedTime(&t, ws.events[i], ws.events[i + 1]);
        if (e != hipSuccess) return e;   // hipErrorNotReady: the stream has not been synchronised
        ms += t;
        ++launches;
    }
    ws.events_used = 0;
    *ms_out = ms;
    *launches_out = launches;
    return hipSuccess;
}

}  // namespace rt
