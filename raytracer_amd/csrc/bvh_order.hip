// The reference's visit order of the shapes, computed on the GPU.
//
// The reference builds its tree top-down (Code/acceleration/bvh.cpp:32-92): the box of a span picks the axis (longest,
// strict comparisons, x on ties), std::nth_element puts the span/2 shapes with the smallest box centres on that axis to
// the left, pairs are ordered by that centre.  Its depth-first leaf order is the order in which BVHNode::intersect
// (bvh.cpp:101-114) meets the shapes, which decides hits at exactly equal distance (device_rt.h: tie_new_wins), so
// primitives are numbered in it (host_bvh.cpp: build_bvh does the same thing on host threads).
//
// Two facts make a level-synchronous GPU version exact.  (1) Span sizes depend on the shape count only, so the spans
// of a level are known without looking at the data.  (2) When the centres of a span are distinct at the split, WHICH
// shapes go left is determined by the keys alone -- whatever std::nth_element does inside -- and the order inside a
// half does not matter to the levels below, which select by key again.  So every level SORTS its spans by the key of
// the span's axis (two stable radix sorts over all shapes: by key, then by span), and halves them.  Where two equal
// centres meet at a split, or in a pair, the reference's result depends on the inner workings of its nth_element: the
// kernels flag it and the caller falls back to the host pass.  Scenes of random shapes never do; grids of quads do and
// are small.
//
// Per level: k_order_bounds (span boxes: serial for short spans, wave reduction + 64-bit atomics on order-preserving
// encodings for long ones), k_order_keys, rocPRIM radix_sort_pairs x 2, k_order_apply (+ tie check), k_order_split.
// HBM-bound integer/byte work.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>   // rocprim/iterator/texture_cache_iterator.hpp calls memset without including it

#include <rocprim/rocprim.hpp>

#include "bvh_build.h"

namespace rt {

namespace {

constexpr int kB = 256;
constexpr int kShortSpan = 32;   // spans up to this size are reduced by one thread

typedef unsigned long long u64;

__device__ __forceinline__ u64 encode(double d) {   // order-preserving: a < b  <=>  encode(a) < encode(b)
    const u64 b = (u64)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double decode(u64 e) {
    const u64 b = (e >> 63) ? (e & 0x7fffffffffffffffull) : ~e;
    return __longlong_as_double((long long)b);
}

// Span boxes, indexed by the span's first position: lo[3 * start + a] (min, encoded), hi[...] (max, encoded).
__global__ void __launch_bounds__(kB) k_order_bounds(const double* __restrict__ boxes, const int* __restrict__ perm, const int n,
                                                     const int* __restrict__ start, const int* __restrict__ end,
                                                     const unsigned char* __restrict__ done, u64* __restrict__ lo, u64* __restrict__ hi) {
    const int i = blockIdx.x * kB + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int s = -1, e = -1;
    bool active = false;
    if (i < n) {
        s = start[i];
        e = end[i];
        active = !done[i];
    }
    const bool longspan = active && (e - s) > kShortSpan;
    if (active && !longspan && i == s) {   // the first element of a short span reduces it
        double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = s; k < e; ++k) {
            const double* b = boxes + 6 * (size_t)perm[k];
            for (int a = 0; a < 3; ++a) {
                mn[a] = fmin(mn[a], b[a]);
                mx[a] = fmax(mx[a], b[3 + a]);
            }
        }
        for (int a = 0; a < 3; ++a) {
            lo[3 * (size_t)s + a] = encode(mn[a]);
            hi[3 * (size_t)s + a] = encode(mx[a]);
        }
    }
    // long spans: lanes of one wave mostly share a span; reduce in the wave when they all do.  Every lane of the wave
    // takes part in what follows (lanes outside long spans with the identity), because the shuffles below read all 64
    // lanes: a lane that had left the kernel would hand over whatever its registers hold.
    const unsigned long long m = __ballot(longspan);
    if (m == 0ull) return;
    u64 mn[3] = {~0ull, ~0ull, ~0ull}, mx[3] = {0ull, 0ull, 0ull};
    if (longspan) {
        const double* b = boxes + 6 * (size_t)perm[i];
        for (int a = 0; a < 3; ++a) {
            mn[a] = encode(b[a]);
            mx[a] = encode(b[3 + a]);
        }
    }
    const int first = __ffsll((long long)m) - 1;
    const int s0 = __shfl(s, first);
    const bool uniform = __ballot(longspan && s != s0) == 0ull;
    if (uniform) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            for (int a = 0; a < 3; ++a) {
                const u64 on = __shfl_xor(mn[a], off), ox = __shfl_xor(mx[a], off);
                mn[a] = on < mn[a] ? on : mn[a];
                mx[a] = ox > mx[a] ? ox : mx[a];
            }
        if (lane == first)
            for (int a = 0; a < 3; ++a) {
                atomicMin(&lo[3 * (size_t)s0 + a], mn[a]);
                atomicMax(&hi[3 * (size_t)s0 + a], mx[a]);
            }
    } else if (longspan) {
        for (int a = 0; a < 3; ++a) {
            atomicMin(&lo[3 * (size_t)s + a], mn[a]);
            atomicMax(&hi[3 * (size_t)s + a], mx[a]);
        }
    }
}

// Sort key of every position: the centre on its span's axis (bvh.cpp:52-58: longest extent, strict comparisons, x on
// ties; centre = (min + max) * 0.5 as in host_bvh.cpp), or the offset inside the span where the span is final.
__global__ void __launch_bounds__(kB) k_order_keys(const double* __restrict__ boxes, const int* __restrict__ perm, const int n,
                                                   const int* __restrict__ start, const unsigned char* __restrict__ done,
                                                   const u64* __restrict__ lo, const u64* __restrict__ hi, u64* __restrict__ keys,
                                                   unsigned* __restrict__ vals) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    vals[i] = (unsigned)i;
    const int s = start[i];
    if (done[i]) { keys[i] = (u64)(unsigned)(i - s); return; }
    double ext[3];
    for (int a = 0; a < 3; ++a) ext[a] = decode(hi[3 * (size_t)s + a]) - decode(lo[3 * (size_t)s + a]);
    int axis = 0;
    if (ext[1] > ext[0] && ext[1] > ext[2]) axis = 1;
    else if (ext[2] > ext[0] && ext[2] > ext[1]) axis = 2;
    const double* b = boxes + 6 * (size_t)perm[i];
    keys[i] = encode((b[axis] + b[3 + axis]) * 0.5);
}

__global__ void __launch_bounds__(kB) k_order_span_keys(const unsigned* __restrict__ src, const int n, const int* __restrict__ start,
                                                        unsigned* __restrict__ span_keys) {
    const int j = blockIdx.x * kB + threadIdx.x;
    if (j < n) span_keys[j] = (unsigned)start[src[j]];   // spans keep their positions: the span of source position src[j]
}

// Position p receives the shape that stood at source position src[p].
__global__ void __launch_bounds__(kB) k_order_apply(const unsigned* __restrict__ src, const int n, const int* __restrict__ perm_in,
                                                    const u64* __restrict__ keys_in, int* __restrict__ perm_out, u64* __restrict__ keys_out) {
    const int p = blockIdx.x * kB + threadIdx.x;
    if (p >= n) return;
    const unsigned s = src[p];
    perm_out[p] = perm_in[s];
    keys_out[p] = keys_in[s];
}

// Halve the spans (bvh.cpp:60-78: mid = span / 2; spans of one or two shapes are final) and flag equal keys where the
// reference's result would depend on its nth_element's internals.
__global__ void __launch_bounds__(kB) k_order_split(const int n, int* __restrict__ start, int* __restrict__ end, unsigned char* __restrict__ done,
                                                    const u64* __restrict__ keys, unsigned* __restrict__ flags) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n || done[i]) return;
    const int s = start[i], e = end[i], m = e - s;
    if (m <= 2) {
        if (m == 2 && i == s && keys[s] == keys[s + 1]) flags[0] = 1u;
        done[i] = 1;
        return;
    }
    const int mid = s + m / 2;
    if (i == s && keys[mid - 1] == keys[mid]) flags[0] = 1u;
    if (i < mid) end[i] = mid;
    else start[i] = mid;
    if ((i < mid ? mid - s : e - mid) == 1) done[i] = 1;
    flags[1] = 1u;   // some span is still being split
}

__global__ void __launch_bounds__(kB) k_order_init(const int n, int* __restrict__ perm, int* __restrict__ start, int* __restrict__ end,
                                                   unsigned char* __restrict__ done) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    perm[i] = i;
    start[i] = 0;
    end[i] = n;
    done[i] = n <= 1 ? 1 : 0;
}

size_t align_up(size_t v) { return (v + 255) & ~size_t(255); }

}  // namespace

OrderStatus reference_order_device(const std::vector<Box>& shape_boxes, hipStream_t stream, std::vector<int32_t>* order, float* device_ms,
                                   hipError_t* err) {
    *err = hipSuccess;
    const size_t n = shape_boxes.size();
    if (n < 2 || n > (size_t(1) << 28)) { *err = hipErrorInvalidValue; return OrderStatus::device_error; }
    const int ni = static_cast<int>(n);
    int span_bits = 1;
    while ((size_t(1) << span_bits) < n) ++span_bits;   // span keys are first positions: < n
    int levels = span_bits + 1;

    hipError_t e;
    size_t sort1_bytes = 0, sort2_bytes = 0;
    e = rocprim::radix_sort_pairs(nullptr, sort1_bytes, (u64*)nullptr, (u64*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, n, 0, 64, stream);
    if (e != hipSuccess) { *err = e; return OrderStatus::device_error; }
    e = rocprim::radix_sort_pairs(nullptr, sort2_bytes, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, n, 0,
                                  (unsigned)span_bits, stream);
    if (e != hipSuccess) { *err = e; return OrderStatus::device_error; }
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t at = off; off += align_up(bytes); return at; };
    const size_t o_boxes = carve(n * 6 * sizeof(double));
    const size_t o_lo = carve(n * 3 * sizeof(u64)), o_hi = carve(n * 3 * sizeof(u64));
    const size_t o_keys = carve(n * sizeof(u64)), o_keys2 = carve(n * sizeof(u64));
    const size_t o_vals = carve(n * sizeof(unsigned)), o_vals2 = carve(n * sizeof(unsigned)), o_vals3 = carve(n * sizeof(unsigned));
    const size_t o_skeys = carve(n * sizeof(unsigned)), o_skeys2 = carve(n * sizeof(unsigned));
    const size_t o_perm = carve(n * sizeof(int)), o_perm2 = carve(n * sizeof(int));
    const size_t o_start = carve(n * sizeof(int)), o_end = carve(n * sizeof(int)), o_done = carve(n);
    const size_t o_flags = carve(2 * sizeof(unsigned) * 64);   // one pair of flag words per level
    const size_t o_temp = carve(std::max(sort1_bytes, sort2_bytes));
    char* base = nullptr;
    if ((e = hipMalloc(reinterpret_cast<void**>(&base), off)) != hipSuccess) { *err = e; return OrderStatus::device_error; }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    auto fail = [&](hipError_t code) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(base);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        *err = code;
        return OrderStatus::device_error;
    };
#define ORDER_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(e_); } while (0)
    double* d_boxes = reinterpret_cast<double*>(base + o_boxes);
    u64* d_lo = reinterpret_cast<u64*>(base + o_lo);
    u64* d_hi = reinterpret_cast<u64*>(base + o_hi);
    u64* d_keys = reinterpret_cast<u64*>(base + o_keys);
    u64* d_keys2 = reinterpret_cast<u64*>(base + o_keys2);
    unsigned* d_vals = reinterpret_cast<unsigned*>(base + o_vals);
    unsigned* d_vals2 = reinterpret_cast<unsigned*>(base + o_vals2);
    unsigned* d_vals3 = reinterpret_cast<unsigned*>(base + o_vals3);
    unsigned* d_skeys = reinterpret_cast<unsigned*>(base + o_skeys);
    unsigned* d_skeys2 = reinterpret_cast<unsigned*>(base + o_skeys2);
    int* d_perm = reinterpret_cast<int*>(base + o_perm);
    int* d_perm2 = reinterpret_cast<int*>(base + o_perm2);
    int* d_start = reinterpret_cast<int*>(base + o_start);
    int* d_end = reinterpret_cast<int*>(base + o_end);
    unsigned char* d_done = reinterpret_cast<unsigned char*>(base + o_done);
    unsigned* d_flags = reinterpret_cast<unsigned*>(base + o_flags);
    void* d_temp = base + o_temp;
    static_assert(sizeof(Box) == 6 * sizeof(double), "Box is six doubles");
    if (levels > 64) levels = 64;

    ORDER_TRY(hipEventCreate(&ev0));
    ORDER_TRY(hipEventCreate(&ev1));
    ORDER_TRY(hipMemcpyAsync(d_boxes, shape_boxes.data(), n * 6 * sizeof(double), hipMemcpyHostToDevice, stream));
    ORDER_TRY(hipEventRecord(ev0, stream));
    ORDER_TRY(hipMemsetAsync(d_flags, 0, 2 * sizeof(unsigned) * 64, stream));
    const unsigned g = static_cast<unsigned>((n + kB - 1) / kB);
    hipLaunchKernelGGL(k_order_init, dim3(g), dim3(kB), 0, stream, ni, d_perm, d_start, d_end, d_done);
    for (int level = 0; level < levels; ++level) {
        ORDER_TRY(hipMemsetAsync(d_lo, 0xff, n * 3 * sizeof(u64), stream));
        ORDER_TRY(hipMemsetAsync(d_hi, 0x00, n * 3 * sizeof(u64), stream));
        hipLaunchKernelGGL(k_order_bounds, dim3(g), dim3(kB), 0, stream, d_boxes, d_perm, ni, d_start, d_end, d_done, d_lo, d_hi);
        hipLaunchKernelGGL(k_order_keys, dim3(g), dim3(kB), 0, stream, d_boxes, d_perm, ni, d_start, d_done, d_lo, d_hi, d_keys, d_vals);
        size_t tb = sort1_bytes;
        ORDER_TRY(rocprim::radix_sort_pairs(d_temp, tb, d_keys, d_keys2, d_vals, d_vals2, n, 0, 64, stream));
        hipLaunchKernelGGL(k_order_span_keys, dim3(g), dim3(kB), 0, stream, d_vals2, ni, d_start, d_skeys);
        tb = sort2_bytes;
        ORDER_TRY(rocprim::radix_sort_pairs(d_temp, tb, d_skeys, d_skeys2, d_vals2, d_vals3, n, 0, (unsigned)span_bits, stream));
        hipLaunchKernelGGL(k_order_apply, dim3(g), dim3(kB), 0, stream, d_vals3, ni, d_perm, d_keys, d_perm2, d_keys2);
        hipLaunchKernelGGL(k_order_split, dim3(g), dim3(kB), 0, stream, ni, d_start, d_end, d_done, d_keys2, d_flags + 2 * level);
        std::swap(d_perm, d_perm2);
    }
    ORDER_TRY(hipEventRecord(ev1, stream));
    std::vector<unsigned> flags(2 * 64, 0u);
    order->resize(n);
    ORDER_TRY(hipMemcpyAsync(flags.data(), d_flags, flags.size() * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    ORDER_TRY(hipMemcpyAsync(order->data(), d_perm, n * sizeof(int), hipMemcpyDeviceToHost, stream));
    ORDER_TRY(hipStreamSynchronize(stream));
    ORDER_TRY(hipGetLastError());
#undef ORDER_TRY
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ev0, ev1);
    if (device_ms) *device_ms = ms;
    (void)hipFree(base);
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    bool ties = false, unfinished = flags[2 * (levels - 1) + 1] != 0u;
    for (int level = 0; level < levels; ++level) ties = ties || flags[2 * level] != 0u;
    if (ties || unfinished) return OrderStatus::ties;
    return OrderStatus::ok;
}

}  // namespace rt
