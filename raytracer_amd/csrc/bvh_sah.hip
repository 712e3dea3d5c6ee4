// Binned surface-area-heuristic build of the traversal hierarchy on the GPU: the algorithm of host_bvh.cpp: SahBuilder
// (32 bins per axis over the centre bounds of a span, cost = area x count, split below a bin boundary, single-primitive
// leaves, depth bounded for the traversal stack), level by level over ALL spans at once.
//
// Spans are positional, as in bvh_order.hip: position i belongs to span [start[i], end[i]); a level never moves a
// primitive out of its span.  The interior node of span [s, e) has the index the host builder gives it, s + (number of
// left turns on the way down), so the tree is the host builder's tree, node for node, whenever the host's fallback for
// spans that cannot be split by bins (coincident centres, no depth to spare) is not taken for more than two primitives;
// there the halves are cut by position, which is as valid a hierarchy.
//
// Per level: centre bounds of every span; bins -- spans of up to 64 primitives by one thread each, longer spans by all
// their primitives with atomics on order-preserving 64-bit encodings, first in an LDS copy of the table of the span the
// workgroup starts in; the best split per span; a left/right flag per primitive; one exclusive scan over all flags,
// which turns the flags into a stable partition inside every span without sorting; new span bounds and the node.
// The binary tree (children, ranges, primitive of every position) then takes the same path as the Morton-order tree:
// segment-tree boxes, 4-wide collapse, quantisation (bvh_build.hip).  HBM-bound integer/byte work.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>   // rocprim/iterator/texture_cache_iterator.hpp calls memset without including it

#include <rocprim/rocprim.hpp>

#include "bvh_build.h"

namespace rt {

namespace {

constexpr int kB = 256;
constexpr int kBins = 32;        // host_bvh.cpp: SahBuilder::kBins
constexpr int kShortSpan = 64;   // spans up to this size are binned by one thread
#ifndef RT_SAH_TINY
#define RT_SAH_TINY 12
#endif
constexpr int kTinySpan = RT_SAH_TINY;    // spans up to this size try every primitive's bin directly (no bin arrays)
constexpr int kTableWords = 3 * kBins * 3;   // per table: min or max of 3 axes x 32 bins x 3 coordinates

typedef unsigned long long u64;

__device__ __forceinline__ u64 encode(double d) {   // order-preserving: a < b  <=>  encode(a) < encode(b)
    const u64 b = (u64)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double decode(u64 e) {
    const u64 b = (e >> 63) ? (e & 0x7fffffffffffffffull) : ~e;
    return __longlong_as_double((long long)b);
}
__device__ __forceinline__ int ceil_log2(unsigned n) {
    int l = 0;
    while ((1u << l) < n) ++l;
    return l;
}
__device__ __forceinline__ double centre_of(const double* b, int a) { return (b[a] + b[3 + a]) * 0.5; }
__device__ __forceinline__ int bin_of(double c, double lo, double scale) {
    return min(kBins - 1, max(0, (int)((c - lo) * scale)));
}

struct Spans {
    int* start;
    int* end;
    unsigned char* done;
    int* lturns;   // left turns from the root to the span: node index = start + lturns
    int* dleft;    // interior levels the subtree of the span may still use
};

__global__ void __launch_bounds__(kB) k_sah_init(const int n, const int depth, int* __restrict__ perm, const Spans sp) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    perm[i] = i;
    sp.start[i] = 0;
    sp.end[i] = n;
    sp.done[i] = n <= 1 ? 1 : 0;
    sp.lturns[i] = 0;
    sp.dleft[i] = depth;
}

// Bounds of the box CENTRES of every span, at the span's first position (encoded): lo[3 s + a], hi[3 s + a].
__global__ void __launch_bounds__(kB) k_sah_centre_bounds(const double* __restrict__ boxes, const int* __restrict__ perm, const int n,
                                                          const Spans sp, u64* __restrict__ lo, u64* __restrict__ hi) {
    const int i = blockIdx.x * kB + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int s = -1, e = -1;
    bool active = false;
    if (i < n) {
        s = sp.start[i];
        e = sp.end[i];
        active = !sp.done[i];
    }
    const bool longspan = active && (e - s) > kShortSpan;
    if (active && !longspan && i == s) {   // the first element of a short span reduces it
        double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = s; k < e; ++k) {
            const double* b = boxes + 6 * (size_t)perm[k];
            for (int a = 0; a < 3; ++a) {
                const double c = centre_of(b, a);
                mn[a] = fmin(mn[a], c);
                mx[a] = fmax(mx[a], c);
            }
        }
        for (int a = 0; a < 3; ++a) {
            lo[3 * (size_t)s + a] = encode(mn[a]);
            hi[3 * (size_t)s + a] = encode(mx[a]);
        }
    }
    // Every lane of the wave takes part in what follows (lanes outside long spans with the identity), because the
    // shuffles below read all 64 lanes: a lane that had left the kernel would hand over whatever its registers hold.
    const unsigned long long m = __ballot(longspan);
    if (m == 0ull) return;
    u64 mn[3] = {~0ull, ~0ull, ~0ull}, mx[3] = {0ull, 0ull, 0ull};
    if (longspan) {
        const double* b = boxes + 6 * (size_t)perm[i];
        for (int a = 0; a < 3; ++a) mn[a] = mx[a] = encode(centre_of(b, a));
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

// host_bvh.cpp: SahBuilder::build, the sweep over one axis: bb(b) -> box of bin b as 6 doubles, cnt(b) -> count.
template <class BoxAt, class CountAt>
__device__ __forceinline__ void sweep_axis(const int axis, BoxAt&& bb, CountAt&& cnt, const int depth_left, double& best_cost, int& best_axis,
                                           int& best_bin) {
    double right_area[kBins];
    unsigned right_cnt[kBins];
    double acc[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    unsigned c = 0;
    auto join = [&](int b) {
        double v[6];
        bb(b, v);
        for (int k = 0; k < 3; ++k) {
            acc[k] = fmin(acc[k], v[k]);
            acc[3 + k] = fmax(acc[3 + k], v[3 + k]);
        }
    };
    auto half_area = [&]() {
        const double ex = acc[3] - acc[0], ey = acc[4] - acc[1], ez = acc[5] - acc[2];
        return ex * ey + ey * ez + ez * ex;
    };
    for (int b = kBins - 1; b > 0; --b) {
        const unsigned k = cnt(b);
        if (k) join(b);
        c += k;
        right_area[b] = c ? half_area() : 0.0;
        right_cnt[b] = c;
    }
    for (int k = 0; k < 3; ++k) { acc[k] = INFINITY; acc[3 + k] = -INFINITY; }
    c = 0;
    for (int b = 0; b + 1 < kBins; ++b) {   // split between bin b and b + 1
        const unsigned k = cnt(b);
        if (k) join(b);
        c += k;
        const unsigned nl = c, nr = right_cnt[b + 1];
        if (nl == 0 || nr == 0) continue;
        if (ceil_log2(max(nl, nr)) > depth_left - 1) continue;   // keep the stack bound
        const double cost = half_area() * (double)nl + right_area[b + 1] * (double)nr;
        if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
    }
}

// Best split of one axis per span, at [3 * first position + axis]: cost (+inf = none) and the bin below whose upper
// boundary the span is cut.  k_sah_pick then takes the cheapest axis, the lowest on equal cost, as the host's loop over
// the axes does.  Spans of 3 .. kTinySpan shapes: one thread per (span, axis).
__global__ void __launch_bounds__(kB) k_sah_choose_short(const double* __restrict__ boxes, const int* __restrict__ perm, const int n,
                                                         const Spans sp, const u64* __restrict__ lo, const u64* __restrict__ hi,
                                                         double* __restrict__ axis_cost, int* __restrict__ axis_bin) {
    const int j = blockIdx.x * kB + threadIdx.x;
    const int i = j / 3, axis = j - 3 * i;
    if (i >= n || sp.done[i]) return;
    const int s = sp.start[i], e = sp.end[i], m = e - s;
    if (i != s || m > kTinySpan || m == 2) return;
    double best_cost = INFINITY;
    int best_bin = -1;
    const double l = decode(lo[3 * (size_t)s + axis]), ext = decode(hi[3 * (size_t)s + axis]) - l;
    if (ext > 0.0 && isfinite(ext) && m <= kTinySpan) {
        // A handful of primitives occupy a handful of bins: instead of filling and sweeping 32 bins in scratch memory,
        // every primitive's bin is tried as the split (left = bins up to it) with the two boxes joined from scratch.
        // The sweep keeps the first of equally cheap bins; taking the smallest (cost, bin) is the same choice, and
        // the cost is the sweep's, bit for bit (the same boxes, the same arithmetic).
        const double scale = kBins / ext;
        const int depth_left = sp.dleft[i];
        for (int c = s; c < e; ++c) {
            const int B = bin_of(centre_of(boxes + 6 * (size_t)perm[c], axis), l, scale);
            double lb[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
            double rb[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
            unsigned nl = 0, nr = 0;
            for (int k = s; k < e; ++k) {
                const double* pb = boxes + 6 * (size_t)perm[k];
                const bool left = bin_of(centre_of(pb, axis), l, scale) <= B;
                double* dst = left ? lb : rb;
                for (int q = 0; q < 3; ++q) {
                    dst[q] = fmin(dst[q], pb[q]);
                    dst[3 + q] = fmax(dst[3 + q], pb[3 + q]);
                }
                if (left) ++nl; else ++nr;
            }
            if (nl == 0 || nr == 0) continue;
            if (ceil_log2(max(nl, nr)) > depth_left - 1) continue;
            const double lx = lb[3] - lb[0], ly = lb[4] - lb[1], lz = lb[5] - lb[2];
            const double rx = rb[3] - rb[0], ry = rb[4] - rb[1], rz = rb[5] - rb[2];
            const double cost = (lx * ly + ly * lz + lz * lx) * (double)nl + (rx * ry + ry * rz + rz * rx) * (double)nr;
            if (cost < best_cost || (cost == best_cost && B < best_bin)) { best_cost = cost; best_bin = B; }
        }
    }
    axis_cost[3 * (size_t)s + axis] = best_cost;
    axis_bin[3 * (size_t)s + axis] = best_bin;
}

// Spans of kTinySpan + 1 .. 64 shapes: one WAVE per span (the wave that holds the span's first position takes its spans
// in turn).  Lane k bins shape k into a table in LDS (64-bit atomics on the order-preserving encodings), lane b then
// evaluates the split below bin b from the table -- the boxes left and right of it are the unions the host's sweep
// accumulates, so the cost is the sweep's bit for bit -- and the cheapest (cost, bin) of the lanes is the sweep's choice.
__global__ void __launch_bounds__(kB) k_sah_choose_mid(const double* __restrict__ boxes, const int* __restrict__ perm, const int n,
                                                       const Spans sp, const u64* __restrict__ lo, const u64* __restrict__ hi,
                                                       double* __restrict__ axis_cost, int* __restrict__ axis_bin) {
    __shared__ u64 s_min[kB / 64][kBins][3], s_max[kB / 64][kBins][3];
    __shared__ unsigned s_cnt[kB / 64][kBins];
    const int i = blockIdx.x * kB + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    bool leader = false;
    if (i < n && !sp.done[i]) {
        const int m = sp.end[i] - sp.start[i];
        leader = i == sp.start[i] && m > kTinySpan && m <= kShortSpan;
    }
    unsigned long long todo = __ballot(leader);
    const int base = i - lane;
    while (todo != 0ull) {   // wave-uniform
        const int L = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const int s = base + L;
        const int e = sp.end[s], m = e - s, depth_left = sp.dleft[s];
        double pb[6] = {0, 0, 0, 0, 0, 0};
        if (lane < m) {
            const double* src = boxes + 6 * (size_t)perm[s + lane];
            for (int k = 0; k < 6; ++k) pb[k] = src[k];
        }
        for (int axis = 0; axis < 3; ++axis) {
            const double l = decode(lo[3 * (size_t)s + axis]), ext = decode(hi[3 * (size_t)s + axis]) - l;
            double cost = INFINITY;
            int bin = -1;
            if (ext > 0.0 && isfinite(ext)) {
                if (lane < kBins) {
                    for (int k = 0; k < 3; ++k) { s_min[w][lane][k] = ~0ull; s_max[w][lane][k] = 0ull; }
                    s_cnt[w][lane] = 0u;
                }
                __threadfence_block();
                if (lane < m) {
                    const int b = bin_of((pb[axis] + pb[3 + axis]) * 0.5, l, kBins / ext);
                    for (int k = 0; k < 3; ++k) {
                        atomicMin(&s_min[w][b][k], encode(pb[k]));
                        atomicMax(&s_max[w][b][k], encode(pb[3 + k]));
                    }
                    atomicAdd(&s_cnt[w][b], 1u);
                }
                __threadfence_block();
                if (lane + 1 < kBins) {   // split between bin `lane` and the next
                    double lb[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    double rb[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    unsigned nl = 0, nr = 0;
                    for (int b = 0; b < kBins; ++b) {
                        const unsigned c = s_cnt[w][b];
                        if (c == 0u) continue;
                        double* dst = b <= lane ? lb : rb;
                        for (int k = 0; k < 3; ++k) {
                            dst[k] = fmin(dst[k], decode(s_min[w][b][k]));
                            dst[3 + k] = fmax(dst[3 + k], decode(s_max[w][b][k]));
                        }
                        if (b <= lane) nl += c; else nr += c;
                    }
                    if (nl != 0u && nr != 0u && !(ceil_log2(max(nl, nr)) > depth_left - 1)) {
                        const double lx = lb[3] - lb[0], ly = lb[4] - lb[1], lz = lb[5] - lb[2];
                        const double rx = rb[3] - rb[0], ry = rb[4] - rb[1], rz = rb[5] - rb[2];
                        cost = (lx * ly + ly * lz + lz * lx) * (double)nl + (rx * ry + ry * rz + rz * rx) * (double)nr;
                        bin = lane;
                    }
                }
                __threadfence_block();   // the table is read before the next axis clears it
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {   // the cheapest split; the lowest bin among equally cheap ones
                    const double oc = __shfl_xor(cost, off);
                    const int ob = __shfl_xor(bin, off);
                    if (ob >= 0 && (bin < 0 || oc < cost || (oc == cost && ob < bin))) { cost = oc; bin = ob; }
                }
            }
            if (lane == 0) {
                axis_cost[3 * (size_t)s + axis] = bin >= 0 ? cost : INFINITY;
                axis_bin[3 * (size_t)s + axis] = bin;
            }
        }
    }
}

// Decision per span, at its first position: axis >= 0 and bin = split below that bin boundary; axis -1 = cut the span in
// halves by position; axis -2 = a pair.
__global__ void __launch_bounds__(kB) k_sah_pick(const int n, const Spans sp, const double* __restrict__ axis_cost, const int* __restrict__ axis_bin,
                                                 int* __restrict__ dec_axis, int* __restrict__ dec_bin) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n || sp.done[i] || i != sp.start[i]) return;
    if (sp.end[i] - i == 2) { dec_axis[i] = -2; dec_bin[i] = 0; return; }
    double best_cost = INFINITY;
    int best_axis = -1, best_bin = -1;
    for (int axis = 0; axis < 3; ++axis) {
        const double c = axis_cost[3 * (size_t)i + axis];
        if (c < best_cost) { best_cost = c; best_axis = axis; best_bin = axis_bin[3 * (size_t)i + axis]; }
    }
    dec_axis[i] = best_axis;
    dec_bin[i] = best_bin;
}

// Tables of the long spans: table t = start >> 6 (long spans start more than 64 positions apart).
// tmin / tmax: [t][axis][bin][coordinate] encoded, tcnt: [t][axis][bin].
__global__ void __launch_bounds__(kB) k_sah_bin_long(const double* __restrict__ boxes, const int* __restrict__ perm, const int n, const Spans sp,
                                                     const u64* __restrict__ lo, const u64* __restrict__ hi, u64* __restrict__ tmin,
                                                     u64* __restrict__ tmax, unsigned* __restrict__ tcnt) {
    __shared__ u64 s_min[kTableWords], s_max[kTableWords];
    __shared__ unsigned s_cnt[3 * kBins];
    __shared__ int s_span;
    const int i = blockIdx.x * kB + threadIdx.x;
    for (int k = threadIdx.x; k < kTableWords; k += kB) { s_min[k] = ~0ull; s_max[k] = 0ull; }
    for (int k = threadIdx.x; k < 3 * kBins; k += kB) s_cnt[k] = 0u;
    int s = -1;
    bool longspan = false;
    if (i < n && !sp.done[i]) {
        s = sp.start[i];
        longspan = sp.end[i] - s > kShortSpan;
    }
    if (threadIdx.x == 0) s_span = longspan ? s : -1;   // the span the workgroup starts in gets the LDS copy
    __syncthreads();
    const int local = s_span;
    if (longspan) {
        const double* pb = boxes + 6 * (size_t)perm[i];
        const size_t t = (size_t)(s >> 6);
        for (int axis = 0; axis < 3; ++axis) {
            const double l = decode(lo[3 * (size_t)s + axis]), ext = decode(hi[3 * (size_t)s + axis]) - l;
            if (!(ext > 0.0) || !isfinite(ext)) continue;
            const int b = bin_of(centre_of(pb, axis), l, kBins / ext);
            const int w = (axis * kBins + b) * 3;
            if (s == local) {
                for (int c = 0; c < 3; ++c) {
                    atomicMin(&s_min[w + c], encode(pb[c]));
                    atomicMax(&s_max[w + c], encode(pb[3 + c]));
                }
                atomicAdd(&s_cnt[axis * kBins + b], 1u);
            } else {
                for (int c = 0; c < 3; ++c) {
                    atomicMin(&tmin[t * kTableWords + w + c], encode(pb[c]));
                    atomicMax(&tmax[t * kTableWords + w + c], encode(pb[3 + c]));
                }
                atomicAdd(&tcnt[t * 3 * kBins + axis * kBins + b], 1u);
            }
        }
    }
    __syncthreads();
    if (local < 0) return;
    const size_t t = (size_t)(local >> 6);
    for (int k = threadIdx.x; k < 3 * kBins; k += kB) {
        const unsigned c = s_cnt[k];
        if (c == 0u) continue;
        atomicAdd(&tcnt[t * 3 * kBins + k], c);
        for (int q = 0; q < 3; ++q) {
            atomicMin(&tmin[t * kTableWords + k * 3 + q], s_min[k * 3 + q]);
            atomicMax(&tmax[t * kTableWords + k * 3 + q], s_max[k * 3 + q]);
        }
    }
}

__global__ void __launch_bounds__(kB) k_sah_choose_long(const int n, const Spans sp, const u64* __restrict__ lo, const u64* __restrict__ hi,
                                                        const u64* __restrict__ tmin, const u64* __restrict__ tmax,
                                                        const unsigned* __restrict__ tcnt, double* __restrict__ axis_cost, int* __restrict__ axis_bin) {
    const int j = blockIdx.x * kB + threadIdx.x;
    const int i = j / 3, axis = j - 3 * i;
    if (i >= n || sp.done[i]) return;
    const int s = sp.start[i], m = sp.end[i] - s;
    if (i != s || m <= kShortSpan) return;
    const size_t t = (size_t)(s >> 6);
    double best_cost = INFINITY;
    int best_axis = -1, best_bin = -1;
    const double ext = decode(hi[3 * (size_t)s + axis]) - decode(lo[3 * (size_t)s + axis]);
    if (ext > 0.0 && isfinite(ext))
        sweep_axis(axis,
                   [&](int b, double* v) {
                       const size_t w = t * kTableWords + (size_t)(axis * kBins + b) * 3;
                       for (int k = 0; k < 3; ++k) { v[k] = decode(tmin[w + k]); v[3 + k] = decode(tmax[w + k]); }
                   },
                   [&](int b) { return tcnt[t * 3 * kBins + axis * kBins + b]; }, sp.dleft[i], best_cost, best_axis, best_bin);
    axis_cost[3 * (size_t)s + axis] = best_cost;
    axis_bin[3 * (size_t)s + axis] = best_bin;
}

// left[i] = 1 when the primitive at position i goes to the left half of its span (0 for final positions).
__global__ void __launch_bounds__(kB) k_sah_side(const double* __restrict__ boxes, int* __restrict__ perm, const int n, const Spans sp,
                                                 const u64* __restrict__ lo, const u64* __restrict__ hi, const int* __restrict__ dec_axis,
                                                 const int* __restrict__ dec_bin, int* __restrict__ left) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    if (sp.done[i]) { left[i] = 0; return; }
    const int s = sp.start[i], e = sp.end[i], m = e - s;
    const int axis = dec_axis[s];
    if (axis >= 0) {
        const double l = decode(lo[3 * (size_t)s + axis]), ext = decode(hi[3 * (size_t)s + axis]) - l;
        const int b = bin_of(centre_of(boxes + 6 * (size_t)perm[i], axis), l, kBins / ext);
        left[i] = b <= dec_bin[s] ? 1 : 0;
    } else if (axis == -2) {
        // a pair: the smaller (centre on the widest axis of the centre bounds, primitive index) first, as the host's
        // balanced fallback orders it (strict comparisons, x on ties)
        if (i == s) {
            double ext[3];
            for (int a = 0; a < 3; ++a) ext[a] = decode(hi[3 * (size_t)s + a]) - decode(lo[3 * (size_t)s + a]);
            int ax = 0;
            if (ext[1] > ext[0] && ext[1] > ext[2]) ax = 1;
            else if (ext[2] > ext[0] && ext[2] > ext[1]) ax = 2;
            const int p = perm[s], q = perm[s + 1];
            const double cp = centre_of(boxes + 6 * (size_t)p, ax), cq = centre_of(boxes + 6 * (size_t)q, ax);
            const bool p_first = cp < cq || (cp == cq && p < q);
            if (!p_first) { perm[s] = q; perm[s + 1] = p; }
        }
        left[i] = i == s ? 1 : 0;
    } else {
        left[i] = i < s + m / 2 ? 1 : 0;
    }
}

__global__ void __launch_bounds__(kB) k_sah_scatter(const int* __restrict__ perm_in, int* __restrict__ perm_out, const int n, const Spans sp,
                                                    const int* __restrict__ left, const int* __restrict__ rank, int* __restrict__ nleft) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    if (sp.done[i]) { perm_out[i] = perm_in[i]; return; }
    const int s = sp.start[i], e = sp.end[i];
    const int before = rank[i] - rank[s];                                           // left-going primitives of the span before i
    const int nl = (e < n ? rank[e] : rank[n - 1] + left[n - 1]) - rank[s];
    const int at = left[i] ? s + before : s + nl + (i - s - before);
    perm_out[at] = perm_in[i];
    if (i == s) nleft[s] = nl;
}

__global__ void __launch_bounds__(kB) k_sah_split(const int n, const Spans sp, const int* __restrict__ nleft, int* __restrict__ child_l,
                                                  int* __restrict__ child_r, int* __restrict__ range_lo, int* __restrict__ range_hi,
                                                  unsigned* __restrict__ flags) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n || sp.done[i]) return;
    const int s = sp.start[i], e = sp.end[i];
    const int nl = nleft[s];
    if (nl <= 0 || nl >= e - s) { flags[1] = 1u; return; }   // cannot happen: every decision leaves both halves non-empty
    const int mid = s + nl, turns = sp.lturns[i], me = s + turns;
    if (i == s) {
        if (me < 0 || me >= n - 1) { flags[1] = 1u; return; }
        child_l[me] = nl == 1 ? ~s : me + 1;
        child_r[me] = e - mid == 1 ? ~mid : mid + turns;
        range_lo[me] = s;
        range_hi[me] = e - 1;
    }
    int size;
    if (i < mid) {
        sp.end[i] = mid;
        sp.lturns[i] = turns + 1;
        size = nl;
    } else {
        sp.start[i] = mid;
        size = e - mid;
    }
    sp.dleft[i] -= 1;
    if (size == 1) sp.done[i] = 1;
    else flags[0] = 1u;   // some span is still to be split
}

__global__ void __launch_bounds__(kB) k_sah_finish(const int n, const int* __restrict__ perm, const unsigned char* __restrict__ done,
                                                   unsigned* __restrict__ vals, unsigned* __restrict__ flags) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    vals[i] = (unsigned)perm[i];
    if (!done[i]) flags[2] = 1u;   // levels ran out before every span was final
}

size_t align_up(size_t v) { return (v + 255) & ~size_t(255); }

}  // namespace

hipError_t sah_binary_tree_device(const double* d_boxes, int n, int max_depth, hipStream_t stream, unsigned* d_vals, int* d_left, int* d_right,
                                  int* d_range_lo, int* d_range_hi, bool* incomplete) {
    *incomplete = false;
    if (n < 2) return hipErrorInvalidValue;
    const size_t N = static_cast<size_t>(n);
    int log2n = 0;
    while ((size_t(1) << log2n) < N) ++log2n;
    const int depth = std::max(max_depth, log2n);   // host_bvh.cpp: build_traversal_bvh
    const int levels = depth + 1;
    const size_t tables = (N >> 6) + 1;

    size_t scan_bytes = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, scan_bytes, (int*)nullptr, (int*)nullptr, 0, N, rocprim::plus<int>(), stream);
    if (e != hipSuccess) return e;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t at = off; off += align_up(bytes); return at; };
    const size_t o_perm = carve(N * sizeof(int)), o_perm2 = carve(N * sizeof(int));
    const size_t o_start = carve(N * sizeof(int)), o_end = carve(N * sizeof(int)), o_done = carve(N);
    const size_t o_lturns = carve(N * sizeof(int)), o_dleft = carve(N * sizeof(int));
    const size_t o_lo = carve(N * 3 * sizeof(u64)), o_hi = carve(N * 3 * sizeof(u64));
    const size_t o_axis = carve(N * sizeof(int)), o_bin = carve(N * sizeof(int)), o_nleft = carve(N * sizeof(int));
    const size_t o_acost = carve(N * 3 * sizeof(double)), o_abin = carve(N * 3 * sizeof(int));
    const size_t o_side = carve(N * sizeof(int)), o_rank = carve(N * sizeof(int));
    const size_t o_tmin = carve(tables * kTableWords * sizeof(u64)), o_tmax = carve(tables * kTableWords * sizeof(u64));
    const size_t o_tcnt = carve(tables * 3 * kBins * sizeof(unsigned));
    const size_t o_flags = carve(4 * sizeof(unsigned));
    const size_t o_temp = carve(scan_bytes);
    char* base = nullptr;
    if ((e = hipMalloc(reinterpret_cast<void**>(&base), off)) != hipSuccess) return e;
    auto fail = [&](hipError_t code) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(base);
        return code;
    };
#define SAH_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(e_); } while (0)
    int* d_perm = reinterpret_cast<int*>(base + o_perm);
    int* d_perm2 = reinterpret_cast<int*>(base + o_perm2);
    Spans sp{reinterpret_cast<int*>(base + o_start), reinterpret_cast<int*>(base + o_end), reinterpret_cast<unsigned char*>(base + o_done),
             reinterpret_cast<int*>(base + o_lturns), reinterpret_cast<int*>(base + o_dleft)};
    u64* d_lo = reinterpret_cast<u64*>(base + o_lo);
    u64* d_hi = reinterpret_cast<u64*>(base + o_hi);
    int* d_axis = reinterpret_cast<int*>(base + o_axis);
    int* d_bin = reinterpret_cast<int*>(base + o_bin);
    int* d_nleft = reinterpret_cast<int*>(base + o_nleft);
    double* d_acost = reinterpret_cast<double*>(base + o_acost);
    int* d_abin = reinterpret_cast<int*>(base + o_abin);
    int* d_side = reinterpret_cast<int*>(base + o_side);
    int* d_rank = reinterpret_cast<int*>(base + o_rank);
    u64* d_tmin = reinterpret_cast<u64*>(base + o_tmin);
    u64* d_tmax = reinterpret_cast<u64*>(base + o_tmax);
    unsigned* d_tcnt = reinterpret_cast<unsigned*>(base + o_tcnt);
    unsigned* d_flags = reinterpret_cast<unsigned*>(base + o_flags);
    void* d_temp = base + o_temp;

    const unsigned g = static_cast<unsigned>((N + kB - 1) / kB), g3 = static_cast<unsigned>((3 * N + kB - 1) / kB);
    SAH_TRY(hipMemsetAsync(d_flags, 0, 4 * sizeof(unsigned), stream));
    hipLaunchKernelGGL(k_sah_init, dim3(g), dim3(kB), 0, stream, n, depth, d_perm, sp);
    unsigned flags[4] = {0, 0, 0, 0};
    for (int level = 0; level < levels; ++level) {
        SAH_TRY(hipMemsetAsync(d_lo, 0xff, N * 3 * sizeof(u64), stream));
        SAH_TRY(hipMemsetAsync(d_hi, 0x00, N * 3 * sizeof(u64), stream));
        SAH_TRY(hipMemsetAsync(d_tmin, 0xff, tables * kTableWords * sizeof(u64), stream));
        SAH_TRY(hipMemsetAsync(d_tmax, 0x00, tables * kTableWords * sizeof(u64), stream));
        SAH_TRY(hipMemsetAsync(d_tcnt, 0x00, tables * 3 * kBins * sizeof(unsigned), stream));
        SAH_TRY(hipMemsetAsync(d_flags, 0, sizeof(unsigned), stream));
        hipLaunchKernelGGL(k_sah_centre_bounds, dim3(g), dim3(kB), 0, stream, d_boxes, d_perm, n, sp, d_lo, d_hi);
        hipLaunchKernelGGL(k_sah_choose_short, dim3(g3), dim3(kB), 0, stream, d_boxes, d_perm, n, sp, d_lo, d_hi, d_acost, d_abin);
        hipLaunchKernelGGL(k_sah_choose_mid, dim3(g), dim3(kB), 0, stream, d_boxes, d_perm, n, sp, d_lo, d_hi, d_acost, d_abin);
        hipLaunchKernelGGL(k_sah_bin_long, dim3(g), dim3(kB), 0, stream, d_boxes, d_perm, n, sp, d_lo, d_hi, d_tmin, d_tmax, d_tcnt);
        hipLaunchKernelGGL(k_sah_choose_long, dim3(g3), dim3(kB), 0, stream, n, sp, d_lo, d_hi, d_tmin, d_tmax, d_tcnt, d_acost, d_abin);
        hipLaunchKernelGGL(k_sah_pick, dim3(g), dim3(kB), 0, stream, n, sp, d_acost, d_abin, d_axis, d_bin);
        hipLaunchKernelGGL(k_sah_side, dim3(g), dim3(kB), 0, stream, d_boxes, d_perm, n, sp, d_lo, d_hi, d_axis, d_bin, d_side);
        size_t tb = scan_bytes;
        SAH_TRY(rocprim::exclusive_scan(d_temp, tb, d_side, d_rank, 0, N, rocprim::plus<int>(), stream));
        hipLaunchKernelGGL(k_sah_scatter, dim3(g), dim3(kB), 0, stream, d_perm, d_perm2, n, sp, d_side, d_rank, d_nleft);
        hipLaunchKernelGGL(k_sah_split, dim3(g), dim3(kB), 0, stream, n, sp, d_nleft, d_left, d_right, d_range_lo, d_range_hi, d_flags);
        std::swap(d_perm, d_perm2);
        if ((level & 3) == 3 || level + 1 == levels) {   // every fourth level: is any span still to be split?
            SAH_TRY(hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, stream));
            SAH_TRY(hipStreamSynchronize(stream));
            if (flags[1]) return fail(hipErrorUnknown);
            if (!flags[0]) break;
        }
    }
    hipLaunchKernelGGL(k_sah_finish, dim3(g), dim3(kB), 0, stream, n, d_perm, sp.done, d_vals, d_flags);
    SAH_TRY(hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, stream));
    SAH_TRY(hipStreamSynchronize(stream));
    SAH_TRY(hipGetLastError());
#undef SAH_TRY
    (void)hipFree(base);
    if (flags[1]) return hipErrorUnknown;
    *incomplete = flags[2] != 0u;
    return hipSuccess;
}

}  // namespace rt
