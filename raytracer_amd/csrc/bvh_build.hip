// GPU build of the traversal hierarchy (bvh_build.h).  Pipeline, all on one stream, no host round trip until the end:
//
//   k_lbvh_codes      63-bit Morton code of every primitive box centre on the root box's 2^21 grid
//   rocprim sort      (code, primitive) pairs
//   k_lbvh_hierarchy  binary radix tree over the sorted codes, one thread per interior node (Karras 2012; codes are
//                     made unique by the sorted position, so duplicate centres need no special case)
//   k_lbvh_leaf_boxes, k_lbvh_seg_level x log2(n), k_lbvh_node_boxes
//                     boxes of the interior nodes: a node covers a contiguous range of the sorted primitives, so its
//                     box is a range query on a segment tree over the sorted boxes (built level by level with plain
//                     loads and stores; no atomics, no fences: a bottom-up walk with one agent-scope fence per step
//                     took 6.2 ms for 10^6 shapes on this chip, whose eight L2s are only coherent through memory)
//   k_lbvh_mark x L   collapse to four children per node, one 4-wide level per launch: a node that roots a 4-wide
//                     node opens, twice, its interior child with the largest surface (host_bvh.cpp: Collapser::open)
//   rocprim scan      compact numbering of the 4-wide nodes (in binary-node order, i.e. along the Morton curve)
//   k_lbvh_emit       DNode4: child boxes moved outward by the pad and quantised outward to 8 bits per plane, the same
//                     arithmetic as host_bvh.cpp: Collapser::quantise
//
// The image is conservative by the same argument as the host builder's (every child box contains the padded boxes of
// all primitives below it; tests/test_gpu_lbvh.py checks exactly that on the nodes read back from the device), and the
// closest hit does not depend on the hierarchy, so frames are bit-identical to those of the SAH builder.
// HBM-bound integer/byte work: nothing here is shaped for MFMA.
#include "bvh_build.h"

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>   // rocprim/iterator/texture_cache_iterator.hpp calls memset without including it

#include <rocprim/rocprim.hpp>

#include "host_bvh.h"

namespace rt {

namespace {

constexpr int kB = 256;
constexpr int kMaxWideLevels = kTravStackDepth / 3;   // a 4-wide node defers at most three children per level
constexpr int kMaxSegLevels = 32;

typedef unsigned long long u64;

struct CodeGrid {
    double lo[3], scale[3];   // cell = (centre - lo) * scale, 2^21 cells per axis (scale 0 on a flat axis)
};

__device__ __forceinline__ u64 spread21(unsigned v) {   // bit k of v -> bit 3k
    u64 x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void __launch_bounds__(kB) k_lbvh_codes(const double* __restrict__ boxes, const int n, const CodeGrid g,
                                                   u64* __restrict__ keys, unsigned* __restrict__ vals) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    u64 code = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double c = (boxes[6 * (size_t)i + a] + boxes[6 * (size_t)i + 3 + a]) * 0.5;
        double q = (c - g.lo[a]) * g.scale[a];
        if (!(q >= 0.0)) q = 0.0;              // also NaN
        if (q > 2097151.0) q = 2097151.0;
        code |= spread21((unsigned)q) << (2 - a);
    }
    keys[i] = code;
    vals[i] = (unsigned)i;
}

// Length of the common prefix of the (code, position) keys i and j; -1 outside the array.
__device__ __forceinline__ int prefix(const u64* __restrict__ keys, const int n, const int i, const int j) {
    if (j < 0 || j >= n) return -1;
    const u64 a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(i ^ j);
}

// References inside the binary tree: >= 0 interior node, < 0 leaf = ~(sorted position).
__global__ void __launch_bounds__(kB) k_lbvh_hierarchy(const u64* __restrict__ keys, const int n, int* __restrict__ left,
                                                       int* __restrict__ right, int* __restrict__ range_lo,
                                                       int* __restrict__ range_hi) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (prefix(keys, n, i, i + 1) - prefix(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = prefix(keys, n, i, i - d);
    int lmax = 2;
    while (lmax < 2 * n && prefix(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = prefix(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = min(i, j), hi = max(i, j);
    // gamma in [lo, hi - 1] by construction; clamp so that a wrong step could never index outside the arrays
    const int gm = min(max(gamma, lo), hi - 1);
    left[i] = (gm == lo) ? ~gm : gm;
    right[i] = (gm + 1 == hi) ? ~(gm + 1) : gm + 1;
    range_lo[i] = lo;   // the sorted primitives this node covers
    range_hi[i] = hi;
}

// Segment tree over the sorted primitive boxes, one array per level: level 0 = the boxes themselves, entry j of level
// k = the union of entries 2j and 2j + 1 of level k - 1 (the last entry of an odd level stands alone).
struct SegLevels {
    unsigned long long offset[kMaxSegLevels];   // in boxes, from the start of the tree's memory
    int count[kMaxSegLevels];
    int levels;
};

__global__ void __launch_bounds__(kB) k_lbvh_leaf_boxes(const double* __restrict__ boxes, const unsigned* __restrict__ vals, const int n,
                                                        double* __restrict__ seg, unsigned* __restrict__ flags) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    const unsigned p = vals[i];
    if (p >= (unsigned)n) { flags[1] = 1u; return; }
#pragma unroll
    for (int k = 0; k < 6; ++k) seg[6 * (size_t)i + k] = boxes[6 * (size_t)p + k];
}

__global__ void __launch_bounds__(kB) k_lbvh_seg_level(const double* __restrict__ prev, const int prev_count, double* __restrict__ cur,
                                                       const int cur_count) {
    const int j = blockIdx.x * kB + threadIdx.x;
    if (j >= cur_count) return;
    double b[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) b[k] = prev[6 * (size_t)(2 * j) + k];
    if (2 * j + 1 < prev_count) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            b[k] = fmin(b[k], prev[6 * (size_t)(2 * j + 1) + k]);
            b[3 + k] = fmax(b[3 + k], prev[6 * (size_t)(2 * j + 1) + 3 + k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) cur[6 * (size_t)j + k] = b[k];
}

// Box of every interior node = union over its range [lo, hi] of sorted primitives: the usual bottom-up range query, at
// most two entries per level.  Most nodes cover a handful of primitives and read two or three entries.
__global__ void __launch_bounds__(kB) k_lbvh_node_boxes(const int n, const int* __restrict__ range_lo, const int* __restrict__ range_hi,
                                                        const double* __restrict__ seg, const SegLevels L, double* __restrict__ nbox,
                                                        unsigned* __restrict__ flags) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n - 1) return;
    int l = range_lo[i], r = range_hi[i] + 1;
    if (l < 0 || r > n || l >= r) { flags[1] = 1u; return; }
    double b[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    auto add = [&](int level, int j) {
        const double* e = seg + 6 * (L.offset[level] + (unsigned long long)j);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            b[k] = fmin(b[k], e[k]);
            b[3 + k] = fmax(b[3 + k], e[3 + k]);
        }
    };
    for (int level = 0; level < L.levels && l < r; ++level) {
        if (l & 1) add(level, l++);
        if (r & 1) add(level, --r);
        l >>= 1;
        r >>= 1;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) nbox[6 * (size_t)i + k] = b[k];
}

__device__ __forceinline__ double half_area6(const double* b) {
    const double ex = b[3] - b[0], ey = b[4] - b[1], ez = b[5] - b[2];
    return ex * ey + ey * ez + ez * ex;
}

// Children of the 4-wide node rooted at binary node b (host_bvh.cpp: Collapser::open): refs[0..count).
__device__ __forceinline__ int open_slots(const int b, const int* __restrict__ left, const int* __restrict__ right,
                                          const double* __restrict__ nbox, int refs[4]) {
    refs[0] = left[b];
    refs[1] = right[b];
    refs[2] = refs[3] = 0;
    int count = 2;
    while (count < 4) {
        int pick = -1;
        double best = -1.0;
        for (int i = 0; i < count; ++i) {
            if (refs[i] < 0) continue;
            const double a = half_area6(&nbox[6 * (size_t)refs[i]]);
            if (!(a <= best)) { best = a; pick = i; }
        }
        if (pick < 0) break;
        const int r = refs[pick];
        refs[pick] = left[r];
        refs[count++] = right[r];
    }
    return count;
}

// level[b] = 4-wide level (1 = root) of the 4-wide node rooted at binary node b, 0 = b is inside some 4-wide node.
__global__ void __launch_bounds__(kB) k_lbvh_mark(const int n, const int lvl, const int max_levels, const int* __restrict__ left, const int* __restrict__ right,
                                                  const double* __restrict__ nbox, int* level, int4* __restrict__ slots,
                                                  unsigned* __restrict__ flags) {
    const int b = blockIdx.x * kB + threadIdx.x;
    if (b >= n - 1 || level[b] != lvl) return;
    int refs[4];
    const int count = open_slots(b, left, right, nbox, refs);
    for (int i = count; i < 4; ++i) refs[i] = kEmptyChild;
    slots[b] = make_int4(refs[0], refs[1], refs[2], refs[3]);
    bool interior = false;
    for (int i = 0; i < count; ++i)
        if (refs[i] >= 0) {
            interior = true;
            if (lvl < max_levels) level[refs[i]] = lvl + 1;
        }
    if (interior && lvl >= max_levels) flags[0] = 1u;   // deeper than the traversal stack allows
    flags[2] = (unsigned)lvl;   // every writer of a launch stores the same value
}

__global__ void __launch_bounds__(kB) k_lbvh_used(const int n, const int* __restrict__ level, int* __restrict__ used) {
    const int b = blockIdx.x * kB + threadIdx.x;
    if (b < n - 1) used[b] = level[b] > 0 ? 1 : 0;
}

__global__ void __launch_bounds__(kB) k_lbvh_emit(const int n, const double pad, const int* __restrict__ level, const int* __restrict__ widx,
                                                  const int4* __restrict__ slots, const double* __restrict__ nbox,
                                                  const double* __restrict__ boxes, const unsigned* __restrict__ vals,
                                                  DNode4* __restrict__ nodes, const int max_nodes, unsigned* __restrict__ flags) {
    const int b = blockIdx.x * kB + threadIdx.x;
    if (b >= n - 1 || level[b] <= 0) return;
    const int me = widx[b];
    if (me < 0 || me >= max_nodes) { flags[1] = 1u; return; }
    const int4 s4 = slots[b];
    const int refs[4] = {s4.x, s4.y, s4.z, s4.w};
    double lo[4][3], hi[4][3];
    int child[4];
    int count = 0;
    for (int c = 0; c < 4; ++c) {
        child[c] = kEmptyChild;
        if (refs[c] == kEmptyChild) continue;
        const double* src;
        if (refs[c] >= 0) {
            src = &nbox[6 * (size_t)refs[c]];
            child[c] = widx[refs[c]];
        } else {
            const unsigned p = vals[~refs[c]];
            src = &boxes[6 * (size_t)p];
            child[c] = make_leaf_ref((int)p, 1);
        }
        for (int a = 0; a < 3; ++a) { lo[c][a] = src[a]; hi[c][a] = src[3 + a]; }
        count = c + 1;   // used slots are the first ones
    }
    DNode4 o;
    for (int c = 0; c < 4; ++c) o.child[c] = child[c];
    // host_bvh.cpp: Collapser::quantise, operation by operation
    for (int a = 0; a < 3; ++a) {
        double l = INFINITY, h = -INFINITY;
        for (int c = 0; c < count; ++c) {
            l = fmin(l, lo[c][a] - pad);
            h = fmax(h, hi[c][a] + pad);
        }
        if (!isfinite(l)) l = -(double)FLT_MAX;
        if (!isfinite(h)) h = (double)FLT_MAX;
        const float org = __double2float_rd(l);
        int e = -126;
        const double ext = h - (double)org;
        if (ext > 0.0) {
            int ex;
            (void)frexp(ext / 255.0, &ex);
            e = max(-126, min(127, ex));
        }
        while (e < 127 && ceil((h - (double)org) / ldexp(1.0, e)) > 255.0) ++e;
        const double step = ldexp(1.0, e);
        unsigned wlo = 0xffffffffu, whi = 0u;   // unused slots: lower plane above upper plane
        for (int c = 0; c < count; ++c) {
            double ql = floor((fmax(lo[c][a] - pad, (double)org) - (double)org) / step);
            double qh = ceil((fmin(hi[c][a] + pad, h) - (double)org) / step);
            if (!(ql >= 0.0)) ql = 0.0;
            if (!(qh <= 255.0)) qh = 255.0;
            if (ql > 255.0) ql = 255.0;
            if (qh < 0.0) qh = 0.0;
            wlo = (wlo & ~(0xffu << (8 * c))) | ((unsigned)ql << (8 * c));
            whi = (whi & ~(0xffu << (8 * c))) | ((unsigned)qh << (8 * c));
        }
        o.org[a] = org;
        (a == 0 ? o.step_x : o.step_yz[a - 1]) = (float)step;
        o.qlo[a] = wlo;
        o.qhi[a] = whi;
    }
    nodes[me] = o;
}

size_t align_up(size_t v) { return (v + 255) & ~size_t(255); }

}  // namespace

LbvhStatus build_lbvh_device(const std::vector<Box>& boxes, const Box& root_box, double max_abs_origin, hipStream_t stream,
                             DeviceBvh4* out, hipError_t* err, bool binned_sah) {
    *err = hipSuccess;
    const size_t n = boxes.size();
    if (n < 2 || n > (size_t(1) << 30)) { *err = hipErrorInvalidValue; return LbvhStatus::device_error; }
    const int ni = static_cast<int>(n);
    static_assert(sizeof(Box) == 6 * sizeof(double), "Box is six doubles");

    CodeGrid grid;
    for (int a = 0; a < 3; ++a) {
        const double ext = root_box.hi[a] - root_box.lo[a];
        grid.lo[a] = std::isfinite(root_box.lo[a]) ? root_box.lo[a] : 0.0;
        grid.scale[a] = (ext > 0.0 && std::isfinite(ext)) ? 2097152.0 / ext : 0.0;
    }

    // one allocation for everything temporary
    u64 *keys_in = nullptr, *keys_out = nullptr;
    size_t sort_bytes = 0, scan_bytes = 0;
    hipError_t e;
    e = rocprim::radix_sort_pairs(nullptr, sort_bytes, keys_in, keys_out, (unsigned*)nullptr, (unsigned*)nullptr, n, 0, 63, stream);
    if (e != hipSuccess) { *err = e; return LbvhStatus::device_error; }
    e = rocprim::exclusive_scan(nullptr, scan_bytes, (int*)nullptr, (int*)nullptr, 0, n - 1, rocprim::plus<int>(), stream);
    if (e != hipSuccess) { *err = e; return LbvhStatus::device_error; }
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t at = off; off += align_up(bytes); return at; };
    const size_t o_boxes = carve(n * 6 * sizeof(double)), o_nbox = carve((n - 1) * 6 * sizeof(double));
    const size_t o_keys_in = carve(n * sizeof(u64)), o_keys_out = carve(n * sizeof(u64));
    const size_t o_vals_in = carve(n * sizeof(unsigned)), o_vals_out = carve(n * sizeof(unsigned));
    const size_t o_left = carve(n * sizeof(int)), o_right = carve(n * sizeof(int));
    const size_t o_rlo = carve(n * sizeof(int)), o_rhi = carve(n * sizeof(int));
    SegLevels seg_levels{};
    size_t seg_boxes = 0;
    for (int count = ni, k = 0;; ++k) {
        if (k >= kMaxSegLevels) { *err = hipErrorInvalidValue; return LbvhStatus::device_error; }
        seg_levels.offset[k] = seg_boxes;
        seg_levels.count[k] = count;
        seg_levels.levels = k + 1;
        seg_boxes += static_cast<size_t>(count);
        if (count == 1) break;
        count = (count + 1) / 2;
    }
    const size_t o_seg = carve(seg_boxes * 6 * sizeof(double));
    const size_t o_zero = off;   // [o_zero, o_zero_end) is cleared: levels, flags
    const size_t o_level = carve(n * sizeof(int)), o_flags = carve(4 * sizeof(unsigned));
    const size_t o_zero_end = off;
    const size_t o_used = carve(n * sizeof(int)), o_widx = carve(n * sizeof(int));
    const size_t o_slots = carve(n * sizeof(int4));
    const size_t o_temp = carve(std::max(sort_bytes, scan_bytes));
    char* base = nullptr;
    if ((e = hipMalloc(reinterpret_cast<void**>(&base), off)) != hipSuccess) { *err = e; return LbvhStatus::device_error; }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    DNode4* d_nodes = nullptr;
    auto fail = [&](hipError_t code) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(base);
        if (d_nodes) (void)hipFree(d_nodes);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        *err = code;
        return LbvhStatus::device_error;
    };
#define LBVH_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(e_); } while (0)
    double* d_boxes = reinterpret_cast<double*>(base + o_boxes);
    double* d_nbox = reinterpret_cast<double*>(base + o_nbox);
    keys_in = reinterpret_cast<u64*>(base + o_keys_in);
    keys_out = reinterpret_cast<u64*>(base + o_keys_out);
    unsigned* vals_in = reinterpret_cast<unsigned*>(base + o_vals_in);
    unsigned* vals_out = reinterpret_cast<unsigned*>(base + o_vals_out);
    int* d_left = reinterpret_cast<int*>(base + o_left);
    int* d_right = reinterpret_cast<int*>(base + o_right);
    int* d_rlo = reinterpret_cast<int*>(base + o_rlo);
    int* d_rhi = reinterpret_cast<int*>(base + o_rhi);
    double* d_seg = reinterpret_cast<double*>(base + o_seg);
    int* d_level = reinterpret_cast<int*>(base + o_level);
    unsigned* d_flags = reinterpret_cast<unsigned*>(base + o_flags);   // [0] too deep, [1] inconsistency, [2] deepest 4-wide level
    int* d_used = reinterpret_cast<int*>(base + o_used);
    int* d_widx = reinterpret_cast<int*>(base + o_widx);
    int4* d_slots = reinterpret_cast<int4*>(base + o_slots);
    void* d_temp = base + o_temp;

    LBVH_TRY(hipEventCreate(&ev0));
    LBVH_TRY(hipEventCreate(&ev1));
    LBVH_TRY(hipMemcpyAsync(d_boxes, boxes.data(), n * 6 * sizeof(double), hipMemcpyHostToDevice, stream));
    LBVH_TRY(hipEventRecord(ev0, stream));
    LBVH_TRY(hipMemsetAsync(base + o_zero, 0, o_zero_end - o_zero, stream));
    const unsigned gl = static_cast<unsigned>((n + kB - 1) / kB);          // one thread per leaf
    const unsigned gn = static_cast<unsigned>((n - 1 + kB - 1) / kB);      // one thread per interior node
    if (binned_sah) {
        bool incomplete = false;
        LBVH_TRY(sah_binary_tree_device(d_boxes, ni, kMaxBinaryDepth, stream, vals_out, d_left, d_right, d_rlo, d_rhi, &incomplete));
        if (incomplete) {   // spans left unsplit after all levels: cannot happen within the depth bound; the host builder takes over
            (void)hipFree(base);
            (void)hipEventDestroy(ev0);
            (void)hipEventDestroy(ev1);
            return LbvhStatus::too_deep;
        }
    } else {
        hipLaunchKernelGGL(k_lbvh_codes, dim3(gl), dim3(kB), 0, stream, d_boxes, ni, grid, keys_in, vals_in);
        size_t tb = sort_bytes;
        LBVH_TRY(rocprim::radix_sort_pairs(d_temp, tb, keys_in, keys_out, vals_in, vals_out, n, 0, 63, stream));
        hipLaunchKernelGGL(k_lbvh_hierarchy, dim3(gn), dim3(kB), 0, stream, keys_out, ni, d_left, d_right, d_rlo, d_rhi);
    }
    hipLaunchKernelGGL(k_lbvh_leaf_boxes, dim3(gl), dim3(kB), 0, stream, d_boxes, vals_out, ni, d_seg, d_flags);
    for (int k = 1; k < seg_levels.levels; ++k)
        hipLaunchKernelGGL(k_lbvh_seg_level, dim3((unsigned)((seg_levels.count[k] + kB - 1) / kB)), dim3(kB), 0, stream,
                           d_seg + 6 * seg_levels.offset[k - 1], seg_levels.count[k - 1], d_seg + 6 * seg_levels.offset[k], seg_levels.count[k]);
    hipLaunchKernelGGL(k_lbvh_node_boxes, dim3(gn), dim3(kB), 0, stream, ni, d_rlo, d_rhi, d_seg, seg_levels, d_nbox, d_flags);
    const int one = 1;
    LBVH_TRY(hipMemcpyAsync(d_level, &one, sizeof(int), hipMemcpyHostToDevice, stream));   // binary node 0 roots the 4-wide root
    int max_levels = kMaxWideLevels;
    if (const char* env = getenv("RT_LBVH_MAX_LEVELS")) max_levels = std::max(1, std::min(kMaxWideLevels, atoi(env)));   // tests of the fallback
    for (int lvl = 1; lvl <= max_levels; ++lvl)
        hipLaunchKernelGGL(k_lbvh_mark, dim3(gn), dim3(kB), 0, stream, ni, lvl, max_levels, d_left, d_right, d_nbox, d_level, d_slots, d_flags);
    hipLaunchKernelGGL(k_lbvh_used, dim3(gn), dim3(kB), 0, stream, ni, d_level, d_used);
    size_t tb = scan_bytes;
    LBVH_TRY(rocprim::exclusive_scan(d_temp, tb, d_used, d_widx, 0, n - 1, rocprim::plus<int>(), stream));
    // the number of 4-wide nodes is needed to size the result: the one round trip of the build
    int last[2] = {0, 0};
    unsigned flags[4] = {0, 0, 0, 0};
    LBVH_TRY(hipMemcpyAsync(&last[0], d_widx + (n - 2), sizeof(int), hipMemcpyDeviceToHost, stream));
    LBVH_TRY(hipMemcpyAsync(&last[1], d_used + (n - 2), sizeof(int), hipMemcpyDeviceToHost, stream));
    LBVH_TRY(hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, stream));
    LBVH_TRY(hipStreamSynchronize(stream));
    LBVH_TRY(hipGetLastError());
    if (flags[1]) return fail(hipErrorUnknown);   // the tree failed its own consistency checks: never traversed
    if (flags[0]) {
        (void)hipFree(base);
        (void)hipEventDestroy(ev0);
        (void)hipEventDestroy(ev1);
        return LbvhStatus::too_deep;
    }
    const int n_wide = last[0] + last[1];
    if (n_wide < 1 || static_cast<size_t>(n_wide) > n - 1) return fail(hipErrorUnknown);
    LBVH_TRY(hipMalloc(reinterpret_cast<void**>(&d_nodes), static_cast<size_t>(n_wide) * sizeof(DNode4)));
    const double pad = node4_pad(root_box, max_abs_origin);
    hipLaunchKernelGGL(k_lbvh_emit, dim3(gn), dim3(kB), 0, stream, ni, pad, d_level, d_widx, d_slots, d_nbox, d_boxes, vals_out, d_nodes,
                       n_wide, d_flags);
    LBVH_TRY(hipEventRecord(ev1, stream));
    LBVH_TRY(hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, stream));
    LBVH_TRY(hipStreamSynchronize(stream));
    LBVH_TRY(hipGetLastError());
    if (flags[1]) return fail(hipErrorUnknown);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ev0, ev1);
#undef LBVH_TRY
    (void)hipFree(base);
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    out->nodes = d_nodes;
    out->n_nodes = n_wide;
    out->root_ref = 0;
    out->max_depth = static_cast<int32_t>(flags[2]);
    out->pad = pad;
    out->build_ms = ms;
    for (int a = 0; a < 3; ++a) {
        out->root_lo[a] = round_down(root_box.lo[a] - pad);
        out->root_hi[a] = round_up(root_box.hi[a] + pad);
    }
    return LbvhStatus::ok;
}

}  // namespace rt
