#include "host_bvh.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>

namespace rt {

static Box transformed_box(const RtShape& s, V3 lmin, V3 lmax, double shutter_time) {
    const double inf = std::numeric_limits<double>::infinity();
    M4 m = from_row_major(s.transform);
    Box b{{inf, inf, inf}, {-inf, -inf, -inf}};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k)
                grow(b, transform_point(m, V3(i ? lmax.x : lmin.x, j ? lmax.y : lmin.y, k ? lmax.z : lmin.z)));
    V3 d = V3(s.velocity) * shutter_time;  // box at shutter open  U  box at shutter close
    return combine(b, Box{b.lo + d, b.hi + d});
}

Box shape_bounds(const RtShape& s, double shutter_time) {
    switch (s.type) {
        case RT_SHAPE_SPHERE:
        case RT_SHAPE_CUBE:
            return transformed_box(s, V3(-1, -1, -1), V3(1, 1, 1), shutter_time);
        case RT_SHAPE_COMPLEX_SPHERE:
        case RT_SHAPE_COMPLEX_CUBE: {
            double r = 1.0 + s.max_displacement;
            return transformed_box(s, V3(-r, -r, -r), V3(r, r, r), shutter_time);
        }
        case RT_SHAPE_COMPLEX_PLANE: {
            // complex_plane.cpp:41-60: per corner, the swept end point is added only when |velocity| > 1e-6.
            // (The reference reads an uninitialised m_shutter_time there, complex_plane.cpp:13-21; the scene's
            // shutter time is used here.)
            const double inf = std::numeric_limits<double>::infinity();
            double zb = s.max_displacement + 0.01;
            M4 m = from_row_major(s.transform);
            V3 vel(s.velocity);
            bool moving = length(vel) > 1e-6;
            Box b{{inf, inf, inf}, {-inf, -inf, -inf}};
            for (int i = 0; i < 8; ++i) {
                V3 p = transform_point(m, V3((i & 1) ? 1.0 : -1.0, (i & 2) ? 1.0 : -1.0, (i & 4) ? zb : -zb));
                grow(b, p);
                if (moving) grow(b, p + vel * shutter_time);
            }
            return b;
        }
        case RT_SHAPE_PLANE:
        default: {
            // plane.cpp:11-46: corners are re-derived from the stored edges, then padded by 1e-4.
            V3 c0(s.corners), c1(s.corners + 3), c2(s.corners + 6), c3(s.corners + 9);
            V3 v1 = c0 + (c1 - c0), v2 = c0 + (c2 - c0), v3 = c1 + (c3 - c1);
            Box b{c0, c0};
            grow(b, v1);
            grow(b, v2);
            grow(b, v3);
            const double pad = 1e-4;
            b.lo = b.lo - V3(pad, pad, pad);
            b.hi = b.hi + V3(pad, pad, pad);
            V3 d = V3(s.velocity) * shutter_time;
            return combine(b, Box{b.lo + d, b.hi + d});
        }
    }
}

namespace {

struct Builder {
    const std::vector<Box>& boxes;
    std::vector<V3> centre;          // (lo+hi)*0.5, the sort key of bvh.cpp:14-15
    std::vector<int32_t> idx;        // working permutation of shape indices
    HostBvh out;

    struct Built {
        int32_t ref;
        Box box;
        int32_t depth;
    };

    bool less_on(int axis, int32_t a, int32_t b) const { return centre[a][axis] < centre[b][axis]; }

    // One reference BVHNode over idx[start, end).  Nodes whose two children are shapes become leaf references;
    // everything else becomes a DNode carrying its children's boxes.
    Built build(size_t start, size_t end) {
        Box span = boxes[idx[start]];
        for (size_t i = start + 1; i < end; ++i) span = combine(span, boxes[idx[i]]);
        V3 ext = span.hi - span.lo;
        int axis = 0;  // bvh.cpp:52-58: strict comparisons, x on ties
        if (ext.y > ext.x && ext.y > ext.z) axis = 1;
        else if (ext.z > ext.x && ext.z > ext.y) axis = 2;

        size_t n = end - start;
        if (n == 1) {  // left == right == the single shape
            int32_t first = static_cast<int32_t>(out.order.size());
            out.order.push_back(idx[start]);
            return {make_leaf_ref(first, 1), boxes[idx[start]], 0};
        }
        if (n == 2) {  // ordered pair (bvh.cpp:67-75)
            int32_t a = idx[start], b = idx[start + 1];
            if (!less_on(axis, a, b)) std::swap(a, b);
            int32_t first = static_cast<int32_t>(out.order.size());
            out.order.push_back(a);
            out.order.push_back(b);
            return {make_leaf_ref(first, 2), combine(boxes[a], boxes[b]), 0};
        }
        size_t mid = start + n / 2;
        std::nth_element(idx.begin() + start, idx.begin() + mid, idx.begin() + end,
                         [&](int32_t a, int32_t b) { return less_on(axis, a, b); });
        int32_t me = static_cast<int32_t>(out.nodes.size());
        out.nodes.emplace_back();
        Built l = build(start, mid);
        Built r = build(mid, end);
        DNode& nd = out.nodes[me];
        store(nd.lo[0], l.box.lo); store(nd.hi[0], l.box.hi);
        store(nd.lo[1], r.box.lo); store(nd.hi[1], r.box.hi);
        nd.child[0] = l.ref;
        nd.child[1] = r.ref;
        for (int& p : nd.pad) p = 0;
        return {me, combine(l.box, r.box), 1 + std::max(l.depth, r.depth)};
    }
};

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// Traversal hierarchy.  The reference's median split leaves pairs of far-apart shapes under one box (two sparse
// shapes share a mostly empty box), which costs ~16 primitive tests and ~90 node visits per ray on the 10^6-shape
// stress scene.  The closest hit does not depend on the hierarchy (ties: device_rt.h tie_new_wins, keyed by the
// reference's visit order, which stays the primitive index), so the device traverses a binned-SAH tree with
// single-primitive leaves built over the same per-shape boxes.
namespace {

inline double half_area(const Box& b) {
    V3 e = b.hi - b.lo;
    return e.x * e.y + e.y * e.z + e.z * e.x;
}
inline int ceil_log2(size_t n) {
    int l = 0;
    while ((size_t(1) << l) < n) ++l;
    return l;
}

struct SahBuilder {
    static constexpr int kBins = 32;
    const std::vector<Box>& boxes;   // per primitive, indexed by reference depth-first order
    std::vector<V3> centre;
    std::vector<int32_t> idx;
    HostBvh out;

    struct Built {
        int32_t ref;
        Box box;
        int32_t depth;
    };

    // depth_left = interior levels this subtree may use; callers guarantee ceil_log2(n) <= depth_left.
    Built build(size_t start, size_t end, int depth_left) {
        const size_t n = end - start;
        if (n == 1) return {make_leaf_ref(idx[start], 1), boxes[idx[start]], 0};
        const double inf = std::numeric_limits<double>::infinity();
        Box cb{{inf, inf, inf}, {-inf, -inf, -inf}};
        for (size_t i = start; i < end; ++i) grow(cb, centre[idx[i]]);
        size_t mid = start + n / 2;
        bool split_done = false;
        if (n > 2) {
            double best_cost = inf;
            int best_axis = -1, best_bin = -1;
            for (int axis = 0; axis < 3; ++axis) {
                const double lo = cb.lo[axis], ext = cb.hi[axis] - cb.lo[axis];
                if (!(ext > 0.0) || !std::isfinite(ext)) continue;
                const double scale = kBins / ext;
                Box bb[kBins];
                size_t cnt[kBins];
                for (int b = 0; b < kBins; ++b) { bb[b] = Box{{inf, inf, inf}, {-inf, -inf, -inf}}; cnt[b] = 0; }
                for (size_t i = start; i < end; ++i) {
                    int b = std::min(kBins - 1, std::max(0, static_cast<int>((centre[idx[i]][axis] - lo) * scale)));
                    bb[b] = combine(bb[b], boxes[idx[i]]);
                    ++cnt[b];
                }
                double right_area[kBins];
                size_t right_cnt[kBins];
                Box acc{{inf, inf, inf}, {-inf, -inf, -inf}};
                size_t c = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    if (cnt[b]) acc = combine(acc, bb[b]);
                    c += cnt[b];
                    right_area[b] = c ? half_area(acc) : 0.0;
                    right_cnt[b] = c;
                }
                acc = Box{{inf, inf, inf}, {-inf, -inf, -inf}};
                c = 0;
                for (int b = 0; b + 1 < kBins; ++b) {   // split between bin b and b + 1
                    if (cnt[b]) acc = combine(acc, bb[b]);
                    c += cnt[b];
                    const size_t nl = c, nr = right_cnt[b + 1];
                    if (nl == 0 || nr == 0) continue;
                    if (ceil_log2(std::max(nl, nr)) > depth_left - 1) continue;   // keep the stack bound
                    double cost = half_area(acc) * static_cast<double>(nl) + right_area[b + 1] * static_cast<double>(nr);
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                const double lo = cb.lo[best_axis], scale = kBins / (cb.hi[best_axis] - cb.lo[best_axis]);
                auto it = std::partition(idx.begin() + start, idx.begin() + end, [&](int32_t p) {
                    int b = std::min(kBins - 1, std::max(0, static_cast<int>((centre[p][best_axis] - lo) * scale)));
                    return b <= best_bin;
                });
                mid = static_cast<size_t>(it - idx.begin());
                split_done = mid > start && mid < end;
                if (!split_done) mid = start + n / 2;
            }
        }
        if (!split_done) {   // two shapes, coincident centres or no depth to spare: balanced split on the widest axis
            V3 ext = cb.hi - cb.lo;
            int axis = 0;
            if (ext.y > ext.x && ext.y > ext.z) axis = 1;
            else if (ext.z > ext.x && ext.z > ext.y) axis = 2;
            std::nth_element(idx.begin() + start, idx.begin() + mid, idx.begin() + end, [&](int32_t a, int32_t b) {
                return centre[a][axis] < centre[b][axis] || (centre[a][axis] == centre[b][axis] && a < b);
            });
        }
        int32_t me = static_cast<int32_t>(out.nodes.size());
        out.nodes.emplace_back();
        Built l = build(start, mid, depth_left - 1);
        Built r = build(mid, end, depth_left - 1);
        DNode& nd = out.nodes[me];
        store(nd.lo[0], l.box.lo); store(nd.hi[0], l.box.hi);
        store(nd.lo[1], r.box.lo); store(nd.hi[1], r.box.hi);
        nd.child[0] = l.ref;
        nd.child[1] = r.ref;
        for (int& p : nd.pad) p = 0;
        return {me, combine(l.box, r.box), 1 + std::max(l.depth, r.depth)};
    }
};

}  // namespace

HostBvh build_traversal_bvh(const std::vector<Box>& boxes, int max_depth) {
    SahBuilder b{boxes, {}, {}, {}};
    const size_t n = boxes.size();
    if (n == 0) {
        b.out.root_ref = make_leaf_ref(0, 1);
        b.out.root_box = Box{};
        return b.out;
    }
    b.centre.resize(n);
    for (size_t i = 0; i < n; ++i) b.centre[i] = (boxes[i].lo + boxes[i].hi) * 0.5;
    b.idx.resize(n);
    std::iota(b.idx.begin(), b.idx.end(), 0);
    b.out.nodes.reserve(n);
    SahBuilder::Built root = b.build(0, n, std::max(max_depth, ceil_log2(n)));
    b.out.root_ref = root.ref;
    b.out.root_box = root.box;
    b.out.max_depth = root.depth;
    return std::move(b.out);
}

static float round_down(double v) {
    float f = static_cast<float>(v);
    if (static_cast<double>(f) > v) f = std::nextafterf(f, -INFINITY);
    return f;
}
static float round_up(double v) {
    float f = static_cast<float>(v);
    if (static_cast<double>(f) < v) f = std::nextafterf(f, INFINITY);
    return f;
}

namespace {

struct Collapser {
    const HostBvh& bvh;
    double pad;
    Bvh4& out;

    struct Slot {
        int32_t ref;
        Box box;
    };

    // One 4-wide node for the binary subtree rooted at interior node b; returns its index and depth (levels).
    std::pair<int32_t, int32_t> emit(int32_t b) {
        const int32_t me = static_cast<int32_t>(out.nodes.size());
        out.nodes.emplace_back();
        std::vector<Slot> slots;
        auto children_of = [&](int32_t n, Slot& l, Slot& r) {
            const DNode& nd = bvh.nodes[n];
            l = Slot{nd.child[0], Box{V3(nd.lo[0]), V3(nd.hi[0])}};
            r = Slot{nd.child[1], Box{V3(nd.lo[1]), V3(nd.hi[1])}};
        };
        Slot l, r;
        children_of(b, l, r);
        slots.push_back(l);
        slots.push_back(r);
        while (slots.size() < 4) {   // open the interior child with the largest surface: it is the most likely to be entered
            int pick = -1;
            double best = -1.0;
            for (size_t i = 0; i < slots.size(); ++i) {
                if (slots[i].ref < 0) continue;
                double a = half_area(slots[i].box);
                if (!(a <= best)) { best = a; pick = static_cast<int>(i); }
            }
            if (pick < 0) break;
            children_of(slots[pick].ref, l, r);
            slots[pick] = l;
            slots.push_back(r);
        }
        int32_t depth = 1;
        int32_t refs[4] = {kEmptyChild, kEmptyChild, kEmptyChild, kEmptyChild};
        for (size_t i = 0; i < slots.size(); ++i) {
            if (slots[i].ref >= 0) {
                auto sub = emit(slots[i].ref);
                refs[i] = sub.first;
                depth = std::max(depth, 1 + sub.second);
            } else {
                refs[i] = slots[i].ref;
            }
        }
        quantise(out.nodes[me], slots, refs);
        return {me, depth};
    }

    void quantise(DNode4& o, const std::vector<Slot>& slots, const int32_t refs[4]) const {
        o.exps = 0;
        for (int a = 0; a < 3; ++a) {
            o.qlo[a] = 0xffffffffu;   // unused slots: lower plane above upper plane
            o.qhi[a] = 0u;
        }
        for (int c = 0; c < 4; ++c) o.child[c] = refs[c];
        o.pad[0] = o.pad[1] = 0;
        for (int a = 0; a < 3; ++a) {
            double lo = std::numeric_limits<double>::infinity(), hi = -lo;
            for (const Slot& s : slots) {
                lo = std::min(lo, s.box.lo[a] - pad);
                hi = std::max(hi, s.box.hi[a] + pad);
            }
            if (!std::isfinite(lo)) lo = -std::numeric_limits<float>::max();
            if (!std::isfinite(hi)) hi = std::numeric_limits<float>::max();
            const float org = round_down(lo);
            // smallest power-of-two step with 255 steps covering [org, hi]
            int e = -126;
            const double ext = hi - static_cast<double>(org);
            if (ext > 0.0) {
                int ex;
                (void)std::frexp(ext / 255.0, &ex);   // ext/255 = m * 2^ex, m in [0.5, 1)  =>  2^ex >= ext/255
                e = std::max(-126, std::min(127, ex));
            }
            auto q_of = [&](double v, bool up, int ee) {
                double g = (v - static_cast<double>(org)) / std::ldexp(1.0, ee);
                g = up ? std::ceil(g) : std::floor(g);
                return g;
            };
            while (e < 127 && q_of(hi, true, e) > 255.0) ++e;
            o.org[a] = org;
            o.exps |= static_cast<uint32_t>(e + 127) << (8 * a);
            uint32_t wlo = o.qlo[a], whi = o.qhi[a];
            for (size_t c = 0; c < slots.size(); ++c) {
                double ql = q_of(std::max(slots[c].box.lo[a] - pad, static_cast<double>(org)), false, e);
                double qh = q_of(std::min(slots[c].box.hi[a] + pad, hi), true, e);
                if (!(ql >= 0.0)) ql = 0.0;       // also NaN
                if (!(qh <= 255.0)) qh = 255.0;
                if (ql > 255.0) ql = 255.0;
                if (qh < 0.0) qh = 0.0;
                wlo = (wlo & ~(0xffu << (8 * c))) | (static_cast<uint32_t>(ql) << (8 * c));
                whi = (whi & ~(0xffu << (8 * c))) | (static_cast<uint32_t>(qh) << (8 * c));
            }
            o.qlo[a] = wlo;
            o.qhi[a] = whi;
        }
    }
};

}  // namespace

Bvh4 to_node4(const HostBvh& bvh, double max_abs_origin) {
    Bvh4 out;
    double m = max_abs_origin;
    auto grow_abs = [&](const double* p) { for (int a = 0; a < 3; ++a) if (std::isfinite(p[a])) m = std::max(m, std::fabs(p[a])); };
    double rl[3], rh[3];
    store(rl, bvh.root_box.lo);
    store(rh, bvh.root_box.hi);
    grow_abs(rl);
    grow_abs(rh);
    // Absolute FP32 errors of the device slab test (device_rt.h: node4_test), in world units, for coordinates bounded
    // by m: rounding the ray origin (2^-24 m), the product o*inv (2^-24 m), org*inv - o*inv (2^-23 m) -- under
    // 5 * 2^-24 m in total.  Ray origins are the camera or surface points inside the root box moved by at most ~1e-3
    // (tracer.h:201, shading.h:68,93).  Relative errors are covered on the device.
    out.pad = (m + 0.01) * std::ldexp(1.0, -21);
    out.root_ref = bvh.root_ref;
    out.max_depth = 0;
    if (!bvh.nodes.empty() && bvh.root_ref >= 0) {
        out.nodes.reserve(bvh.nodes.size() / 2 + 16);
        Collapser c{bvh, out.pad, out};
        auto root = c.emit(bvh.root_ref);
        out.root_ref = root.first;
        out.max_depth = root.second;
    }
    for (int a = 0; a < 3; ++a) {
        out.root_lo[a] = round_down(rl[a] - out.pad);
        out.root_hi[a] = round_up(rh[a] + out.pad);
    }
    return out;
}

HostBvh build_bvh(const RtSceneDesc& desc) {
    std::vector<Box> boxes(desc.n_shapes);
    for (int i = 0; i < desc.n_shapes; ++i) boxes[i] = shape_bounds(desc.shapes[i], desc.shutter_time);
    Builder b{boxes, {}, {}, {}};
    if (desc.n_shapes == 0) {
        b.out.root_ref = make_leaf_ref(0, 1);
        b.out.root_box = Box{};
        return b.out;
    }
    b.centre.resize(desc.n_shapes);
    for (int i = 0; i < desc.n_shapes; ++i) b.centre[i] = (boxes[i].lo + boxes[i].hi) * 0.5;
    b.idx.resize(desc.n_shapes);
    std::iota(b.idx.begin(), b.idx.end(), 0);
    b.out.nodes.reserve(desc.n_shapes);
    b.out.order.reserve(desc.n_shapes);
    Builder::Built root = b.build(0, desc.n_shapes);
    b.out.root_ref = root.ref;
    b.out.root_box = root.box;
    b.out.max_depth = root.depth;
    return std::move(b.out);
}

}  // namespace rt
