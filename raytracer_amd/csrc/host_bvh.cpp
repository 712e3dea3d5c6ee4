#include "host_bvh.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>
#include <unordered_map>

namespace rt {

// ---------------------------------------------------------------------------------------------------------------
// Host threads.  All three passes below (reference-order tree, traversal tree, 4-wide collapse) are top-down
// recursions over disjoint index ranges whose output positions are known before a subtree is built, so a step over
// many primitives hands one half to another thread.  Results do not depend on the number of threads.
namespace {

int build_threads() {
    if (const char* e = std::getenv("RT_BUILD_THREADS")) return std::max(1, std::atoi(e));
    // hardware_concurrency() is the whole machine, not this process's share of it: stay modest
    return static_cast<int>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
}

class Tasks {
    std::atomic<int> free_;

public:
    explicit Tasks(int threads) : free_(threads - 1) {}
    bool acquire() {
        int v = free_.load();
        while (v > 0)
            if (free_.compare_exchange_weak(v, v - 1)) return true;
        return false;
    }
    void release() { free_.fetch_add(1); }
};
constexpr size_t kTaskGrain = 8192;   // primitives below which a subtree is not worth a thread

template <class L, class R>
void fork_join(Tasks& tasks, bool big, L&& left, R&& right) {
    if (big && tasks.acquire()) {
        std::thread th([&] { left(); });
        right();
        th.join();
        tasks.release();
    } else {
        left();
        right();
    }
}

template <class F>
void parallel_for(size_t n, F&& body) {   // body(begin, end)
    const size_t threads = std::min<size_t>(static_cast<size_t>(build_threads()), std::max<size_t>(1, n / kTaskGrain));
    if (threads <= 1) { body(size_t(0), n); return; }
    std::vector<std::thread> pool;
    for (size_t t = 1; t < threads; ++t) pool.emplace_back([&, t] { body(n * t / threads, n * (t + 1) / threads); });
    body(size_t(0), n / threads);
    for (std::thread& th : pool) th.join();
}

}  // namespace

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

// What the TRAVERSAL hierarchy is built over.  The reference bounds a transformed sphere by the box of the eight transformed
// corners of [-1, 1]^3 (transformed_shape.h:42-77) -- the box of the ORIENTED BOX around the ellipsoid, up to sqrt(3) too wide per
// axis for a rotated one.  That box has to stay what the reference-order pass sees (its axis choices and centres decide the
// visit order, which decides ties); the hierarchy the device traverses only culls, so any box that contains the surface
// will do, and the ellipsoid {M p + c : |p| <= r} reaches exactly r * |row_i(M)|_2 along axis i.  A ray that misses this box
// misses the ellipsoid, so the closest hit is unchanged.  The rows come from the forward matrix, as the reference's box does,
// widened by 1e-9 of the extent (the surface itself is defined through the inverse matrix the tests use; the two agree to
// ~1e-15 relative) -- five orders of magnitude below the outward pad of the node boxes (node4_pad).
static Box ellipsoid_box(const RtShape& s, double r, double shutter_time) {
    const M4 m = from_row_major(s.transform);
    const bool affine = m.m[3][0] == 0.0 && m.m[3][1] == 0.0 && m.m[3][2] == 0.0 && m.m[3][3] == 1.0;
    if (!affine) return transformed_box(s, V3(-r, -r, -r), V3(r, r, r), shutter_time);
    Box b;
    double c[3], h[3];
    for (int i = 0; i < 3; ++i) {
        c[i] = m.m[i][3];
        h[i] = r * std::sqrt(m.m[i][0] * m.m[i][0] + m.m[i][1] * m.m[i][1] + m.m[i][2] * m.m[i][2]);
        h[i] += 1e-9 * h[i] + 1e-300;
        if (!std::isfinite(h[i])) return transformed_box(s, V3(-r, -r, -r), V3(r, r, r), shutter_time);
    }
    b.lo = V3(c[0] - h[0], c[1] - h[1], c[2] - h[2]);
    b.hi = V3(c[0] + h[0], c[1] + h[1], c[2] + h[2]);
    const V3 d = V3(s.velocity) * shutter_time;   // every position of the centre over the shutter interval (sphere.cpp:47)
    return combine(b, Box{b.lo + d, b.hi + d});
}

Box traversal_bounds(const RtShape& s, double shutter_time) {
    if (std::getenv("RT_REFERENCE_BOXES")) return shape_bounds(s, shutter_time);   // A/B runs: the reference's boxes everywhere
    if (s.type == RT_SHAPE_SPHERE) return ellipsoid_box(s, 1.0, shutter_time);
    if (s.type == RT_SHAPE_COMPLEX_SPHERE) return ellipsoid_box(s, 1.0 + s.max_displacement, shutter_time);   // complex_sphere.cpp:30-60: hits lie within |p| <= 1 + D
    return shape_bounds(s, shutter_time);
}

namespace {

struct Builder {
    const std::vector<Box>& boxes;
    std::vector<V3> centre;          // (lo+hi)*0.5, the sort key of bvh.cpp:14-15
    std::vector<int32_t> idx;        // working permutation of shape indices; the depth-first leaf order when build() is done
    HostBvh out;
    Tasks tasks{build_threads()};
    std::unordered_map<size_t, size_t> node_count;   // interior nodes of a subtree over n shapes (filled before build())

    struct Built {
        int32_t ref;
        Box box;
        int32_t depth;
    };

    bool less_on(int axis, int32_t a, int32_t b) const { return centre[a][axis] < centre[b][axis]; }

    // The split sizes depend on n only (bvh.cpp:60-78), so the nodes of a subtree occupy a known range of out.nodes
    // (depth-first: node, left subtree, right subtree) and its leaves the range [start, end) of the order.
    size_t count_nodes(size_t n) {
        if (n <= 2) return 0;
        auto it = node_count.find(n);
        if (it != node_count.end()) return it->second;
        const size_t c = 1 + count_nodes(n / 2) + count_nodes(n - n / 2);
        node_count.emplace(n, c);
        return c;
    }

    // One reference BVHNode over idx[start, end), stored at out.nodes[me].  Nodes whose two children are shapes become
    // leaf references; everything else becomes a DNode carrying its children's boxes.
    Built build(size_t start, size_t end, size_t me) {
        Box span = boxes[idx[start]];
        for (size_t i = start + 1; i < end; ++i) span = combine(span, boxes[idx[i]]);
        V3 ext = span.hi - span.lo;
        int axis = 0;  // bvh.cpp:52-58: strict comparisons, x on ties
        if (ext.y > ext.x && ext.y > ext.z) axis = 1;
        else if (ext.z > ext.x && ext.z > ext.y) axis = 2;

        size_t n = end - start;
        if (n == 1)  // left == right == the single shape
            return {make_leaf_ref(static_cast<int32_t>(start), 1), boxes[idx[start]], 0};
        if (n == 2) {  // ordered pair (bvh.cpp:67-75)
            int32_t a = idx[start], b = idx[start + 1];
            if (!less_on(axis, a, b)) std::swap(a, b);
            idx[start] = a;
            idx[start + 1] = b;
            return {make_leaf_ref(static_cast<int32_t>(start), 2), combine(boxes[a], boxes[b]), 0};
        }
        size_t mid = start + n / 2;
        std::nth_element(idx.begin() + start, idx.begin() + mid, idx.begin() + end,
                         [&](int32_t a, int32_t b) { return less_on(axis, a, b); });
        Built l, r;
        const size_t left_nodes = node_count.count(mid - start) ? node_count.at(mid - start) : 0;
        fork_join(tasks, n > kTaskGrain, [&] { l = build(start, mid, me + 1); }, [&] { r = build(mid, end, me + 1 + left_nodes); });
        DNode& nd = out.nodes[me];
        store(nd.lo[0], l.box.lo); store(nd.hi[0], l.box.hi);
        store(nd.lo[1], r.box.lo); store(nd.hi[1], r.box.hi);
        nd.child[0] = l.ref;
        nd.child[1] = r.ref;
        for (int& p : nd.pad) p = 0;
        return {static_cast<int32_t>(me), combine(l.box, r.box), 1 + std::max(l.depth, r.depth)};
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
    static constexpr size_t kWideStep = size_t(1) << 17;
    const std::vector<Box>& boxes;   // per primitive, indexed by reference depth-first order
    std::vector<V3> centre;
    std::vector<int32_t> idx;
    HostBvh out;
    Tasks tasks{build_threads()};

    struct Built {
        int32_t ref;
        Box box;
        int32_t depth;
    };

    // depth_left = interior levels this subtree may use; callers guarantee ceil_log2(n) <= depth_left.
    // A subtree over n primitives has n - 1 interior nodes: they go to out.nodes[me, me + n - 1), depth-first.
    Built build(size_t start, size_t end, int depth_left, size_t me) {
        const size_t n = end - start;
        if (n == 1) return {make_leaf_ref(idx[start], 1), boxes[idx[start]], 0};
        const double inf = std::numeric_limits<double>::infinity();
        const Box empty{{inf, inf, inf}, {-inf, -inf, -inf}};
        const bool wide = n >= kWideStep;   // near the root there are fewer subtrees than threads: split the passes over the range instead
        Box cb = empty;
        if (wide) {
            std::mutex m;
            parallel_for(n, [&](size_t b0, size_t e0) {
                Box part = empty;
                for (size_t i = start + b0; i < start + e0; ++i) grow(part, centre[idx[i]]);
                std::lock_guard<std::mutex> g(m);
                cb = combine(cb, part);
            });
        } else {
            for (size_t i = start; i < end; ++i) grow(cb, centre[idx[i]]);
        }
        size_t mid = start + n / 2;
        bool split_done = false;
        if (n > 2) {
            // bins of all three axes in one pass over the primitives (min / max / counts: the order does not matter)
            struct Bins {
                Box bb[3][kBins];
                size_t cnt[3][kBins];
            };
            double lo3[3], scale3[3];
            bool use[3];
            for (int axis = 0; axis < 3; ++axis) {
                const double ext = cb.hi[axis] - cb.lo[axis];
                use[axis] = ext > 0.0 && std::isfinite(ext);
                lo3[axis] = cb.lo[axis];
                scale3[axis] = use[axis] ? kBins / ext : 0.0;
            }
            auto clear = [&](Bins& B) {
                for (int axis = 0; axis < 3; ++axis)
                    for (int b = 0; b < kBins; ++b) { B.bb[axis][b] = empty; B.cnt[axis][b] = 0; }
            };
            auto fill = [&](Bins& B, size_t b0, size_t e0) {
                for (size_t i = b0; i < e0; ++i) {
                    const int32_t p = idx[i];
                    const Box& pb = boxes[p];
                    for (int axis = 0; axis < 3; ++axis) {
                        if (!use[axis]) continue;
                        int b = std::min(kBins - 1, std::max(0, static_cast<int>((centre[p][axis] - lo3[axis]) * scale3[axis])));
                        B.bb[axis][b] = combine(B.bb[axis][b], pb);
                        ++B.cnt[axis][b];
                    }
                }
            };
            Bins bins;
            clear(bins);
            if (wide) {
                std::mutex m;
                parallel_for(n, [&](size_t b0, size_t e0) {
                    auto part = std::make_unique<Bins>();
                    clear(*part);
                    fill(*part, start + b0, start + e0);
                    std::lock_guard<std::mutex> g(m);
                    for (int axis = 0; axis < 3; ++axis)
                        for (int b = 0; b < kBins; ++b) {
                            bins.bb[axis][b] = combine(bins.bb[axis][b], part->bb[axis][b]);
                            bins.cnt[axis][b] += part->cnt[axis][b];
                        }
                });
            } else {
                fill(bins, start, end);
            }
            double best_cost = inf;
            int best_axis = -1, best_bin = -1;
            for (int axis = 0; axis < 3; ++axis) {
                if (!use[axis]) continue;
                const Box* bb = bins.bb[axis];
                const size_t* cnt = bins.cnt[axis];
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
        Built l, r;
        fork_join(tasks, n > kTaskGrain, [&] { l = build(start, mid, depth_left - 1, me + 1); },
                  [&] { r = build(mid, end, depth_left - 1, me + (mid - start)); });
        DNode& nd = out.nodes[me];
        store(nd.lo[0], l.box.lo); store(nd.hi[0], l.box.hi);
        store(nd.lo[1], r.box.lo); store(nd.hi[1], r.box.hi);
        nd.child[0] = l.ref;
        nd.child[1] = r.ref;
        for (int& p : nd.pad) p = 0;
        return {static_cast<int32_t>(me), combine(l.box, r.box), 1 + std::max(l.depth, r.depth)};
    }
};

}  // namespace

double node4_pad(const Box& root_box, double max_abs_origin) {
    double m = max_abs_origin;
    auto grow_abs = [&](V3 p) { for (int a = 0; a < 3; ++a) if (std::isfinite(p[a])) m = std::max(m, std::fabs(p[a])); };
    grow_abs(root_box.lo);
    grow_abs(root_box.hi);
    // FP32 errors of the device slab test (device_rt.h: node4_test), as plane displacements in world units, for
    // coordinates bounded by m.  Absolute ones: rounding the ray origin (2^-24 m), the product o*inv (2^-24 m),
    // org*inv - o*inv (2^-23 m) -- under 5 * 2^-24 m in total, covered by 2^-21 m.  Relative ones: the reciprocal of
    // the direction (v_rcp_f32 of the rounded component: 1.5 * 2^-23 + 2^-24) and the final FMA (2^-24) -- under
    // 1.25 * 2^-22 of the slab distance, i.e. of |plane - o| <= 2 m along the axis: under 1.25 * 2^-21 m, covered by
    // 2^-19 m (3.2 times over).  Ray origins are the camera or surface points inside the root box moved by at most
    // ~1e-3 (tracer.h:201, shading.h:68,93), hence the 0.01.
    return (m + 0.01) * (std::ldexp(1.0, -21) + std::ldexp(1.0, -19));
}
float round_down(double v) {
    float f = static_cast<float>(v);
    if (static_cast<double>(f) > v) f = std::nextafterf(f, -INFINITY);
    return f;
}
float round_up(double v) {
    float f = static_cast<float>(v);
    if (static_cast<double>(f) < v) f = std::nextafterf(f, INFINITY);
    return f;
}

HostBvh build_traversal_bvh(const std::vector<Box>& boxes, int max_depth) {
    SahBuilder b{boxes};
    const size_t n = boxes.size();
    if (n == 0) {
        b.out.root_ref = make_leaf_ref(0, 1);
        b.out.root_box = Box{};
        return std::move(b.out);
    }
    b.centre.resize(n);
    b.idx.resize(n);
    parallel_for(n, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            b.centre[i] = (boxes[i].lo + boxes[i].hi) * 0.5;
            b.idx[i] = static_cast<int32_t>(i);
        }
    });
    b.out.nodes.resize(n - 1);
    SahBuilder::Built root = b.build(0, n, std::max(max_depth, ceil_log2(n)), 0);
    b.out.root_ref = root.ref;
    b.out.root_box = root.box;
    b.out.max_depth = root.depth;
    return std::move(b.out);
}

namespace {

struct Collapser {
    const HostBvh& bvh;
    double pad;
    Bvh4& out;
    std::vector<int32_t> wide_count;   // per binary node that roots a 4-wide node: 4-wide nodes of its subtree (pass 1)
    Tasks tasks{build_threads()};
    static constexpr int kForkLevels = 4;   // 4-wide levels from the root whose children may go to other threads

    struct Slot {
        int32_t ref;
        Box box;
    };

    // The children of the 4-wide node rooted at binary interior node b.
    std::vector<Slot> open(int32_t b) const {
        std::vector<Slot> slots;
        slots.reserve(4);
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
        return slots;
    }

    template <class F>
    void for_children(const std::vector<Slot>& slots, int level, F&& f) {   // f(i) for every slot, the two halves on two threads near the root
        const size_t h = slots.size() / 2;
        fork_join(tasks, level < kForkLevels,
                  [&] { for (size_t i = 0; i < h; ++i) f(i); },
                  [&] { for (size_t i = h; i < slots.size(); ++i) f(i); });
    }

    // Pass 1: number of 4-wide nodes below (and including) the one rooted at b, so that pass 2 knows where every
    // subtree goes (depth-first: node, then the subtrees of its children in slot order).
    int32_t count(int32_t b, int level) {
        const std::vector<Slot> slots = open(b);
        int32_t sub[4] = {0, 0, 0, 0};
        for_children(slots, level, [&](size_t i) { if (slots[i].ref >= 0) sub[i] = count(slots[i].ref, level + 1); });
        return wide_count[b] = 1 + sub[0] + sub[1] + sub[2] + sub[3];
    }

    // Pass 2: the 4-wide node for the binary subtree rooted at interior node b, written to out.nodes[me]; returns its depth (levels).
    int32_t emit(int32_t b, int32_t me, int level) {
        const std::vector<Slot> slots = open(b);
        int32_t refs[4] = {kEmptyChild, kEmptyChild, kEmptyChild, kEmptyChild};
        int32_t depths[4] = {0, 0, 0, 0};
        int32_t next = me + 1;
        for (size_t i = 0; i < slots.size(); ++i) {
            if (slots[i].ref >= 0) {
                refs[i] = next;
                next += wide_count[slots[i].ref];
            } else {
                refs[i] = slots[i].ref;
            }
        }
        for_children(slots, level, [&](size_t i) { if (slots[i].ref >= 0) depths[i] = emit(slots[i].ref, refs[i], level + 1); });
        quantise(out.nodes[me], slots, refs);
        return 1 + std::max(std::max(depths[0], depths[1]), std::max(depths[2], depths[3]));
    }

    void quantise(DNode4& o, const std::vector<Slot>& slots, const int32_t refs[4]) const {
        for (int a = 0; a < 3; ++a) {
            o.qlo[a] = 0xffffffffu;   // unused slots: lower plane above upper plane
            o.qhi[a] = 0u;
        }
        for (int c = 0; c < 4; ++c) o.child[c] = refs[c];
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
            (a == 0 ? o.step_x : o.step_yz[a - 1]) = std::ldexp(1.0f, e);   // -126 <= e <= 127: a normal FP32 number, never zero
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
    double rl[3], rh[3];
    store(rl, bvh.root_box.lo);
    store(rh, bvh.root_box.hi);
    out.pad = node4_pad(bvh.root_box, max_abs_origin);
    out.root_ref = bvh.root_ref;
    out.max_depth = 0;
    if (!bvh.nodes.empty() && bvh.root_ref >= 0) {
        Collapser c{bvh, out.pad, out};
        c.wide_count.assign(bvh.nodes.size(), 0);
        out.nodes.resize(static_cast<size_t>(c.count(bvh.root_ref, 0)));
        out.max_depth = c.emit(bvh.root_ref, 0, 0);
        out.root_ref = 0;
    }
    for (int a = 0; a < 3; ++a) {
        out.root_lo[a] = round_down(rl[a] - out.pad);
        out.root_hi[a] = round_up(rh[a] + out.pad);
    }
    return out;
}

std::vector<Box> shape_boxes(const RtSceneDesc& desc, const int32_t* order) {
    const size_t n = static_cast<size_t>(std::max(desc.n_shapes, 0));
    std::vector<Box> boxes(n);
    parallel_for(n, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) boxes[i] = shape_bounds(desc.shapes[order ? order[i] : static_cast<int32_t>(i)], desc.shutter_time);
    });
    return boxes;
}

std::vector<Box> traversal_boxes(const RtSceneDesc& desc, const std::vector<int32_t>& order) {
    std::vector<Box> boxes(order.size());
    const bool reference = std::getenv("RT_REFERENCE_BOXES") != nullptr;
    parallel_for(order.size(), [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i)
            boxes[i] = reference ? shape_bounds(desc.shapes[order[i]], desc.shutter_time) : traversal_bounds(desc.shapes[order[i]], desc.shutter_time);
    });
    return boxes;
}

std::vector<Box> permuted_boxes(const std::vector<Box>& boxes, const std::vector<int32_t>& order) {
    std::vector<Box> out(order.size());
    parallel_for(order.size(), [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) out[i] = boxes[order[i]];
    });
    return out;
}

void parallel_ranges(size_t n, const std::function<void(size_t, size_t)>& body) { parallel_for(n, body); }

Box union_of(const std::vector<Box>& boxes) {
    const double inf = std::numeric_limits<double>::infinity();
    Box all{{inf, inf, inf}, {-inf, -inf, -inf}};
    std::mutex m;
    parallel_for(boxes.size(), [&](size_t b, size_t e) {
        Box part{{inf, inf, inf}, {-inf, -inf, -inf}};
        for (size_t i = b; i < e; ++i) part = combine(part, boxes[i]);
        std::lock_guard<std::mutex> g(m);
        all = combine(all, part);
    });
    return all;
}

HostBvh build_bvh(const RtSceneDesc& desc) {
    const size_t n = static_cast<size_t>(std::max(desc.n_shapes, 0));
    std::vector<Box> boxes = shape_boxes(desc, nullptr);
    Builder b{boxes};
    if (n == 0) {
        b.out.root_ref = make_leaf_ref(0, 1);
        b.out.root_box = Box{};
        return std::move(b.out);
    }
    b.centre.resize(n);
    b.idx.resize(n);
    parallel_for(n, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            b.centre[i] = (boxes[i].lo + boxes[i].hi) * 0.5;
            b.idx[i] = static_cast<int32_t>(i);
        }
    });
    b.out.nodes.resize(b.count_nodes(n));
    Builder::Built root = b.build(0, n, 0);
    b.out.order = std::move(b.idx);
    b.out.shape_boxes = std::move(boxes);   // b.boxes refers to it: not used after this point
    b.out.root_ref = root.ref;
    b.out.root_box = root.box;
    b.out.max_depth = root.depth;
    return std::move(b.out);
}

}  // namespace rt
