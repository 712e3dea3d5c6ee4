// Model of what an 8-wide, octant-ordered node would cost the traversal of the C2 frame, run on the host (no GPU):
// the binned-SAH binary tree of host_bvh.cpp over the scene's traversal boxes is collapsed to W children per node, every
// child box quantised to 8 bits per plane on the node's grid like DNode4's, and the primary rays of every k-th pixel are
// traced to their closest ellipsoid with
//   A  W = 4, children sorted by entry distance, (child, distance) pairs deferred, stale entries skipped at the pop:
//      what k_traverse does (its parked leaves aside) -- this row must reproduce the device's counters;
//   B  W = 8, the same, as the upper bound of what eight children can give;
//   C  W = 8, children in slots by octant, visited in the order slot ^ ray octant, ONE deferred entry per node (the mask of
//      the hit children that are left), no distances: a deferred child is fetched and tested whatever the closest hit has become;
//   D  as C with the leaves of a node tested before its interior children are entered.
// Output per variant: node fetches, primitive tests, pushes and pops per ray.
// Build + run: scratch/wide_model.sh
#include <omp.h>

#include <array>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../raytracer_amd/csrc/host_bvh.h"

using namespace rt;

namespace {

struct WNode {
    int n = 0;
    int32_t ref[8];
    float lo[8][3], hi[8][3];
};

struct Wide {
    int W;
    bool octant;
    const HostBvh& bvh;
    double pad;
    std::vector<WNode> nodes;
    struct Slot { int32_t ref; Box box; };

    static double half_area(const Box& b) {
        const double x = b.hi.x - b.lo.x, y = b.hi.y - b.lo.y, z = b.hi.z - b.lo.z;
        return x * y + y * z + z * x;
    }
    std::vector<Slot> open(int32_t b) const {
        std::vector<Slot> s;
        auto kids = [&](int32_t n, Slot& l, Slot& r) {
            const DNode& nd = bvh.nodes[n];
            l = Slot{nd.child[0], Box{V3(nd.lo[0]), V3(nd.hi[0])}};
            r = Slot{nd.child[1], Box{V3(nd.lo[1]), V3(nd.hi[1])}};
        };
        Slot l, r;
        kids(b, l, r);
        s.push_back(l);
        s.push_back(r);
        while ((int)s.size() < W) {
            int pick = -1;
            double best = -1.0;
            for (size_t i = 0; i < s.size(); ++i)
                if (s[i].ref >= 0 && half_area(s[i].box) > best) { best = half_area(s[i].box); pick = (int)i; }
            if (pick < 0) break;
            kids(s[pick].ref, l, r);
            s[pick] = l;
            s.push_back(r);
        }
        return s;
    }
    int32_t emit(int32_t b) {
        std::vector<Slot> s = open(b);
        const int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        // the node's grid, as Collapser::quantise lays it out
        double org[3], step[3];
        for (int a = 0; a < 3; ++a) {
            double lo = 1e300, hi = -1e300;
            for (const Slot& c : s) { lo = std::min(lo, c.box.lo[a] - pad); hi = std::max(hi, c.box.hi[a] + pad); }
            org[a] = round_down(lo);
            int e = -126;
            const double ext = hi - org[a];
            if (ext > 0.0) { int ex; (void)std::frexp(ext / 255.0, &ex); e = ex; }
            while (std::ceil((hi - org[a]) / std::ldexp(1.0, e)) > 255.0) ++e;
            step[a] = std::ldexp(1.0, e);
        }
        std::array<int, 8> slot_of{};
        for (size_t c = 0; c < s.size(); ++c) slot_of[c] = (int)c;
        int n_slots = (int)s.size();
        if (octant) {   // greedy: the (child, slot) pair whose centre lies furthest along the slot's diagonal goes first
            Box all = s[0].box;
            for (const Slot& c : s) all = combine(all, c.box);
            const V3 mid = (all.lo + all.hi) * 0.5;
            bool child_done[8] = {}, slot_done[8] = {};
            for (size_t k = 0; k < s.size(); ++k) {
                double best = -1e300;
                int bc = -1, bs = -1;
                for (size_t c = 0; c < s.size(); ++c) {
                    if (child_done[c]) continue;
                    const V3 ctr = (s[c].box.lo + s[c].box.hi) * 0.5 - mid;
                    for (int sl = 0; sl < 8; ++sl) {
                        if (slot_done[sl]) continue;
                        const double v = (sl & 1 ? ctr.x : -ctr.x) + (sl & 2 ? ctr.y : -ctr.y) + (sl & 4 ? ctr.z : -ctr.z);
                        if (v > best) { best = v; bc = (int)c; bs = sl; }
                    }
                }
                child_done[bc] = slot_done[bs] = true;
                slot_of[bc] = bs;
            }
            n_slots = 8;
        }
        WNode nd;
        nd.n = n_slots;
        for (int k = 0; k < 8; ++k) {
            nd.ref[k] = kEmptyChild;
            for (int a = 0; a < 3; ++a) { nd.lo[k][a] = 1e30f; nd.hi[k][a] = -1e30f; }
        }
        for (size_t c = 0; c < s.size(); ++c) {
            const int k = slot_of[c];
            for (int a = 0; a < 3; ++a) {
                double ql = std::floor((std::max(s[c].box.lo[a] - pad, org[a]) - org[a]) / step[a]);
                double qh = std::ceil((s[c].box.hi[a] + pad - org[a]) / step[a]);
                ql = std::min(std::max(ql, 0.0), 255.0);
                qh = std::min(std::max(qh, 0.0), 255.0);
                nd.lo[k][a] = (float)(org[a] + ql * step[a]);
                nd.hi[k][a] = (float)(org[a] + qh * step[a]);
            }
            nd.ref[k] = s[c].ref;   // binary reference for now
        }
        nodes[me] = nd;
        for (int k = 0; k < 8; ++k)
            if (nd.ref[k] != kEmptyChild && nd.ref[k] >= 0) {
                const int32_t child = emit(nd.ref[k]);
                nodes[me].ref[k] = child;
            }
        return me;
    }
};

struct Sphere {
    double m[12];
};

struct Counters {
    double nodes = 0, prims = 0, pushes = 0, pops = 0, stale = 0, hits = 0, maxnodes = 0;
    void add(const Counters& o) {
        nodes += o.nodes; prims += o.prims; pushes += o.pushes; pops += o.pops; stale += o.stale; hits += o.hits;
        maxnodes = std::max(maxnodes, o.maxnodes);
    }
};

inline bool hit_sphere(const Sphere& s, const double o[3], const double d[3], double tmin, double& tmax) {
    double p[3], q[3];
    for (int r = 0; r < 3; ++r) {
        p[r] = s.m[r * 4] * o[0] + s.m[r * 4 + 1] * o[1] + s.m[r * 4 + 2] * o[2] + s.m[r * 4 + 3];
        q[r] = s.m[r * 4] * d[0] + s.m[r * 4 + 1] * d[1] + s.m[r * 4 + 2] * d[2];
    }
    const double a = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
    const double hb = p[0] * q[0] + p[1] * q[1] + p[2] * q[2];
    const double c = p[0] * p[0] + p[1] * p[1] + p[2] * p[2] - 1.0;
    const double disc = hb * hb - a * c;
    if (disc < 0.0) return false;
    const double sq = std::sqrt(disc);
    double t = (-hb - sq) / a;
    if (t <= tmin || t >= tmax) {
        t = (-hb + sq) / a;
        if (t <= tmin || t >= tmax) return false;
    }
    tmax = t;
    return true;
}

struct Tracer {
    const Wide& w;
    const std::vector<Sphere>& spheres;
    int variant;   // 0 sorted + distances, 1 octant mask, 2 octant mask with leaves first

    // entry distance of child k or a negative number for a miss
    inline float slab(const WNode& nd, int k, const float o[3], const float inv[3], float tmax) const {
        float t0 = 0.0f, t1 = tmax;
        for (int a = 0; a < 3; ++a) {
            float ta = (nd.lo[k][a] - o[a]) * inv[a], tb = (nd.hi[k][a] - o[a]) * inv[a];
            if (ta > tb) std::swap(ta, tb);
            if (ta > t0) t0 = ta;   // NaN drops out, as on the device
            if (tb < t1) t1 = tb;
        }
        return t0 <= t1 ? t0 : -1.0f;
    }
    void leaf(int32_t ref, const double o[3], const double d[3], double& tmax, Counters& c) const {
        const uint32_t u = ~(uint32_t)ref;
        const int first = (int)(u >> 1), cnt = (int)(u & 1) + 1;
        for (int i = 0; i < cnt; ++i) {
            c.prims += 1;
            if (hit_sphere(spheres[first + i], o, d, 1e-4, tmax)) c.hits += 0;
        }
    }
    Counters trace(const double o[3], const double d[3]) const {
        Counters c;
        double tmax = 1e5;
        float of[3], inv[3];
        for (int a = 0; a < 3; ++a) { of[a] = (float)o[a]; inv[a] = 1.0f / (float)d[a]; }
        const int oct = (d[0] < 0 ? 1 : 0) | (d[1] < 0 ? 2 : 0) | (d[2] < 0 ? 4 : 0);
        if (variant == 0) {
            struct E { int32_t ref; float t; };
            E stack[256];
            int sp = 0;
            int32_t cur = 0;
            for (;;) {
                if (cur >= 0) {
                    const WNode& nd = w.nodes[cur];
                    c.nodes += 1;
                    E h[8];
                    int nh = 0;
                    for (int k = 0; k < nd.n; ++k) {
                        if (nd.ref[k] == kEmptyChild) continue;
                        const float t = slab(nd, k, of, inv, (float)tmax * 1.000001f);
                        if (t >= 0.0f) h[nh++] = E{nd.ref[k], t};
                    }
                    std::stable_sort(h, h + nh, [](const E& a, const E& b) { return a.t < b.t; });
                    for (int k = nh - 1; k >= 1; --k) { stack[sp++] = h[k]; c.pushes += 1; }
                    if (nh > 0) { cur = h[0].ref; continue; }
                } else {
                    leaf(cur, o, d, tmax, c);
                }
                for (;;) {
                    if (sp == 0) { c.maxnodes = c.nodes; c.hits = tmax < 1e5; return c; }
                    const E e = stack[--sp];
                    c.pops += 1;
                    if ((double)e.t <= tmax) { cur = e.ref; break; }
                    c.stale += 1;
                }
            }
        }
        struct G { int32_t node; uint32_t mask; };   // mask bit p = the child in slot p ^ oct is still to visit
        G stack[256];
        int sp = 0;
        G cur{0, 0};
        bool fresh = true;   // cur.node has not been fetched yet
        for (;;) {
            if (fresh) {
                const WNode& nd = w.nodes[cur.node];
                c.nodes += 1;
                uint32_t mask = 0;
                for (int k = 0; k < 8; ++k) {
                    if (nd.ref[k] == kEmptyChild) continue;
                    if (slab(nd, k, of, inv, (float)tmax * 1.000001f) >= 0.0f) mask |= 1u << (k ^ oct);
                }
                if (variant == 2)
                    for (int p = 0; p < 8; ++p)
                        if ((mask >> p & 1) && nd.ref[p ^ oct] < 0) { leaf(nd.ref[p ^ oct], o, d, tmax, c); mask &= ~(1u << p); }
                cur.mask = mask;
                fresh = false;
            }
            if (cur.mask == 0) {
                if (sp == 0) { c.maxnodes = c.nodes; c.hits = tmax < 1e5; return c; }
                cur = stack[--sp];
                c.pops += 1;
                continue;
            }
            const int p = __builtin_ctz(cur.mask);
            cur.mask &= cur.mask - 1;
            const int32_t ref = w.nodes[cur.node].ref[p ^ oct];
            if (ref < 0) { leaf(ref, o, d, tmax, c); continue; }
            if (cur.mask) { stack[sp++] = cur; c.pushes += 1; }
            cur = G{ref, 0};
            fresh = true;
        }
    }
};

}  // namespace

extern "C" int wide_model(const RtSceneDesc* desc, int pixel_step, double* out /* [4 variants][7] */) {
    const int n = desc->n_shapes;
    std::vector<int32_t> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::vector<Box> boxes = traversal_boxes(*desc, order);
    HostBvh bvh = build_traversal_bvh(boxes, kMaxBinaryDepth);
    const double pad = node4_pad(bvh.root_box, 0.0);
    std::vector<Sphere> spheres(n);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 12; ++k) spheres[i].m[k] = desc->shapes[i].inv_transform[k];

    const RtCamera& cam = desc->camera;
    const V3 fwd = normalize(V3(cam.gaze_direction));
    const V3 right = normalize(cross(fwd, V3(cam.up_vector)));
    const V3 up = cross(right, fwd);
    const int W = cam.resolution_x, H = cam.resolution_y;
    const double sw = cam.sensor_width, sh = sw * H / W;

    const struct { int width; bool octant; int variant; const char* name; } variants[4] = {
        {4, false, 0, "A  4-wide, sorted, distances on the stack"},
        {8, false, 0, "B  8-wide, sorted, distances on the stack"},
        {8, true, 1, "C  8-wide, octant order, one mask per node"},
        {8, true, 2, "D  as C, leaves of a node first"},
    };
    for (int v = 0; v < 4; ++v) {
        Wide wide{variants[v].width, variants[v].octant, bvh, pad};
        wide.nodes.reserve(bvh.nodes.size());
        wide.emit(bvh.root_ref);
        double filled = 0;
        for (const WNode& nd : wide.nodes)
            for (int k = 0; k < 8; ++k) filled += nd.ref[k] != kEmptyChild;
        Tracer tr{wide, spheres, variants[v].variant};
        Counters total;
        double rays = 0;
#pragma omp parallel
        {
            Counters mine;
            double my_rays = 0;
#pragma omp for schedule(dynamic, 4)
            for (int y = 0; y < H; y += pixel_step)
                for (int x = 0; x < W; x += pixel_step) {
                    const double px = ((x + 0.5) / W - 0.5) * sw, py = (0.5 - (y + 0.5) / H) * sh;
                    const V3 dir = normalize(fwd * cam.focal_length + right * px + up * py);
                    const double o[3] = {cam.location[0], cam.location[1], cam.location[2]}, d[3] = {dir.x, dir.y, dir.z};
                    mine.add(tr.trace(o, d));
                    my_rays += 1;
                }
#pragma omp critical
            { total.add(mine); rays += my_rays; }
        }
        std::printf("%-44s nodes %7zu (%.2f children each)  per ray: node fetches %6.2f  primitive tests %5.2f  pushes %6.2f  pops %6.2f (stale %5.2f)  hit %.3f  longest ray %4.0f fetches\n",
                    variants[v].name, wide.nodes.size(), filled / wide.nodes.size(), total.nodes / rays, total.prims / rays, total.pushes / rays,
                    total.pops / rays, total.stale / rays, total.hits / rays, total.maxnodes);
        std::fflush(stdout);
        double* o = out + v * 7;
        o[0] = total.nodes / rays; o[1] = total.prims / rays; o[2] = total.pushes / rays; o[3] = total.pops / rays; o[4] = total.stale / rays;
        o[5] = (double)wide.nodes.size(); o[6] = total.maxnodes;
    }
    return 0;
}
