// TEST INFRASTRUCTURE ONLY — CPU oracle for the render hot path.  Nothing in the product (raytracer_amd/,
// librt_mi355x.so, rtcli) includes, links or calls this file; only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py do, and only as the checker / the reported CPU baseline.
//
// What it is: a plain recursive FP64 restatement of the reference's per-pixel algorithm
// (alexsykes8/raytracer, Code/main.cpp:281-322 and everything it calls), written against the POD scene of
// include/rt_mi355x.h so that the HIP path and the oracle can be fed identical inputs.  Every function cites the
// reference file:line it follows.  Structure deliberately mirrors the reference (pointer BVH with per-node
// boxes, left-then-right traversal, true recursion in ray_colour, duplicate visibility evaluation) rather
// than the device design, so that the two implementations fail independently.
//
// Pinning: tests/test_oracle_golden.py checks this oracle against images rendered here by the *real* reference
// (oracle/_ref/ref_raytracer = unmodified reference sources + oracle/ref_shim.h) and committed under
// tests/golden/ by oracle/gen_golden.py.
//
// Build: make -C oracle port   (g++ -O2 -fopenmp -ffp-contract=off, no GPU involved)

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "rt_mi355x.h"

namespace {

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

// ---------------------------------------------------------------- math (Code/utilities/vector3.h:15-110)
struct Vec {
    double x = 0, y = 0, z = 0;
    Vec() = default;
    Vec(double a, double b, double c) : x(a), y(b), z(c) {}
    explicit Vec(const double* p) : x(p[0]), y(p[1]), z(p[2]) {}
    Vec operator-() const { return {-x, -y, -z}; }
    Vec operator-(const Vec& o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vec operator+(const Vec& o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vec operator*(double t) const { return {x * t, y * t, z * t}; }
    Vec operator/(double t) const { return {x / t, y / t, z / t}; }
    double dot(const Vec& o) const { return x * o.x + y * o.y + z * o.z; }
    Vec cross(const Vec& o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
    double length() const { return std::sqrt(x * x + y * y + z * z); }
    Vec normalize() const {  // vector3.h:92-102
        double len = length();
        if (len > 1e-6) return {x / len, y / len, z / len};
        return {};
    }
};
inline Vec operator*(double t, const Vec& v) { return v * t; }
inline Vec cmul(const Vec& a, const Vec& b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }  // shading.h:38-40

struct Ray {  // Code/utilities/ray.h:14-33
    Vec origin, direction;
    double time = 0;
    Vec at(double t) const { return origin + (t * direction); }
};

// Code/utilities/matrix4x4.cpp:80-113
struct Mat {
    double m[4][4];
    Vec point(const Vec& v) const {
        double x = m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z + m[0][3];
        double y = m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z + m[1][3];
        double z = m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z + m[2][3];
        double w = m[3][0] * v.x + m[3][1] * v.y + m[3][2] * v.z + m[3][3];
        if (w != 0.0 && w != 1.0) return {x / w, y / w, z / w};
        return {x, y, z};
    }
    Vec dir(const Vec& v) const {
        return {m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
                m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z};
    }
    Mat transpose() const {
        Mat r;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) r.m[i][j] = m[j][i];
        return r;
    }
    static Mat from(const double* p) {
        Mat r;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) r.m[i][j] = p[i * 4 + j];
        return r;
    }
};

// ---------------------------------------------------------------- RNG (reference: random_utils.h:13-35, camera.cpp:13-24)
// D0 reproduces oracle/ref_shim.h mode "d0"; SEEDED is a per-(pixel,sample) splitmix64 stream (statistical use only).
struct Rng {
    bool d0 = true;
    uint64_t s = 0;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return d0 ? 0.5 : (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    Vec in_unit_sphere() {  // random_utils.h:29-35
        if (d0) return {};
        while (true) {
            Vec p(uniform() * 2 - 1, uniform() * 2 - 1, uniform() * 2 - 1);
            if (p.dot(p) < 1.0) return p;
        }
    }
    Vec in_unit_disk() {  // camera.cpp:13-24
        if (d0) return {};
        while (true) {
            Vec p(uniform() * 2.0 - 1.0, uniform() * 2.0 - 1.0, 0.0);
            if (p.dot(p) < 1.0) return p;
        }
    }
};

// ---------------------------------------------------------------- images
struct Tex {  // Code/utilities/Image.cpp
    int w = 0, h = 0;
    const uint8_t* rgb = nullptr;
    struct Px { unsigned char r, g, b; };
    Px get(int x, int y) const {  // Image.cpp:191-200 (callers clamp; out-of-range would throw there)
        size_t i = (static_cast<size_t>(y) * w + x) * 3;
        return {rgb[i], rgb[i + 1], rgb[i + 2]};
    }
    Px bilinear(double u, double v) const {  // Image.cpp:7-65: clamp, interpolate, truncate to 8 bit
        u = std::max(0.0, std::min(1.0, u));
        v = std::max(0.0, std::min(1.0, v));
        double px = u * (w - 1), py = v * (h - 1);
        int x0 = static_cast<int>(std::floor(px)), y0 = static_cast<int>(std::floor(py));
        int x1 = std::min(x0 + 1, w - 1), y1 = std::min(y0 + 1, h - 1);
        double dx = px - x0, dy = py - y0;
        Px c00 = get(x0, y0), c10 = get(x1, y0), c01 = get(x0, y1), c11 = get(x1, y1);
        double rt = (1.0 - dx) * c00.r + dx * c10.r, gt = (1.0 - dx) * c00.g + dx * c10.g, bt = (1.0 - dx) * c00.b + dx * c10.b;
        double rb = (1.0 - dx) * c01.r + dx * c11.r, gb = (1.0 - dx) * c01.g + dx * c11.g, bb = (1.0 - dx) * c01.b + dx * c11.b;
        return {static_cast<unsigned char>((1.0 - dy) * rt + dy * rb), static_cast<unsigned char>((1.0 - dy) * gt + dy * gb),
                static_cast<unsigned char>((1.0 - dy) * bt + dy * bb)};
    }
    double intensity_at(int x, int y) const {  // sphere.cpp:143-152
        x = std::min(std::max(x, 0), w - 1);
        y = std::min(std::max(y, 0), h - 1);
        Px p = get(x, y);
        return (p.r + p.g + p.b) / (3.0 * 255.0);
    }
};

struct Env {  // Code/environment/HDRImage.cpp:112-182
    int present = 0, w = 0, h = 0;
    const float* rgb = nullptr;
    Vec sample(double u, double v) const {
        if (!rgb || w <= 0 || h <= 0) return {};
        double U = u / (2.0 * M_PI), V = v / M_PI;
        double px = U * (w - 1), py = V * (h - 1);
        px = std::fmod(px, w);
        if (px < 0) px += w;
        py = std::clamp(py, 0.0, double(h - 1));
        int x0 = static_cast<int>(std::floor(px)), y0 = static_cast<int>(std::floor(py));
        int x1 = x0 + 1, y1 = y0 + 1;
        double dx = px - x0, dy = py - y0;
        x1 = x1 % w;
        if (x1 < 0) x1 += w;
        y1 = std::min(y1, h - 1);
        auto px_at = [&](int x, int y) {
            size_t i = (static_cast<size_t>(y) * w + x) * 3;
            return Vec(rgb[i], rgb[i + 1], rgb[i + 2]);
        };
        Vec c00 = px_at(x0, y0), c10 = px_at(x1, y0), c01 = px_at(x0, y1), c11 = px_at(x1, y1);
        Vec bottom = c00 * (1.0 - dx) + c10 * dx, top = c01 * (1.0 - dx) + c11 * dx;
        return bottom * (1.0 - dy) + top * dy;
    }
};

// ---------------------------------------------------------------- scene objects
struct Material {  // Code/shapes/material.h:17-51
    Vec ambient, diffuse, specular;
    double shininess, reflectivity, transparency, ior;
    const Tex* texture = nullptr;
    const Tex* bump = nullptr;
    bool metal = false;
};

struct Hit {  // Code/shapes/hittable.h:16-43
    double t = 0;
    Vec point, normal;
    const Material* mat = nullptr;
    double u = 0, v = 0;
    bool front_face = false;
    void set_face_normal(const Ray& r, const Vec& outward) {
        front_face = r.direction.dot(outward) < 0.0;
        normal = front_face ? outward : -outward;
    }
};

struct Box {  // Code/acceleration/aabb.{h,cpp}
    Vec lo, hi;
    bool hit(const Ray& r, double tmin, double tmax) const {  // aabb.cpp:4-38
        for (int a = 0; a < 3; ++a) {
            double d = a == 0 ? r.direction.x : (a == 1 ? r.direction.y : r.direction.z);
            double o = a == 0 ? r.origin.x : (a == 1 ? r.origin.y : r.origin.z);
            double mn = a == 0 ? lo.x : (a == 1 ? lo.y : lo.z), mx = a == 0 ? hi.x : (a == 1 ? hi.y : hi.z);
            double inv = 1.0 / d;
            double t0 = (mn - o) * inv, t1 = (mx - o) * inv;
            if (inv < 0.0) std::swap(t0, t1);
            tmin = t0 > tmin ? t0 : tmin;
            tmax = t1 < tmax ? t1 : tmax;
            if (tmax <= tmin) return false;
        }
        return true;
    }
    static Box join(const Box& a, const Box& b) {  // aabb.cpp:41-56
        return {{std::min(a.lo.x, b.lo.x), std::min(a.lo.y, b.lo.y), std::min(a.lo.z, b.lo.z)},
                {std::max(a.hi.x, b.hi.x), std::max(a.hi.y, b.hi.y), std::max(a.hi.z, b.hi.z)}};
    }
    void grow(const Vec& p) {  // aabb.h:24-31
        lo.x = std::min(lo.x, p.x); lo.y = std::min(lo.y, p.y); lo.z = std::min(lo.z, p.z);
        hi.x = std::max(hi.x, p.x); hi.y = std::max(hi.y, p.y); hi.z = std::max(hi.z, p.z);
    }
};

struct Settings {  // what Scene + Config hand to the hot loop
    RtRenderParams p;
    double shutter = 0;
};

struct Counters {
    uint64_t rays = 0, boxes = 0, prims = 0;
};
thread_local Counters* tl_counters = nullptr;

struct Object {
    virtual ~Object() {}
    virtual bool intersect(const Ray& r, double tmin, double tmax, Hit& rec) const = 0;
    virtual Box bounds() const = 0;
};

// transformed_shape.h:13-78
struct Transformed : Object {
    Mat fwd, inv, inv_t;
    Material mat;
    Vec velocity;
    double shutter = 0;
    const Settings* cfg = nullptr;
    Box bounds_of(const Vec& lmin, const Vec& lmax) const {
        const double inf = std::numeric_limits<double>::infinity();
        Box b{{inf, inf, inf}, {-inf, -inf, -inf}};
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) b.grow(fwd.point(Vec(i ? lmax.x : lmin.x, j ? lmax.y : lmin.y, k ? lmax.z : lmin.z)));
        Vec d = velocity * shutter;
        return Box::join(b, Box{b.lo + d, b.hi + d});
    }
    // sphere.cpp:116-167 == cube.cpp:178-229: finite-difference bump on nearest texels, tangent from world Y.
    Vec bumped(const Vec& N, double u, double v) const {
        Vec Y(0, 1, 0), T;
        if (std::abs(N.dot(Y)) > 0.999) T = Vec(1, 0, 0);
        else T = Y.cross(N).normalize();
        Vec B = N.cross(T).normalize();
        int w = mat.bump->w, h = mat.bump->h;
        int x = static_cast<int>(u * (w - 1)), y = static_cast<int>((1.0 - v) * (h - 1));
        double hc = mat.bump->intensity_at(x, y), hu = mat.bump->intensity_at(x + 1, y), hv = mat.bump->intensity_at(x, y + 1);
        double bu = (hu - hc) * w, bv = (hv - hc) * h;
        return (N + (T * bu + B * bv) * 0.0075).normalize();
    }
};

void sphere_uv(const Vec& p, double& u, double& v) {  // sphere.cpp:28-41
    double theta = std::asin(p.y);
    double phi = std::atan2(-p.z, p.x) + M_PI;
    u = phi / (2.0 * M_PI);
    v = (theta + M_PI / 2.0) / M_PI;
}

struct Sphere : Transformed {  // sphere.cpp:44-172
    Box bounds() const override { return bounds_of(Vec(-1, -1, -1), Vec(1, 1, 1)); }
    bool intersect(const Ray& ray, double tmin, double tmax, Hit& rec) const override {
        if (tl_counters) tl_counters->prims++;
        Vec o0 = ray.origin - velocity * ray.time;
        Vec oo = inv.point(o0), od = inv.dir(ray.direction);
        double a = od.dot(od), b = 2.0 * oo.dot(od), c = oo.dot(oo) - 1.0;
        double disc = b * b - 4 * a * c;
        if (disc < 0) return false;
        double root = (-b - std::sqrt(disc)) / (2.0 * a);
        if (root < tmin || root > tmax) {
            root = (-b + std::sqrt(disc)) / (2.0 * a);
            if (root < tmin || root > tmax) return false;
        }
        rec.t = root;
        rec.point = ray.at(root);
        Vec op = oo + (root * od);
        Vec on = op - Vec(0, 0, 0);
        Vec outward = inv_t.point(on).normalize();
        rec.set_face_normal(ray, outward);
        rec.mat = &mat;
        sphere_uv(on.normalize(), rec.u, rec.v);
        if (mat.bump) outward = bumped(outward, rec.u, rec.v);
        rec.set_face_normal(ray, outward);
        return true;
    }
};

// cube.cpp:102-175 and complex_cube.cpp:49-118 share the face / atlas logic (the latter clamps raw uv).
void cube_face_uv(const Vec& p, bool clamp_raw, Vec& n, double& u, double& v) {
    double ax = std::abs(p.x), ay = std::abs(p.y), az = std::abs(p.z);
    int axis;
    n = Vec(0, 0, 0);
    if (ax >= ay && ax >= az) { axis = 0; n.x = (p.x > 0) ? 1.0 : -1.0; }
    else if (ay >= ax && ay >= az) { axis = 1; n.y = (p.y > 0) ? 1.0 : -1.0; }
    else { axis = 2; n.z = (p.z > 0) ? 1.0 : -1.0; }
    double ru, rv;
    if (axis == 0) { ru = (p.y * (n.x > 0 ? -1 : 1) + 1.0) * 0.5; rv = (p.z + 1.0) * 0.5; }
    else if (axis == 1) { ru = (p.x * (n.y > 0 ? 1 : -1) + 1.0) * 0.5; rv = (p.z + 1.0) * 0.5; }
    else { ru = (p.x + 1.0) * 0.5; rv = (p.y + 1.0) * 0.5; }
    if (clamp_raw) {
        ru = std::max(0.0, std::min(1.0, ru));
        rv = std::max(0.0, std::min(1.0, rv));
    }
    double uo, vo;
    if (axis == 2) { uo = 1.0; vo = (n.z > 0) ? 2.0 : 0.0; }
    else if (axis == 1) { if (n.y > 0) { uo = 1.0; vo = 1.0; } else { uo = 3.0; vo = 1.0; } }
    else { if (n.x > 0) { uo = 2.0; vo = 1.0; } else { uo = 0.0; vo = 1.0; } }
    u = (ru + uo) * 0.25;
    v = (rv + vo) * (1.0 / 3.0);
}

struct Cube : Transformed {  // cube.cpp:25-232
    Box bounds() const override { return bounds_of(Vec(-1, -1, -1), Vec(1, 1, 1)); }
    bool intersect(const Ray& ray, double tmin, double tmax, Hit& rec) const override {
        if (tl_counters) tl_counters->prims++;
        Vec o0 = ray.origin - velocity * ray.time;
        Vec oo = inv.point(o0), od = inv.dir(ray.direction);
        const double inf = std::numeric_limits<double>::infinity();
        double tn = -inf, tf = inf;
        for (int i = 0; i < 3; ++i) {
            double o = i == 0 ? oo.x : (i == 1 ? oo.y : oo.z), d = i == 0 ? od.x : (i == 1 ? od.y : od.z);
            double t0, t1;
            if (d == 0.0) {
                if (o < -1.0 || o > 1.0) return false;
                t0 = -inf;
                t1 = inf;
            } else {
                double id = 1.0 / d;
                t0 = (-1.0 - o) * id;
                t1 = (1.0 - o) * id;
                if (id < 0.0) std::swap(t0, t1);
            }
            if (t0 > tn) tn = t0;
            if (t1 < tf) tf = t1;
            if (tn > tf) return false;
            if (tf < 0.0) return false;
        }
        double th = tn;
        if (th < tmin || th > tmax) {
            th = tf;
            if (th < tmin || th > tmax) return false;
        }
        rec.t = th;
        rec.point = ray.at(th);
        rec.mat = &mat;
        Vec p = oo + od * th, on;
        double u, v;
        cube_face_uv(p, false, on, u, v);
        Vec outward = inv_t.point(on).normalize();
        rec.set_face_normal(ray, outward);
        rec.u = u;
        rec.v = v;
        if (mat.bump) rec.set_face_normal(ray, bumped(outward, u, v));
        return true;
    }
};

struct Quad : Object {  // plane.cpp
    Vec v0, e1, e2, w0, f1, f2, n, velocity;
    Material mat;
    double shutter = 0;
    const Settings* cfg = nullptr;
    Quad(const Vec& c0, const Vec& c1, const Vec& c2, const Vec& c3) {  // plane.cpp:49-74
        v0 = c0; e1 = c1 - c0; e2 = c2 - c0;
        w0 = c1; f1 = c3 - c1; f2 = c2 - c1;
        n = e1.cross(e2).normalize();
    }
    Box bounds() const override {  // plane.cpp:11-46
        Vec a = v0, b = v0 + e1, c = v0 + e2, d = w0 + f1;
        Box bx{a, a};
        bx.grow(b); bx.grow(c); bx.grow(d);
        const double eps = 1e-4;
        bx.lo = bx.lo - Vec(eps, eps, eps);
        bx.hi = bx.hi + Vec(eps, eps, eps);
        Vec disp = velocity * shutter;
        return Box::join(bx, Box{bx.lo + disp, bx.hi + disp});
    }
    bool tri(const Ray& ray, double tmin, double tmax, const Vec& a, const Vec& ea, const Vec& eb, double& ot, double& ou, double& ov) const {
        const double EPS = cfg->p.eps_triangle;  // plane.cpp:83
        Vec h = ray.direction.cross(eb);
        double det = ea.dot(h);
        if (det > -EPS && det < EPS) return false;
        double f = 1.0 / det;
        Vec s = ray.origin - a;
        double u = f * s.dot(h);
        if (u < 0.0 || u > 1.0) return false;
        Vec q = s.cross(ea);
        double v = f * ray.direction.dot(q);
        if (v < 0.0 || u + v > 1.0) return false;
        ot = f * eb.dot(q);
        if (ot > tmin && ot < tmax) { ou = u; ov = v; return true; }
        return false;
    }
    bool intersect(const Ray& ray, double tmin, double tmax, Hit& rec) const override {  // plane.cpp:137-244
        if (tl_counters) tl_counters->prims++;
        Ray r0{ray.origin - velocity * ray.time, ray.direction, ray.time};
        double t1, u1, v1, t2, u2, v2;
        bool h1 = tri(r0, tmin, tmax, v0, e1, e2, t1, u1, v1);
        bool h2 = tri(r0, tmin, tmax, w0, f1, f2, t2, u2, v2);
        if (!h1 && !h2) return false;
        bool first = (h1 && h2) ? (t1 < t2) : h1;
        double th = first ? t1 : t2;
        rec.t = th;
        rec.point = ray.at(th);
        rec.set_face_normal(ray, n);
        Vec outward = n;
        rec.mat = &mat;
        if (first) { rec.u = u1; rec.v = v1; } else { rec.u = 1.0 - v2; rec.v = u2 + v2; }
        if (mat.bump) {  // plane.cpp:199-239
            Vec T = e1.normalize(), B = e2.normalize();
            int w = mat.bump->w, h = mat.bump->h;
            auto val = [&](double u, double v) {
                Tex::Px p = mat.bump->bilinear(u, v);
                return (p.r + p.g + p.b) / (3.0 * 255.0);
            };
            double sx = 1.0 / w, sy = 1.0 / h;
            double hc = val(rec.u, rec.v), hu = val(rec.u + sx, rec.v), hv = val(rec.u, rec.v + sy);
            double bu = (hu - hc) * w, bv = (hv - hc) * h;
            bu = std::max(-100.0, std::min(100.0, bu));
            bv = std::max(-100.0, std::min(100.0, bv));
            outward = (n + (T * bu + B * bv) * 0.0075).normalize();
        }
        rec.set_face_normal(ray, outward);
        return true;
    }
};

struct ComplexSphere : Transformed {  // complex_sphere.cpp
    double D = 0;
    Box bounds() const override { double r = 1.0 + D; return bounds_of(Vec(-r, -r, -r), Vec(r, r, r)); }
    double sdf(const Vec& q) const {  // complex_sphere.cpp:124-142
        double ql = q.length();
        Vec qu = q / ql;
        double u, v;
        sphere_uv(qu, u, v);
        double disp = 0.0;
        if (mat.bump) {
            Tex::Px p = mat.bump->bilinear(u, 1.0 - v);
            disp = ((p.r + p.g + p.b) / (3.0 * 255.0)) * D;
        }
        return ql - (1.0 + disp);
    }
    bool intersect(const Ray& ray, double tmin, double tmax, Hit& rec) const override {
        if (tl_counters) tl_counters->prims++;
        Vec o0 = ray.origin - velocity * ray.time;
        Vec oo = inv.point(o0), od = inv.dir(ray.direction);
        double max_r = 1.0 + D;
        double a = od.dot(od), b = 2.0 * oo.dot(od), c = oo.dot(oo) - max_r * max_r;
        double disc = b * b - 4 * a * c;
        if (disc < 0) return false;
        double te = (-b - std::sqrt(disc)) / (2.0 * a), tx = (-b + std::sqrt(disc)) / (2.0 * a);
        if (tx < tmin || te > tmax) return false;
        double t = std::max(te, tmin), tl = std::min(tx, tmax);
        const int STEPS = cfg->p.ray_march_steps;
        const double EPS = cfg->p.eps_march;
        for (int i = 0; i < STEPS; ++i) {
            if (t > tl) break;
            Vec p = oo + od * t;
            double dc = p.length();
            Vec pu = p / dc;
            double u, v;
            sphere_uv(pu, u, v);
            double disp = 0.0;
            if (mat.bump) {
                Tex::Px px = mat.bump->bilinear(u, 1.0 - v);
                disp = ((px.r + px.g + px.b) / (3.0 * 255.0)) * D;
            }
            double ds = dc - (1.0 + disp);
            if (ds < EPS) {
                rec.t = t;
                rec.point = ray.at(t);
                rec.mat = &mat;
                rec.u = u;
                rec.v = v;
                double e = 0.005;
                double dx = sdf(p + Vec(e, 0, 0)) - sdf(p - Vec(e, 0, 0));
                double dy = sdf(p + Vec(0, e, 0)) - sdf(p - Vec(0, e, 0));
                double dz = sdf(p + Vec(0, 0, e)) - sdf(p - Vec(0, 0, e));
                Vec ln = Vec(dx, dy, dz).normalize();
                rec.set_face_normal(ray, inv_t.point(ln).normalize());
                return true;
            }
            t += std::max(ds * 0.5, EPS);
        }
        return false;
    }
};

double box_sdf(const Vec& p, const Vec& half) {  // complex_cube.cpp:34-46, complex_plane.cpp:78-92
    Vec d = Vec(std::abs(p.x), std::abs(p.y), std::abs(p.z)) - half;
    double inside = std::min(std::max(d.x, std::max(d.y, d.z)), 0.0);
    double outside = Vec(std::max(d.x, 0.0), std::max(d.y, 0.0), std::max(d.z, 0.0)).length();
    return inside + outside;
}

// Slab entry/exit shared by complex_cube.cpp:130-160 and complex_plane.cpp:123-158.
bool slab_range(const Vec& oo, const Vec& od, const Vec& bound, double& tn, double& tf) {
    const double inf = std::numeric_limits<double>::infinity();
    tn = -inf;
    tf = inf;
    for (int i = 0; i < 3; ++i) {
        double o = i == 0 ? oo.x : (i == 1 ? oo.y : oo.z), d = i == 0 ? od.x : (i == 1 ? od.y : od.z);
        double bv = i == 0 ? bound.x : (i == 1 ? bound.y : bound.z);
        if (d == 0.0) {
            if (o < -bv || o > bv) return false;
        } else {
            double t0 = (-bv - o) / d, t1 = (bv - o) / d;
            if (d < 0.0) std::swap(t0, t1);
            if (t0 > tn) tn = t0;
            if (t1 < tf) tf = t1;
        }
    }
    return !(tn > tf || tf < 0.0);
}

struct ComplexCube : Transformed {  // complex_cube.cpp
    double D = 0;
    Box bounds() const override { double e = 1.0 + D; return bounds_of(Vec(-e, -e, -e), Vec(e, e, e)); }
    double disp_at(double u, double v) const {  // complex_cube.cpp:189-209 (nearest texel)
        if (!mat.bump) return 0.0;
        int w = mat.bump->w, h = mat.bump->h;
        int x = static_cast<int>(u * (w - 1)), y = static_cast<int>((1.0 - v) * (h - 1));
        x = std::max(0, std::min(x, w - 1));
        y = std::max(0, std::min(y, h - 1));
        Tex::Px p = mat.bump->get(x, y);
        return ((p.r + p.g + p.b) / (3.0 * 255.0)) * D;
    }
    double sdf(const Vec& q) const {
        double u, v;
        Vec n;
        cube_face_uv(q, true, n, u, v);
        return box_sdf(q, Vec(1, 1, 1)) - disp_at(u, v);
    }
    bool intersect(const Ray& ray, double tmin, double tmax, Hit& rec) const override {
        if (tl_counters) tl_counters->prims++;
        Vec o0 = ray.origin - velocity * ray.time;
        Vec oo = inv.point(o0), od = inv.dir(ray.direction);
        double bound = 1.0 + D, tn, tf;
        if (!slab_range(oo, od, Vec(bound, bound, bound), tn, tf)) return false;
        double t = std::max(tn, tmin), tl = std::min(tf, tmax);
        const int STEPS = cfg->p.ray_march_steps;
        const double EPS = cfg->p.eps_march;
        for (int i = 0; i < STEPS; ++i) {
            if (t > tl) break;
            Vec p = oo + od * t;
            double base = box_sdf(p, Vec(1, 1, 1));
            double u, v;
            Vec nd;
            cube_face_uv(p, true, nd, u, v);
            double ds = base - disp_at(u, v);
            if (ds < EPS) {
                rec.t = t;
                rec.point = ray.at(t);
                rec.mat = &mat;
                double d = 0.005;
                double gx = sdf(p + Vec(d, 0, 0)) - sdf(p - Vec(d, 0, 0));
                double gy = sdf(p + Vec(0, d, 0)) - sdf(p - Vec(0, d, 0));
                double gz = sdf(p + Vec(0, 0, d)) - sdf(p - Vec(0, 0, d));
                Vec ln = Vec(gx, gy, gz).normalize();
                rec.set_face_normal(ray, inv_t.point(ln).normalize());
                rec.u = u;
                rec.v = v;
                return true;
            }
            t += std::max(ds * 0.6, EPS);
        }
        return false;
    }
};

struct ComplexPlane : Transformed {  // complex_plane.cpp
    double D = 0;
    Box bounds() const override {  // complex_plane.cpp:24-75
        const double inf = std::numeric_limits<double>::infinity();
        double zb = D + 0.01;
        Box b{{inf, inf, inf}, {-inf, -inf, -inf}};
        bool moving = velocity.length() > 1e-6;
        for (int i = 0; i < 8; ++i) {
            Vec p = fwd.point(Vec((i & 1) ? 1.0 : -1.0, (i & 2) ? 1.0 : -1.0, (i & 4) ? zb : -zb));
            b.grow(p);
            if (moving) b.grow(p + velocity * shutter);  // reference reads an uninitialised m_shutter_time here
        }
        return b;
    }
    double disp_at(double u, double v) const {  // complex_plane.cpp:170-178
        if (!mat.bump) return 0.0;
        Tex::Px p = mat.bump->bilinear(u, 1.0 - v);
        return ((p.r + p.g + p.b) / (3.0 * 255.0)) * D;
    }
    static void uv_of(const Vec& p, double& u, double& v) {  // complex_plane.cpp:95-109
        u = (p.x + 1.0) * 0.5;
        v = (p.y + 1.0) * 0.5;
        u = std::max(0.0, std::min(1.0, u));
        v = std::max(0.0, std::min(1.0, v));
    }
    double sdf(const Vec& q) const {
        double u, v;
        uv_of(q, u, v);
        return box_sdf(q, Vec(1.0, 1.0, 0.001)) - disp_at(u, v);
    }
    bool intersect(const Ray& ray, double tmin, double tmax, Hit& rec) const override {
        if (tl_counters) tl_counters->prims++;
        Vec o0 = ray.origin - velocity * ray.time;
        Vec oo = inv.point(o0), od = inv.dir(ray.direction);
        double scale = od.length();
        double tn, tf;
        if (!slab_range(oo, od, Vec(1.0, 1.0, D + 0.01), tn, tf)) return false;
        double t = std::max(tn, tmin), tl = std::min(tf, tmax);
        const int STEPS = cfg->p.ray_march_steps;
        const double EPS = cfg->p.eps_march, MULT = cfg->p.step_multiplier;
        for (int i = 0; i < STEPS; ++i) {
            if (t > tl) break;
            Vec p = oo + od * t;
            double base = box_sdf(p, Vec(1.0, 1.0, 0.001));
            double u, v;
            uv_of(p, u, v);
            double ds = base - disp_at(u, v);
            if (ds < EPS) {
                rec.t = t;
                rec.point = ray.at(t);
                rec.mat = &mat;
                rec.u = u;
                rec.v = v;
                double d = 0.005;
                double gx = sdf(p + Vec(d, 0, 0)) - sdf(p - Vec(d, 0, 0));
                double gy = sdf(p + Vec(0, d, 0)) - sdf(p - Vec(0, d, 0));
                double gz = sdf(p + Vec(0, 0, d)) - sdf(p - Vec(0, 0, d));
                Vec ln = Vec(gx, gy, gz).normalize();
                rec.set_face_normal(ray, inv_t.point(ln).normalize());
                return true;
            }
            double step_world = ds / scale;
            t += std::max(step_world * MULT, EPS);
        }
        return false;
    }
};

// ---------------------------------------------------------------- BVH (Code/acceleration/bvh.cpp)
struct Node : Object {
    const Object* left = nullptr;
    const Object* right = nullptr;
    Box box;
    Box bounds() const override { return box; }
    bool intersect(const Ray& r, double tmin, double tmax, Hit& rec) const override {  // bvh.cpp:101-114
        if (tl_counters) tl_counters->boxes++;
        if (!box.hit(r, tmin, tmax)) return false;
        bool hl = left->intersect(r, tmin, tmax, rec);
        bool hr = right->intersect(r, tmin, hl ? rec.t : tmax, rec);
        return hl || hr;
    }
};

struct World {
    std::vector<std::unique_ptr<Object>> shapes;   // scene-file order
    std::vector<std::unique_ptr<Node>> nodes;
    std::vector<Box> boxes;                        // per shape, cached (the reference recomputes them)
    const Object* root = nullptr;                  // BVH root or null
    std::vector<Tex> textures;
    std::vector<RtLight> lights;
    Env env;
    Settings cfg;
    // camera (camera.cpp:27-86)
    Vec cam_loc, cam_u, cam_v, cam_w;
    double focal = 0, sensor_w = 0, sensor_h = 0, aperture = 0, focal_dist = 0;
    int W = 0, H = 0;

    // bvh.cpp:32-92 over indices (sort keys are the cached box centres, bvh.cpp:14-19)
    const Object* build(std::vector<int>& idx, size_t start, size_t end) {
        auto node = std::make_unique<Node>();
        Box span = boxes[idx[start]];
        for (size_t i = start + 1; i < end; ++i) span = Box::join(span, boxes[idx[i]]);
        Vec ext = span.hi - span.lo;
        int axis = 0;
        if (ext.y > ext.x && ext.y > ext.z) axis = 1;
        else if (ext.z > ext.x && ext.z > ext.y) axis = 2;
        auto key = [&](int i) {
            Vec c = (boxes[i].lo + boxes[i].hi) * 0.5;
            return axis == 0 ? c.x : (axis == 1 ? c.y : c.z);
        };
        auto cmp = [&](int a, int b) { return key(a) < key(b); };
        size_t n = end - start;
        Box bl, br;
        if (n == 1) {
            node->left = node->right = shapes[idx[start]].get();
            bl = br = boxes[idx[start]];
        } else if (n == 2) {
            int a = idx[start], b = idx[start + 1];
            if (!cmp(a, b)) std::swap(a, b);
            node->left = shapes[a].get();
            node->right = shapes[b].get();
            bl = boxes[a];
            br = boxes[b];
        } else {
            size_t mid = start + n / 2;
            std::nth_element(idx.begin() + start, idx.begin() + mid, idx.begin() + end, cmp);
            node->left = build(idx, start, mid);
            node->right = build(idx, mid, end);
            bl = node->left->bounds();
            br = node->right->bounds();
        }
        node->box = Box::join(bl, br);
        const Object* ret = node.get();
        nodes.push_back(std::move(node));
        return ret;
    }

    // hittable_list.cpp:8-33.  With the BVH the list holds only the root (scene.cpp:111-112).
    bool intersect(const Ray& r, double tmin, double tmax, Hit& rec) const {
        if (tl_counters) tl_counters->rays++;
        if (cfg.p.use_bvh) {
            if (!root) return false;
            Hit tmp;
            if (root->intersect(r, tmin, tmax, tmp)) { rec = tmp; return true; }
            return false;
        }
        Hit tmp;
        bool any = false;
        double closest = tmax;
        for (const auto& o : shapes)
            if (o->intersect(r, tmin, closest, tmp)) { any = true; closest = tmp.t; rec = tmp; }
        return any;
    }

    Ray generate_ray(float px, float py, double time, Rng& rng) const {  // camera.cpp:89-135
        double uc = (0.5 - px) * sensor_w, vc = (py - 0.5) * sensor_h;
        Vec dir = ((focal * cam_w) + (uc * cam_u) + (vc * cam_v)).normalize();
        if (aperture <= 0.0) return Ray{cam_loc, dir, time};
        Vec focal_point = cam_loc + dir * focal_dist;
        Vec disk = rng.in_unit_disk() * aperture;
        Vec origin = cam_loc + (cam_u * disk.x + cam_v * disk.y);
        return Ray{origin, (focal_point - origin).normalize(), time};
    }
};

// ---------------------------------------------------------------- shading (Code/utilities/shading.h)
double shadow_schlick(double cos_i, double n1, double n2) {  // shading.h:24-28
    double r0 = ((n1 - n2) / (n1 + n2));
    r0 = r0 * r0;
    return r0 + (1.0 - r0) * std::pow((1.0 - cos_i), 5);
}

Vec shadow_transmission(const World& w, const Ray& sr, double dist) {  // shading.h:42-75
    Vec T(1.0, 1.0, 1.0);
    Hit rec;
    if (w.intersect(sr, 0.001, dist - 0.001, rec)) {
        if (rec.mat->transparency > 0.0) {
            double n1, n2;
            if (rec.front_face) { n1 = 1.0; n2 = rec.mat->ior; } else { n1 = rec.mat->ior; n2 = 1.0; }
            double eta = n1 / n2;
            double cos_i = -sr.direction.dot(rec.normal);
            double sin2 = eta * eta * (1.0 - cos_i * cos_i);
            if (sin2 > 1.0) return {};
            double factor = 1.0 - shadow_schlick(cos_i, n1, n2);
            T = cmul(T, rec.mat->diffuse) * factor;
            if (T.length() < 0.001) return {};
            Ray next{rec.point + sr.direction * 0.001, sr.direction, sr.time};
            return cmul(T, shadow_transmission(w, next, dist - rec.t));
        }
        return {};
    }
    return T;
}

Vec light_visibility(const World& w, const RtLight& L, const Vec& P, const Vec& N, double time, Rng& rng) {  // shading.h:79-101
    if (!w.cfg.p.shadows) return {1.0, 1.0, 1.0};
    int samples = w.cfg.p.shadow_samples;
    double eps = w.cfg.p.eps_shadow;
    Vec acc;
    for (int i = 0; i < samples; ++i) {
        Vec lp(L.position);
        if (L.radius != 0.0) lp = lp + rng.in_unit_sphere().normalize() * L.radius;  // shading.h:31-36
        Vec d = lp - P;
        double dist = d.length();
        d = d.normalize();
        Ray sr{P + N * eps, d, time};
        acc = acc + shadow_transmission(w, sr, dist);
    }
    return acc * (1.0 / samples);
}

Vec local_ad(const World& w, const Hit& rec, double time, Rng& rng) {  // shading.h:104-175
    const Material& m = *rec.mat;
    Vec base;
    if (m.texture) {
        int tw = m.texture->w, th = m.texture->h;
        int x = static_cast<int>(rec.u * (tw - 1)), y = static_cast<int>((1.0 - rec.v) * (th - 1));
        x = std::min(std::max(x, 0), tw - 1);
        y = std::min(std::max(y, 0), th - 1);
        Tex::Px p = m.texture->get(x, y);
        base = Vec(p.r / 255.0, p.g / 255.0, p.b / 255.0);
    } else {
        base = m.diffuse;
    }
    Vec amb(w.cfg.p.ambient_light);
    Vec colour = cmul(cmul(m.ambient, amb), base);
    const Vec P = rec.point, N = rec.normal.normalize();
    double exposure = w.cfg.p.exposure;
    for (const RtLight& L : w.lights) {
        Vec vis = light_visibility(w, L, P, N, time, rng);
        if (vis.x > 0 || vis.y > 0 || vis.z > 0) {
            Vec Lr = Vec(L.position) - P;
            double falloff = 1.0 / Lr.dot(Lr);
            Vec Ld = Lr.normalize();
            Vec I = Vec(L.intensity) * falloff * exposure;
            double ldn = std::max(0.0, Ld.dot(N));
            Vec diff = cmul(base, I) * ldn;
            diff = diff * (1.0 - m.transparency);
            colour = colour + cmul(diff, vis);
        }
    }
    return colour;
}

Vec specular(const World& w, const Hit& rec, const Ray& view, Rng& rng) {  // shading.h:181-217
    const Vec P = rec.point, N = rec.normal.normalize();
    const Vec V = (view.origin - P).normalize();
    const Material& m = *rec.mat;
    double exposure = w.cfg.p.exposure;
    Vec out;
    for (const RtLight& L : w.lights) {
        Vec vis = light_visibility(w, L, P, N, view.time, rng);
        if (vis.length() > 0) {
            Vec Lr = Vec(L.position) - P;
            double falloff = 1.0 / Lr.dot(Lr);
            Vec Ld = Lr.normalize();
            Vec Hv = (Ld + V).normalize();
            Vec I = Vec(L.intensity) * falloff * exposure;
            double hdn = std::max(0.0, Hv.dot(N));
            Vec part = cmul(m.specular, I) * std::exp(m.shininess * std::log(hdn));  // fast_pow, shading.h:177
            out = out + cmul(part, vis);
        }
    }
    return out;
}

// ---------------------------------------------------------------- integrator (Code/utilities/tracer.h)
Vec reflect(const Vec& V, const Vec& N) { return V - 2 * V.dot(N) * N; }  // tracer.h:57-60

double schlick(double cos_i, double n1, double n2) {  // tracer.h:88-97
    double r0 = ((n1 - n2) / (n1 + n2)) * ((n1 - n2) / (n1 + n2));
    double x = 1.0 - cos_i;
    return r0 + (1.0 - r0) * (x * x * x * x * x);
}

bool refraction(const Vec& V, const Vec& N, double ior, bool front, Vec& out_dir, double& reflect_prob, bool fresnel) {  // tracer.h:100-143
    double n1 = front ? 1.0 : ior, n2 = front ? ior : 1.0;
    double eta = n1 / n2;
    double cos_i = -V.dot(N);
    double sin2 = eta * eta * (1.0 - cos_i * cos_i);
    if (sin2 > 1.0) { reflect_prob = 1.0; return false; }
    double cos_t = std::sqrt(1.0 - sin2);
    out_dir = (eta * V) + (eta * cos_i - cos_t) * N;
    reflect_prob = fresnel ? schlick(cos_i, n1, n2) : 0.0;
    return true;
}

Vec ray_colour(const World& w, const Ray& r, int depth, Rng& rng) {  // tracer.h:147-295
    if (depth <= 0) return {};
    const RtRenderParams& P = w.cfg.p;
    double eps = P.eps_scene;
    Hit rec;
    if (w.intersect(r, eps, 100000.0, rec)) {
        if (P.normals) return (rec.normal + Vec(1, 1, 1)) * 0.5;
        Vec ad = local_ad(w, rec, r.time, rng);
        Vec spec = specular(w, rec, r, rng);
        Vec refl_col, refr_col;
        const Material& m = *rec.mat;
        bool transparent = m.transparency > 0;
        bool has_refl = m.reflectivity > 0;
        if (transparent && P.fresnel) has_refl = true;
        if (has_refl) {
            int samples = P.glossy_samples;
            if (depth < P.max_bounces) samples = 1;
            double rough = 1.0 / std::sqrt(m.shininess);
            Vec V = r.direction.normalize();
            Vec perfect = reflect(V, rec.normal).normalize();
            if (samples > 0) {
                for (int i = 0; i < samples; ++i) {
                    Vec off = rng.in_unit_sphere() * rough;
                    Vec target = (perfect + off).normalize();
                    if (target.dot(rec.normal) > 0) {
                        Ray rr{rec.point + rec.normal * eps, target, r.time};
                        refl_col = refl_col + ray_colour(w, rr, depth - 1, rng);
                    }
                }
                refl_col = refl_col * (1.0 / samples);
            } else {
                Ray rr{rec.point + rec.normal * eps, perfect, r.time};
                refl_col = ray_colour(w, rr, depth - 1, rng);
            }
            if (m.metal) refl_col = cmul(refl_col, m.diffuse);
        }
        double reflect_prob = m.reflectivity, transmit_prob = m.transparency;
        if (transparent) {
            Vec rdir;
            double fr = 0.0;
            Vec Vin = r.direction.normalize(), Nh = rec.normal.normalize();
            bool ok = refraction(Vin, Nh, m.ior, rec.front_face, rdir, fr, P.fresnel != 0);
            if (ok) {
                Ray tr{rec.point, rdir.normalize(), r.time};
                refr_col = cmul(ray_colour(w, tr, depth - 1, rng), m.diffuse);
                if (P.fresnel) { reflect_prob = fr; transmit_prob = 1.0 - reflect_prob; }
            } else {
                transmit_prob = 0.0;
                reflect_prob = 1.0;
                if (!has_refl) {
                    Ray rr{rec.point + Nh * eps, reflect(Vin, Nh).normalize(), r.time};
                    refl_col = ray_colour(w, rr, depth - 1, rng);
                }
            }
        }
        if (transparent) return (refl_col * reflect_prob) + (refr_col * transmit_prob) + spec;
        return ad * (1.0 - m.reflectivity) + refl_col * m.reflectivity + spec;
    }
    if (w.env.present) {  // tracer.h:282-288, get_sphere_uv tracer.h:63-82
        Vec d = r.direction.normalize();
        double u = std::atan2(d.y, d.x), v = M_PI - std::acos(d.z);
        return w.env.sample(u, v);
    }
    return Vec(P.background);
}

// tracer.h:23-54
Vec tonemap(const Vec& c, int mode) {
    if (mode == RT_TONEMAP_REINHARD) return {c.x / (1.0 + c.x), c.y / (1.0 + c.y), c.z / (1.0 + c.z)};
    if (mode == RT_TONEMAP_ACES) {
        auto f = [](double x) { return (x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14); };
        return {f(c.x), f(c.y), f(c.z)};
    }
    if (mode == RT_TONEMAP_FILMIC) {
        auto f = [](double x) {
            const double A = 0.15, B = 0.50, C = 0.10, D = 0.20, E = 0.02, F = 0.30;
            return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
        };
        double ws = 1.0 / f(11.2);
        return {f(c.x) * ws, f(c.y) * ws, f(c.z) * ws};
    }
    return c;
}

unsigned char quantise(double v) {  // tracer.h:297-308
    double c = std::max(0.0, std::min(1.0, v));
    return static_cast<unsigned char>(255 * c);
}

Material make_material(const RtMaterial& m, const std::vector<Tex>& tex) {
    Material o;
    o.ambient = Vec(m.ambient);
    o.diffuse = Vec(m.diffuse);
    o.specular = Vec(m.specular);
    o.shininess = m.shininess;
    o.reflectivity = m.reflectivity;
    o.transparency = m.transparency;
    o.ior = m.refractive_index;
    o.texture = (m.texture >= 0 && m.texture < (int)tex.size()) ? &tex[m.texture] : nullptr;
    o.bump = (m.bump_map >= 0 && m.bump_map < (int)tex.size()) ? &tex[m.bump_map] : nullptr;
    o.metal = m.is_metal != 0;
    return o;
}

std::unique_ptr<World> make_world(const RtSceneDesc& d, const RtRenderParams& p) {
    auto w = std::make_unique<World>();
    w->cfg.p = p;
    w->cfg.shutter = d.shutter_time;
    w->textures.resize(d.n_textures);
    for (int i = 0; i < d.n_textures; ++i) w->textures[i] = Tex{d.textures[i].width, d.textures[i].height, d.textures[i].rgb};
    w->lights.assign(d.lights, d.lights + d.n_lights);
    w->env.present = d.env.present;
    w->env.w = d.env.width;
    w->env.h = d.env.height;
    w->env.rgb = d.env.rgb;
    for (int i = 0; i < d.n_shapes; ++i) {
        const RtShape& s = d.shapes[i];
        Material m = make_material(d.materials[s.material], w->textures);
        if (s.type == RT_SHAPE_PLANE) {
            auto q = std::make_unique<Quad>(Vec(s.corners), Vec(s.corners + 3), Vec(s.corners + 6), Vec(s.corners + 9));
            q->mat = m;
            q->velocity = Vec(s.velocity);
            q->shutter = d.shutter_time;
            q->cfg = &w->cfg;
            w->shapes.push_back(std::move(q));
            continue;
        }
        std::unique_ptr<Transformed> t;
        switch (s.type) {
            case RT_SHAPE_SPHERE: t = std::make_unique<Sphere>(); break;
            case RT_SHAPE_CUBE: t = std::make_unique<Cube>(); break;
            case RT_SHAPE_COMPLEX_SPHERE: { auto c = std::make_unique<ComplexSphere>(); c->D = s.max_displacement; t = std::move(c); break; }
            case RT_SHAPE_COMPLEX_CUBE: { auto c = std::make_unique<ComplexCube>(); c->D = s.max_displacement; t = std::move(c); break; }
            default: { auto c = std::make_unique<ComplexPlane>(); c->D = s.max_displacement; t = std::move(c); break; }
        }
        t->fwd = Mat::from(s.transform);
        t->inv = Mat::from(s.inv_transform);
        t->inv_t = t->inv.transpose();
        t->mat = m;
        t->velocity = Vec(s.velocity);
        t->shutter = d.shutter_time;
        t->cfg = &w->cfg;
        w->shapes.push_back(std::move(t));
    }
    if (p.use_bvh && !w->shapes.empty()) {  // scene.cpp:102-118
        w->boxes.resize(w->shapes.size());
        for (size_t i = 0; i < w->shapes.size(); ++i) w->boxes[i] = w->shapes[i]->bounds();
        std::vector<int> idx(w->shapes.size());
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)i;
        w->nodes.reserve(w->shapes.size());
        w->root = w->build(idx, 0, idx.size());
    }
    // camera.cpp:27-86
    const RtCamera& c = d.camera;
    w->cam_loc = Vec(c.location);
    w->focal = c.focal_length;
    w->sensor_w = c.sensor_width;
    w->W = c.resolution_x;
    w->H = c.resolution_y;
    double aspect = static_cast<double>(c.resolution_x) / static_cast<double>(c.resolution_y);
    w->sensor_h = c.sensor_width / aspect;
    w->focal_dist = c.focal_distance;
    double fl_m = c.focal_length / 1000.0;
    w->aperture = (c.f_stop > 0) ? fl_m / (2.0 * c.f_stop) : 0.0;
    w->cam_w = Vec(c.gaze_direction).normalize();
    w->cam_u = Vec(c.up_vector).cross(w->cam_w).normalize();
    w->cam_v = w->cam_u.cross(w->cam_w);
    return w;
}

}  // namespace

static double g_last_render_seconds = 0.0;
static int g_last_threads = 1;

extern "C" {

// Render rows [row_begin, row_end) with stride row_step of the frame (row_step > 1 renders a sample of rows for
// timing); untouched rows keep their previous contents.  counters: rays, box tests, primitive tests (may be NULL).
// threads <= 0 uses the OpenMP default.  Returns 0, or -1 on bad arguments.
int rt_oracle_render_rows(const RtSceneDesc* desc, const RtRenderParams* params, uint8_t* rgb8_out, int row_begin,
                          int row_end, int row_step, int threads, uint64_t* counters) {
    if (!desc || !params || !rgb8_out || row_step <= 0) return -1;
    std::unique_ptr<World> w = make_world(*desc, *params);
    const int W = w->W, H = w->H;
    row_begin = std::max(0, row_begin);
    row_end = std::min(H, row_end);
    const int spp = params->samples_per_pixel;
    uint64_t tot_rays = 0, tot_boxes = 0, tot_prims = 0;
    int nrows = (row_end - row_begin + row_step - 1) / row_step;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    double t_begin = 0.0;
#ifdef _OPENMP
    t_begin = omp_get_wtime();
#endif
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : tot_rays, tot_boxes, tot_prims)
    for (int k = 0; k < nrows; ++k) {  // main.cpp:281
        int y = row_begin + k * row_step;
        Counters local;
        tl_counters = counters ? &local : nullptr;
        for (int x = 0; x < W; ++x) {
            Vec acc;
            for (int s = 0; s < spp; ++s) {  // main.cpp:288-306
                Rng rng;
                rng.d0 = params->rng_mode == RT_RNG_D0;
                rng.s = (params->seed * 0x9E3779B97F4A7C15ull) ^ ((uint64_t(y) * W + x) * 0xD1B54A32D192ED03ull + uint64_t(s) * 0x8CB92BA72F3D8DD7ull);
                float ru = rng.uniform(), rv = rng.uniform();
                float px = (static_cast<float>(x) + ru) / W;
                float py = (static_cast<float>(y) + rv) / H;
                double time = rng.uniform() * w->cfg.shutter;
                Ray ray = w->generate_ray(px, py, time, rng);
                acc = acc + ray_colour(*w, ray, params->max_depth, rng);
            }
            Vec avg = tonemap(acc * (1.0 / spp), params->tonemap);
            uint8_t* o = rgb8_out + (static_cast<size_t>(y) * W + x) * 3;
            o[0] = quantise(avg.x);
            o[1] = quantise(avg.y);
            o[2] = quantise(avg.z);
        }
        tot_rays += local.rays;
        tot_boxes += local.boxes;
        tot_prims += local.prims;
        tl_counters = nullptr;
    }
#ifdef _OPENMP
    g_last_render_seconds = omp_get_wtime() - t_begin;
    g_last_threads = omp_get_max_threads();
#endif
    if (counters) { counters[0] = tot_rays; counters[1] = tot_boxes; counters[2] = tot_prims; }
    return 0;
}

// Wall time of the pixel loop of the last rt_oracle_render_rows call (scene construction and BVH build excluded —
// the reference's own timer includes them, main.cpp:230,350) and the OpenMP thread count it ran with.
double rt_oracle_last_render_seconds(void) { return g_last_render_seconds; }
int rt_oracle_last_threads(void) { return g_last_threads; }

int rt_oracle_render(const RtSceneDesc* desc, const RtRenderParams* params, uint8_t* rgb8_out, int threads, uint64_t* counters) {
    if (!desc) return -1;
    return rt_oracle_render_rows(desc, params, rgb8_out, 0, desc->camera.resolution_y, 1, threads, counters);
}

// Float (pre tone-map, pre quantisation) image for PSNR work: rgb_out[H*W*3] doubles.
int rt_oracle_render_float(const RtSceneDesc* desc, const RtRenderParams* params, double* rgb_out, int threads) {
    if (!desc || !params || !rgb_out) return -1;
    std::unique_ptr<World> w = make_world(*desc, *params);
    const int W = w->W, H = w->H, spp = params->samples_per_pixel;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            Vec acc;
            for (int s = 0; s < spp; ++s) {
                Rng rng;
                rng.d0 = params->rng_mode == RT_RNG_D0;
                rng.s = (params->seed * 0x9E3779B97F4A7C15ull) ^ ((uint64_t(y) * W + x) * 0xD1B54A32D192ED03ull + uint64_t(s) * 0x8CB92BA72F3D8DD7ull);
                float ru = rng.uniform(), rv = rng.uniform();
                float px = (static_cast<float>(x) + ru) / W, py = (static_cast<float>(y) + rv) / H;
                double time = rng.uniform() * w->cfg.shutter;
                Ray ray = w->generate_ray(px, py, time, rng);
                acc = acc + ray_colour(*w, ray, params->max_depth, rng);
            }
            Vec avg = acc * (1.0 / spp);
            double* o = rgb_out + (static_cast<size_t>(y) * W + x) * 3;
            o[0] = avg.x; o[1] = avg.y; o[2] = avg.z;
        }
    return 0;
}

}  // extern "C"
