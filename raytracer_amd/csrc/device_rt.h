// Device-side building blocks shared by the render kernels (render_wavefront.hip, render_mega.hip): FP64 vector
// algebra, RNG, texture / environment lookups, primitive tests, BVH traversal step helpers and hit-record
// reconstruction.  Every function cites the reference code it reproduces (paths relative to /root/reference/Code).
// Arithmetic is FP64 throughout (the parity bar is 8-bit max|diff| <= 1 against an FP64 reference); the including
// translation units are compiled with -ffp-contract=off so that sums and products round where the reference's do.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace rt {

namespace {

constexpr double kPi = 3.14159265358979323846;

// ------------------------------------------------------------------------------------------ small vector algebra
struct D3 {
    double x, y, z;
};
__device__ __forceinline__ D3 mk(double x, double y, double z) { return D3{x, y, z}; }
__device__ __forceinline__ D3 ld3(const double* p) { return D3{p[0], p[1], p[2]}; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator-(D3 a) { return D3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ D3 operator*(D3 a, double s) { return D3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ D3 operator*(double s, D3 a) { return D3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ D3 operator/(D3 a, double s) { return D3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ D3 cmul(D3 a, D3 b) { return D3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 cross(D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double length(D3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
// vector3.h:92-102
// The three quotients share their denominator.  The compiler expands an FP64 division into v_div_scale, v_rcp_f64, two Newton
// steps on the reciprocal, q = x r, the remainder x - q len and one correction (v_div_fmas, v_div_fixup): correctly rounded, ~ 11
// instructions and a quarter-rate reciprocal EACH.  Here the refined reciprocal is computed once and every component takes the
// last three steps: the same operations on the same values -- hence the same bits -- whenever the scaling of v_div_scale is the
// identity, which it is for a length in (1e-6, 1e150) and components no larger than it (RT_PLAIN_DIVISIONS: A/B runs).
__device__ __forceinline__ D3 normalize(D3 a) {
    double len = length(a);
#ifndef RT_PLAIN_DIVISIONS
    if (len > 1e-6 && len < 1e150) {
        double y = __builtin_amdgcn_rcp(len);
        double e = fma(-len, y, 1.0);
        y = fma(y, e, y);
        e = fma(-len, y, 1.0);
        y = fma(y, e, y);
        double qx = a.x * y, qy = a.y * y, qz = a.z * y;
        return D3{fma(fma(-len, qx, a.x), y, qx), fma(fma(-len, qy, a.y), y, qy), fma(fma(-len, qz, a.z), y, qz)};
    }
#endif
    if (len > 1e-6) return D3{a.x / len, a.y / len, a.z / len};
    return D3{0.0, 0.0, 0.0};
}
// std::min / std::max semantics (a NaN second argument is ignored, a NaN first argument is returned)
__device__ __forceinline__ double smin(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ double smax(double a, double b) { return (a < b) ? b : a; }
__device__ __forceinline__ int imin(int a, int b) { return b < a ? b : a; }
__device__ __forceinline__ int imax(int a, int b) { return a < b ? b : a; }
// x86 cvttsd2si: out-of-range and NaN give INT_MIN (the device conversion would saturate / give 0).
__device__ __forceinline__ int to_int(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000;
    return (int)v;
}

struct Ray {
    D3 o, d;
    double time;
};

// ------------------------------------------------------------------------------------------ random numbers
// Counter-based stream per (seed, pixel, sample): splitmix64.  RT_RNG_D0 returns the constants of the oracle's "d0" mode.
struct Rng {
    unsigned long long s;
    bool d0;
    __device__ __forceinline__ unsigned long long next() {
        unsigned long long z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    __device__ __forceinline__ double uniform() { return d0 ? 0.5 : (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    // A direction uniform on the unit sphere from 42 bits of one draw: height z = 1 - 2u (uniform by Archimedes), azimuth 2 pi v,
    // in FP32 (v_sin_f32 / v_cos_f32 take their argument in revolutions).
    static __device__ __forceinline__ void unit_direction(unsigned long long z, float& x, float& y, float& w) {
        const float u = ((float)(unsigned)(z & 0x1FFFFFull) + 0.5f) * (1.0f / 2097152.0f);
        const float v = ((float)(unsigned)((z >> 21) & 0x1FFFFFull) + 0.5f) * (1.0f / 2097152.0f);
        w = 1.0f - 2.0f * u;
        const float s = __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - w * w));
        x = s * __builtin_amdgcn_cosf(v);
        y = s * __builtin_amdgcn_sinf(v);
    }
    // random_utils.h:29-35: a point uniform in the unit ball.  The reference rejects points of the cube; a wave would repeat
    // such a loop until its unluckiest lane has accepted (6-8 rounds at 52 % acceptance, each with the generator's two 64-bit
    // multiplies), so the same distribution is drawn directly instead: a uniform direction times a radius u^(1/3), from ONE
    // draw (21 + 21 + 22 bits), in FP32.  The deterministic mode returns the centre, as the oracle's shim does.
    __device__ D3 in_unit_sphere() {
        if (d0) return mk(0, 0, 0);
        const unsigned long long z = next();
        float x, y, w;
        unit_direction(z, x, y, w);
        const float t = ((float)(unsigned)(z >> 42) + 0.5f) * (1.0f / 4194304.0f);
        const float r = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(t) * (1.0f / 3.0f));
        return mk((double)(r * x), (double)(r * y), (double)(r * w));
    }
    // random_in_unit_sphere().normalize() (shading.h:35): a direction uniform on the sphere, which is what normalising a uniform
    // point of the ball gives; unit length to FP32 rounding.  Deterministic mode: the zero vector (vector3.h:92-102 on the centre).
    __device__ D3 on_unit_sphere() {
        if (d0) return mk(0, 0, 0);
        float x, y, w;
        unit_direction(next(), x, y, w);
        return mk((double)x, (double)y, (double)w);
    }
    // camera.cpp:13-24: a point uniform in the unit disk (the reference rejects points of the square), drawn directly like the
    // ball's: radius sqrt(u), azimuth 2 pi v, 24 bits of one draw each, FP32.
    __device__ D3 in_unit_disk() {
        if (d0) return mk(0, 0, 0);
        const unsigned long long z = next();
        const float u = ((float)(unsigned)(z >> 40) + 0.5f) * (1.0f / 16777216.0f);
        const float v = ((float)(unsigned)(z & 0xFFFFFFull) + 0.5f) * (1.0f / 16777216.0f);
        const float r = __builtin_amdgcn_sqrtf(u);
        return mk((double)(r * __builtin_amdgcn_cosf(v)), (double)(r * __builtin_amdgcn_sinf(v)), 0.0);
    }
};

// ------------------------------------------------------------------------------------------ textures / environment
struct Px {
    int r, g, b;
};
__device__ __forceinline__ Px tex_get(const DTexture& t, int x, int y) {  // Image.cpp:191-200 (callers clamp)
    const uint8_t* p = t.rgb + ((size_t)y * t.width + x) * 3;
    return Px{p[0], p[1], p[2]};
}
// Image.cpp:7-65: clamp, interpolate, truncate each channel to 8 bits.
__device__ Px tex_bilinear(const DTexture& t, double u, double v) {
    u = smax(0.0, smin(1.0, u));
    v = smax(0.0, smin(1.0, v));
    double px = u * (t.width - 1), py = v * (t.height - 1);
    int x0 = (int)floor(px), y0 = (int)floor(py);
    int x1 = imin(x0 + 1, t.width - 1), y1 = imin(y0 + 1, t.height - 1);
    double dx = px - x0, dy = py - y0;
    Px c00 = tex_get(t, x0, y0), c10 = tex_get(t, x1, y0), c01 = tex_get(t, x0, y1), c11 = tex_get(t, x1, y1);
    double rt = (1.0 - dx) * c00.r + dx * c10.r, gt = (1.0 - dx) * c00.g + dx * c10.g, bt = (1.0 - dx) * c00.b + dx * c10.b;
    double rb = (1.0 - dx) * c01.r + dx * c11.r, gb = (1.0 - dx) * c01.g + dx * c11.g, bb = (1.0 - dx) * c01.b + dx * c11.b;
    return Px{(int)(unsigned char)((1.0 - dy) * rt + dy * rb), (int)(unsigned char)((1.0 - dy) * gt + dy * gb),
              (int)(unsigned char)((1.0 - dy) * bt + dy * bb)};
}
__device__ __forceinline__ double px_intensity(Px p) { return (p.r + p.g + p.b) / (3.0 * 255.0); }
// sphere.cpp:143-152: clamped nearest texel intensity
__device__ __forceinline__ double tex_intensity_at(const DTexture& t, int x, int y) {
    x = imin(imax(x, 0), t.width - 1);
    y = imin(imax(y, 0), t.height - 1);
    return px_intensity(tex_get(t, x, y));
}

// HDRImage.cpp:112-182 with the angles of tracer.h:63-82
__device__ D3 env_sample(const DScene& S, D3 dir) {
    if (!S.env || S.env_w <= 0) return mk(0, 0, 0);
    D3 d = normalize(dir);
    double u = atan2(d.y, d.x), v = kPi - acos(d.z);
    const int w = S.env_w, h = S.env_h;
    double U = u / (2.0 * kPi), V = v / kPi;
    double px = U * (w - 1), py = V * (h - 1);
    px = fmod(px, (double)w);
    if (px < 0) px += w;
    {   // std::clamp(py, 0, h-1)
        double hi = (double)(h - 1);
        py = (py < 0.0) ? 0.0 : ((hi < py) ? hi : py);
    }
    int x0 = (int)floor(px), y0 = (int)floor(py);
    int x1 = x0 + 1, y1 = y0 + 1;
    double dx = px - x0, dy = py - y0;
    x1 = x1 % w;
    if (x1 < 0) x1 += w;
    y1 = imin(y1, h - 1);
    x0 = imin(imax(x0, 0), w - 1);  // a NaN direction cannot index outside the map
    y0 = imin(imax(y0, 0), h - 1);
    auto at = [&](int x, int y) {
        const float* p = S.env + ((size_t)y * w + x) * 3;
        return mk((double)p[0], (double)p[1], (double)p[2]);
    };
    D3 c00 = at(x0, y0), c10 = at(x1, y0), c01 = at(x0, y1), c11 = at(x1, y1);
    D3 bottom = c00 * (1.0 - dx) + c10 * dx, top = c01 * (1.0 - dx) + c11 * dx;
    return bottom * (1.0 - dy) + top * dy;
}

// ------------------------------------------------------------------------------------------ object space
struct ObjRay {
    D3 o, d;
};
// sphere.cpp:47-53: origin moved back by velocity*time, then the point / direction transforms of matrix4x4.cpp:80-103.
__device__ __forceinline__ ObjRay to_object(const DXform& X, const Ray& r) {
    D3 o0 = r.o - ld3(X.vel) * r.time;
    const double* m = X.inv;
    double x = m[0] * o0.x + m[1] * o0.y + m[2] * o0.z + m[3];
    double y = m[4] * o0.x + m[5] * o0.y + m[6] * o0.z + m[7];
    double z = m[8] * o0.x + m[9] * o0.y + m[10] * o0.z + m[11];
    double w = X.inv33;  // row 3 is (0,0,0,inv33) for the affine transforms the host accepts
    ObjRay q;
    q.o = (w != 0.0 && w != 1.0) ? mk(x / w, y / w, z / w) : mk(x, y, z);
    q.d = mk(m[0] * r.d.x + m[1] * r.d.y + m[2] * r.d.z, m[4] * r.d.x + m[5] * r.d.y + m[6] * r.d.z,
             m[8] * r.d.x + m[9] * r.d.y + m[10] * r.d.z);
    return q;
}
// Normal transform: (inverse transpose) * n as a *point* (sphere.cpp:104, matrix4x4.cpp:80-93), then normalize.
__device__ __forceinline__ D3 normal_to_world(const DXform& X, D3 n) {
    const double* m = X.inv;
    double x = m[0] * n.x + m[4] * n.y + m[8] * n.z;
    double y = m[1] * n.x + m[5] * n.y + m[9] * n.z;
    double z = m[2] * n.x + m[6] * n.y + m[10] * n.z;
    double w = m[3] * n.x + m[7] * n.y + m[11] * n.z + X.inv33;
    D3 v = (w != 0.0 && w != 1.0) ? mk(x / w, y / w, z / w) : mk(x, y, z);
    return normalize(v);
}

__device__ __forceinline__ void sphere_uv(D3 p, double& u, double& v) {  // sphere.cpp:28-41
    double theta = asin(p.y);
    double phi = atan2(-p.z, p.x) + kPi;
    u = phi / (2.0 * kPi);
    v = (theta + kPi / 2.0) / kPi;
}

// cube.cpp:102-175 / complex_cube.cpp:49-118: face from the dominant coordinate, uv into the 4x3 cross atlas.
__device__ void cube_face_uv(D3 p, bool clamp_raw, D3& n, double& u, double& v) {
    double ax = fabs(p.x), ay = fabs(p.y), az = fabs(p.z);
    int axis;
    n = mk(0, 0, 0);
    if (ax >= ay && ax >= az) { axis = 0; n.x = (p.x > 0) ? 1.0 : -1.0; }
    else if (ay >= ax && ay >= az) { axis = 1; n.y = (p.y > 0) ? 1.0 : -1.0; }
    else { axis = 2; n.z = (p.z > 0) ? 1.0 : -1.0; }
    double ru, rv;
    if (axis == 0) { ru = (p.y * (n.x > 0 ? -1 : 1) + 1.0) * 0.5; rv = (p.z + 1.0) * 0.5; }
    else if (axis == 1) { ru = (p.x * (n.y > 0 ? 1 : -1) + 1.0) * 0.5; rv = (p.z + 1.0) * 0.5; }
    else { ru = (p.x + 1.0) * 0.5; rv = (p.y + 1.0) * 0.5; }
    if (clamp_raw) {
        ru = smax(0.0, smin(1.0, ru));
        rv = smax(0.0, smin(1.0, rv));
    }
    double uo, vo;
    if (axis == 2) { uo = 1.0; vo = (n.z > 0) ? 2.0 : 0.0; }
    else if (axis == 1) { uo = (n.y > 0) ? 1.0 : 3.0; vo = 1.0; }
    else { uo = (n.x > 0) ? 2.0 : 0.0; vo = 1.0; }
    u = (ru + uo) * 0.25;
    v = (rv + vo) * (1.0 / 3.0);
}

// complex_cube.cpp:34-46 / complex_plane.cpp:78-92
__device__ __forceinline__ double box_sdf(D3 p, D3 half) {
    D3 d = mk(fabs(p.x), fabs(p.y), fabs(p.z)) - half;
    double inside = smin(smax(d.x, smax(d.y, d.z)), 0.0);
    double outside = length(mk(smax(d.x, 0.0), smax(d.y, 0.0), smax(d.z, 0.0)));
    return inside + outside;
}

// Displaced-surface distance functions (complex_sphere.cpp:124-142, complex_cube.cpp:227-248, complex_plane.cpp:209-214).
__device__ double sdf_complex_sphere(const DScene& S, const DMaterial& M, double D, D3 q, double* ou, double* ov) {
    double ql = length(q);
    D3 qu = q / ql;
    double u, v;
    sphere_uv(qu, u, v);
    double disp = 0.0;
    if (M.bump >= 0) disp = px_intensity(tex_bilinear(S.textures[M.bump], u, 1.0 - v)) * D;
    if (ou) { *ou = u; *ov = v; }
    return ql - (1.0 + disp);
}
__device__ double sdf_complex_cube(const DScene& S, const DMaterial& M, double D, D3 q, double* ou, double* ov) {
    double u, v;
    D3 n;
    cube_face_uv(q, true, n, u, v);
    double disp = 0.0;
    if (M.bump >= 0) {
        const DTexture& t = S.textures[M.bump];
        int x = to_int(u * (t.width - 1)), y = to_int((1.0 - v) * (t.height - 1));
        x = imax(0, imin(x, t.width - 1));
        y = imax(0, imin(y, t.height - 1));
        disp = px_intensity(tex_get(t, x, y)) * D;
    }
    if (ou) { *ou = u; *ov = v; }
    return box_sdf(q, mk(1, 1, 1)) - disp;
}
__device__ double sdf_complex_plane(const DScene& S, const DMaterial& M, double D, D3 q, double* ou, double* ov) {
    double u = (q.x + 1.0) * 0.5, v = (q.y + 1.0) * 0.5;
    u = smax(0.0, smin(1.0, u));
    v = smax(0.0, smin(1.0, v));
    double disp = 0.0;
    if (M.bump >= 0) disp = px_intensity(tex_bilinear(S.textures[M.bump], u, 1.0 - v)) * D;
    if (ou) { *ou = u; *ov = v; }
    return box_sdf(q, mk(1.0, 1.0, 0.001)) - disp;
}
__device__ __forceinline__ double sdf_complex(int type, const DScene& S, const DMaterial& M, double D, D3 q, double* ou, double* ov) {
    if (type == RT_SHAPE_COMPLEX_SPHERE) return sdf_complex_sphere(S, M, D, q, ou, ov);
    if (type == RT_SHAPE_COMPLEX_CUBE) return sdf_complex_cube(S, M, D, q, ou, ov);
    return sdf_complex_plane(S, M, D, q, ou, ov);
}

// ------------------------------------------------------------------------------------------ primitive tests
// Candidate = the closest accepted hit so far.  tri/bu/bv carry the quad's triangle and barycentrics.
struct Cand {
    double t;
    int prim;   // -1 none
    int tri;
    double bu, bv;
};

// plane.cpp:77-134 (Moeller-Trumbore with the reference's un-normalised parallel test and strict t range)
// `closed_top` admits t == tmax: used once a hit exists, so that an equal-t triangle reaches the tie rule instead of
// being decided by this kernel's own visiting order (see tie_new_wins).
__device__ __forceinline__ bool tri_test(D3 o, D3 d, double tmin, double tmax, bool closed_top, double eps, D3 v0, D3 e1, D3 e2,
                                         double& ot, double& ou, double& ov) {
    D3 h = cross(d, e2);
    double a = dot(e1, h);
    if (a > -eps && a < eps) return false;
    double f = 1.0 / a;
    D3 s = o - v0;
    double u = f * dot(s, h);
    if (u < 0.0 || u > 1.0) return false;
    D3 q = cross(s, e1);
    double v = f * dot(d, q);
    if (v < 0.0 || u + v > 1.0) return false;
    ot = f * dot(e2, q);
    if (ot > tmin && (ot < tmax || (closed_top && ot == tmax))) { ou = u; ov = v; return true; }
    return false;
}

// Slab entry/exit for the displaced cube / plane bounds (complex_cube.cpp:130-160, complex_plane.cpp:123-158).
__device__ __forceinline__ bool slab_range(D3 oo, D3 od, D3 b, double& tn, double& tf) {
    tn = -INFINITY;
    tf = INFINITY;
    const double o[3] = {oo.x, oo.y, oo.z}, d[3] = {od.x, od.y, od.z}, bv[3] = {b.x, b.y, b.z};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (d[i] == 0.0) {
            if (o[i] < -bv[i] || o[i] > bv[i]) return false;
        } else {
            double t0 = (-bv[i] - o[i]) / d[i], t1 = (bv[i] - o[i]) / d[i];
            if (d[i] < 0.0) { double s = t0; t0 = t1; t1 = s; }
            if (t0 > tn) tn = t0;
            if (t1 < tf) tf = t1;
        }
    }
    return !(tn > tf || tf < 0.0);
}

// Ray-marched displaced shapes (complex_sphere.cpp:30-168, complex_cube.cpp:121-278, complex_plane.cpp:112-233).
__device__ bool march_complex(const DScene& S, const DParams& P, const DPrim& pr, const Ray& r, double tmin, double tmax,
                                           double& t_out) {
    const DXform& X = S.xforms[pr.data];
    const DMaterial& M = S.materials[pr.material];
    const double D = X.disp;
    ObjRay q = to_object(X, r);
    double t, tl, step_scale = 1.0, step_mul;
    if (pr.type == RT_SHAPE_COMPLEX_SPHERE) {
        double max_r = 1.0 + D;
        double a = dot(q.d, q.d), b = 2.0 * dot(q.o, q.d), c = dot(q.o, q.o) - max_r * max_r;
        double disc = b * b - 4 * a * c;
        if (disc < 0) return false;
        double te = (-b - sqrt(disc)) / (2.0 * a), tx = (-b + sqrt(disc)) / (2.0 * a);
        if (tx < tmin || te > tmax) return false;
        t = smax(te, tmin);
        tl = smin(tx, tmax);
        step_mul = 0.5;
    } else {
        double tn, tf;
        D3 bound = (pr.type == RT_SHAPE_COMPLEX_CUBE) ? mk(1.0 + D, 1.0 + D, 1.0 + D) : mk(1.0, 1.0, D + 0.01);
        if (!slab_range(q.o, q.d, bound, tn, tf)) return false;
        t = smax(tn, tmin);
        tl = smin(tf, tmax);
        if (pr.type == RT_SHAPE_COMPLEX_CUBE) {
            step_mul = 0.6;
        } else {
            step_scale = length(q.d);  // complex_plane.cpp:121,229: only the plane converts to world units
            step_mul = P.step_multiplier;
        }
    }
    const double EPS = P.eps_march;
    for (int i = 0; i < P.march_steps; ++i) {
        if (t > tl) break;
        D3 p = q.o + q.d * t;
        double ds = sdf_complex(pr.type, S, M, D, p, nullptr, nullptr);
        if (ds < EPS) { t_out = t; return true; }
        if (pr.type == RT_SHAPE_COMPLEX_PLANE) t += smax((ds / step_scale) * step_mul, EPS);
        else t += smax(ds * step_mul, EPS);
    }
    return false;
}

// Per-lane traversal stack of (child reference, entry distance) pairs.  The first kTravLdsEntries entries live in
// LDS: entry i of this lane is lds[i * stride] (stride = threads per block, so that the lanes of a wave fall on
// consecutive banks); deeper entries, which only unbalanced hierarchies reach, go to per-lane scratch memory.
struct TravStack {
    unsigned long long* lds;
    int stride;
    unsigned long long spill[kTravSpillEntries];
    __device__ __forceinline__ void push(int& sp, int ref, float t) {
        const unsigned long long v = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned)ref;
        if (sp < kTravLdsEntries) lds[sp * stride] = v;
        else spill[sp - kTravLdsEntries] = v;
        ++sp;
    }
    // Defers `count` (0..3) entries; (r1, t1) ends on top, then (r2, t2), then (r3, t3): an entry k is real when
    // k <= count.  With room in LDS for all three the stores are unconditional (unused ones go to the discard row
    // kTravLdsEntries), so the usual case has no branch at all.
    __device__ __forceinline__ void push_sorted(int& sp, int count, int r1, float t1, int r2, float t2, int r3, float t3) {
        if (__builtin_expect(sp + 3 <= kTravLdsEntries, 1)) {
            const int s1 = count >= 1 ? sp + count - 1 : kTravLdsEntries;
            const int s2 = count >= 2 ? sp + count - 2 : kTravLdsEntries;
            const int s3 = count >= 3 ? sp : kTravLdsEntries;
            lds[s1 * stride] = ((unsigned long long)__float_as_uint(t1) << 32) | (unsigned)r1;
            lds[s2 * stride] = ((unsigned long long)__float_as_uint(t2) << 32) | (unsigned)r2;
            lds[s3 * stride] = ((unsigned long long)__float_as_uint(t3) << 32) | (unsigned)r3;
            sp += count;
        } else {
            if (count >= 3) push(sp, r3, t3);
            if (count >= 2) push(sp, r2, t2);
            if (count >= 1) push(sp, r1, t1);
        }
    }
    // pop() for a lane with a parking slot `pend` (0 = free): a popped leaf goes to the free slot and the lane pops on, so
    // that it comes back with node work if it has any (k_traverse).
    __device__ __forceinline__ int pop_parking(int& sp, float tmax, int& pend) {
        while (sp > 0) {
            --sp;
            unsigned long long v;
            if (__builtin_expect(sp < kTravLdsEntries, 1)) {
                v = lds[sp * stride];
            } else {
                v = spill[sp - kTravLdsEntries];
                asm volatile("" : "+v"(v));
            }
            if (__uint_as_float((unsigned)(v >> 32)) <= tmax) {
                const int ref = (int)(unsigned)v;
                if (ref < 0 && pend == 0) { pend = ref; continue; }
                return ref;
            }
        }
        return 0x7fffffff;
    }
    // Next deferred child whose box the ray enters before tmax (the closest hit so far), or kDoneRef.
    __device__ __forceinline__ int pop(int& sp, float tmax) {
        while (sp > 0) {
            --sp;
            unsigned long long v;
            if (__builtin_expect(sp < kTravLdsEntries, 1)) {
                v = lds[sp * stride];
            } else {
                v = spill[sp - kTravLdsEntries];
                asm volatile("" : "+v"(v));   // keeps the two loads apart (merged, they become one flat load)
            }
            if (__uint_as_float((unsigned)(v >> 32)) <= tmax) return (int)(unsigned)v;
        }
        return 0x7fffffff;
    }
};

struct Counters {
    unsigned int rays, nodes, prims, visits;   // closest-hit queries, box tests, primitive tests, interior nodes fetched
};

// Compile-time feature set of a kernel variant (RT_FEAT_* bits of device_types.h).  A scene is rendered by the
// smallest variant that covers scene.features, so that e.g. a spheres-only BVH stress frame does not carry the
// registers of the ray marcher or the texture paths.
template <int F> struct Feat {
    static constexpr bool quad = (F & RT_FEAT_QUAD) != 0;
    static constexpr bool cube = (F & RT_FEAT_CUBE) != 0;
    static constexpr bool complex_ = (F & RT_FEAT_COMPLEX) != 0;
    static constexpr bool tex = (F & RT_FEAT_TEXTURE) != 0;
    static constexpr bool env = (F & RT_FEAT_ENV) != 0;
    static constexpr bool sphere = (F & RT_FEAT_SPHERE) != 0;
};

// Tie rule.  The reference visits primitives in a fixed order and replaces the current hit when a later test
// succeeds with t_max = current t: spheres/cubes/displaced shapes accept t == t_max (sphere.cpp:83, cube.cpp:85),
// triangles do not (plane.cpp:126).  Among equal-t hits the reference therefore keeps the LAST inclusive
// primitive in visit order, or the FIRST primitive if all are triangles.  `order` is that visit order, so the
// result does not depend on the order this kernel happens to test primitives in.
__device__ __forceinline__ bool tie_new_wins(bool new_incl, int new_order, bool cur_incl, int cur_order) {
    if (new_incl != cur_incl) return new_incl;
    return new_incl ? (new_order > cur_order) : (new_order < cur_order);
}

// The FP32 sphere test of the throughput variant (RT_SCHEDULE_WAVEFRONT_FP32, spheres-only scenes): same roots as
// sphere.cpp:44-90 in exact arithmetic, evaluated around the point of closest approach -- with the ray origin hundreds of
// radii away, o.o - 1 of the textbook form loses the sphere in the rounding of o.o (2^-24 * |o|^2), whereas the closest
// approach p = o + tc d is a point near the unit sphere whose error is 2^-24 * |o|.  Hit or miss and t are decided in FP32:
// silhouette pixels and a few t ties may come out differently from the FP64 parity path, which is what the variant's
// mismatch count reports.
struct Ray32 {
    float o[3], d[3], time;
};
__device__ __forceinline__ Ray32 make_ray32(const Ray& r) {
    return Ray32{{(float)r.o.x, (float)r.o.y, (float)r.o.z}, {(float)r.d.x, (float)r.d.y, (float)r.d.z}, (float)r.time};
}
template <bool COUNT>
__device__ __forceinline__ void test_sphere32(const DScene& S, const DParams& P, const Ray32& r, int prim, float tmin, Cand& best,
                                              int& best_order, bool& best_incl, Counters& C) {
    if (COUNT && P.count_tests) C.prims++;
    const float4* X = reinterpret_cast<const float4*>(S.xforms32) + 4 * (size_t)prim;
    const float4 m0 = X[0], m1 = X[1], m2 = X[2], vw = X[3];   // rows of world -> object; velocity xyz, inv33
    const float ox = fmaf(-vw.x, r.time, r.o[0]), oy = fmaf(-vw.y, r.time, r.o[1]), oz = fmaf(-vw.z, r.time, r.o[2]);
    float qx = fmaf(m0.x, ox, fmaf(m0.y, oy, fmaf(m0.z, oz, m0.w)));
    float qy = fmaf(m1.x, ox, fmaf(m1.y, oy, fmaf(m1.z, oz, m1.w)));
    float qz = fmaf(m2.x, ox, fmaf(m2.y, oy, fmaf(m2.z, oz, m2.w)));
    if (vw.w != 0.f && vw.w != 1.f) {   // matrix4x4.cpp:85-90
        const float iw = 1.f / vw.w;
        qx *= iw; qy *= iw; qz *= iw;
    }
    const float dx = fmaf(m0.x, r.d[0], fmaf(m0.y, r.d[1], m0.z * r.d[2]));
    const float dy = fmaf(m1.x, r.d[0], fmaf(m1.y, r.d[1], m1.z * r.d[2]));
    const float dz = fmaf(m2.x, r.d[0], fmaf(m2.y, r.d[1], m2.z * r.d[2]));
    const float a = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
    const float ia = __builtin_amdgcn_rcpf(a);
    const float tc = -fmaf(qx, dx, fmaf(qy, dy, qz * dz)) * ia;          // parameter of the closest approach to the centre
    const float px = fmaf(tc, dx, qx), py = fmaf(tc, dy, qy), pz = fmaf(tc, dz, qz);
    const float h2 = 1.f - fmaf(px, px, fmaf(py, py, pz * pz));          // (half chord)^2 in object units
    if (!(h2 >= 0.f)) return;
    const float s = __builtin_amdgcn_sqrtf(h2 * ia);
    const float tmax = (float)best.t;
    float root = tc - s;
    if (root < tmin) root = tc + s;                                       // sphere.cpp:60-66
    if (root < tmin || root > tmax) return;
    const double t = (double)root;
    // spheres accept t == t_max (sphere.cpp:83): among equal t the later primitive in visit order wins (tie_new_wins)
    if (t < best.t || best.prim < 0 || prim > best_order) {
        best = Cand{t, prim, 0, 0.0, 0.0};
        best_order = prim;
        best_incl = true;
    }
}

// One primitive against the ray; updates `best` when it is the new closest hit.
template <int F, bool COUNT = true>
__device__ __forceinline__ void test_prim(const DScene& S, const DParams& P, const Ray& r, int prim, int order, double tmin,
                                          Cand& best, int& best_order, bool& best_incl, Counters& C) {
    // A scene of plain spheres only (the kernel variant F says so) has one DXform per primitive in primitive order
    // (api.cpp: derive_arrays), so the header need not be fetched before the transform: one memory round trip per test.
    DPrim pr;
    if (F == RT_FEAT_SPHERE) { pr.type = RT_SHAPE_SPHERE; pr.data = prim; pr.material = 0; pr.file_index = 0; }
    else pr = S.prims[prim];
    if (COUNT && P.count_tests) C.prims++;
    double t = 0.0, bu = 0.0, bv = 0.0;
    int tri = 0;
    bool hit = false, incl = true;
    const double tmax = best.t;
    if (Feat<F>::sphere && pr.type == RT_SHAPE_SPHERE) {  // sphere.cpp:44-90
        const DXform& X = S.xforms[pr.data];
        ObjRay q = to_object(X, r);
        double a = dot(q.d, q.d), b = 2.0 * dot(q.o, q.d), c = dot(q.o, q.o) - 1.0;
        double disc = b * b - 4 * a * c;
        if (!(disc < 0)) {
            double sq = sqrt(disc);
            double root = (-b - sq) / (2.0 * a);
            hit = true;
            // sphere.cpp:60-66 tries the far root when the near one lies outside [tmin, tmax].  The far root is never smaller than
            // the near one, in floating point too (sq >= 0, and rounding and the division by the same 2a > 0 are monotone), so a
            // near root beyond tmax -- a sphere behind the closest hit so far, the usual miss -- settles it without the second
            // division; only a near root before tmin needs the far one.
            if (root < tmin) {
                root = (-b + sq) / (2.0 * a);
                if (root < tmin || root > tmax) hit = false;
            } else if (root > tmax) {
                hit = false;
            }
            t = root;
        }
    } else if (Feat<F>::quad && pr.type == RT_SHAPE_PLANE) {  // plane.cpp:137-177
        const DQuad& Q = S.quads[pr.data];
        D3 o0 = r.o - ld3(Q.vel) * r.time;
        double t1, u1, v1, t2, u2, v2;
        bool h1 = tri_test(o0, r.d, tmin, tmax, best.prim >= 0, P.eps_triangle, ld3(Q.v0), ld3(Q.e1), ld3(Q.e2), t1, u1, v1);
        bool h2 = tri_test(o0, r.d, tmin, tmax, best.prim >= 0, P.eps_triangle, ld3(Q.w0), ld3(Q.f1), ld3(Q.f2), t2, u2, v2);
        incl = false;
        if (h1 || h2) {
            hit = true;
            bool first = (h1 && h2) ? (t1 < t2) : h1;
            t = first ? t1 : t2;
            tri = first ? 0 : 1;
            bu = first ? u1 : u2;
            bv = first ? v1 : v2;
        }
    } else if (Feat<F>::cube && pr.type == RT_SHAPE_CUBE) {  // cube.cpp:25-89
        const DXform& X = S.xforms[pr.data];
        ObjRay q = to_object(X, r);
        double tn = -INFINITY, tf = INFINITY;
        const double o[3] = {q.o.x, q.o.y, q.o.z}, d[3] = {q.d.x, q.d.y, q.d.z};
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (!ok) break;
            double t0, t1;
            if (d[i] == 0.0) {
                if (o[i] < -1.0 || o[i] > 1.0) { ok = false; break; }
                t0 = -INFINITY;
                t1 = INFINITY;
            } else {
                double id = 1.0 / d[i];
                t0 = (-1.0 - o[i]) * id;
                t1 = (1.0 - o[i]) * id;
                if (id < 0.0) { double s = t0; t0 = t1; t1 = s; }
            }
            if (t0 > tn) tn = t0;
            if (t1 < tf) tf = t1;
            if (tn > tf) ok = false;
            else if (tf < 0.0) ok = false;
        }
        if (ok) {
            double th = tn;
            hit = true;
            if (th < tmin || th > tmax) {
                th = tf;
                if (th < tmin || th > tmax) hit = false;
            }
            t = th;
        }
    } else if (Feat<F>::complex_) {
        hit = march_complex(S, P, pr, r, tmin, tmax, t);
    }
    if (!hit) return;
    bool take;
    if (best.prim < 0) take = true;               // the test itself applied the range rule against the initial t_max
    else if (t < best.t) take = true;
    else if (t == best.t) take = tie_new_wins(incl, order, best_incl, best_order);
    else take = false;
    if (take) {
        best.t = t;
        best.prim = prim;
        best.tri = tri;
        best.bu = bu;
        best.bv = bv;
        best_order = order;
        best_incl = incl;
    }
}

// Slab test of aabb.cpp:4-38 in FP32 on a widened box (the root box; the node boxes use node4_test below).  It only has to be CONSERVATIVE: whenever the
// reference's FP64 test (running tmin/tmax over the three axes, reject on tmax <= tmin) accepts, this one accepts.
//   * the box was moved outward on the host by more than the FP32 rounding of the ray origin;
//   * every FP32 operation below has relative error <= 2^-24, so entry distances are lowered and exit distances
//     raised by 2^-20 relative before they are compared (the reciprocal of the direction carries up to 1.5 * 2^-23);
//   * tmin is rounded down and tmax up when converted;
//   * a NaN (inf - inf on an axis-parallel ray) is ignored by fmaxf/fminf, which can only widen the interval.
// The reference leaves its axis loop at the first failing axis; since tmin only grows and tmax only shrinks that is
// the same predicate as "final exit <= final entry", evaluated here without branches so that the four loads of a
// node are issued together (one memory round trip per node).
struct RayF {
    float inv[3], oi[3];   // 1/d and o * (1/d), per axis
    unsigned neg[3];       // all ones where the direction component is negative (the near plane is then the upper one)
};
__device__ __forceinline__ RayF make_rayf(const Ray& r) {
    RayF f;
    // FP32 reciprocal of the direction rounded to FP32 (v_rcp_f32, 1 ulp): relative error under 1.5 * 2^-23, which
    // scales every slab distance alike and is part of the relative slack of the box tests below.
    f.inv[0] = __builtin_amdgcn_rcpf((float)r.d.x); f.inv[1] = __builtin_amdgcn_rcpf((float)r.d.y); f.inv[2] = __builtin_amdgcn_rcpf((float)r.d.z);
    f.oi[0] = (float)r.o.x * f.inv[0]; f.oi[1] = (float)r.o.y * f.inv[1]; f.oi[2] = (float)r.o.z * f.inv[2];
    f.neg[0] = f.inv[0] < 0.f ? ~0u : 0u; f.neg[1] = f.inv[1] < 0.f ? ~0u : 0u; f.neg[2] = f.inv[2] < 0.f ? ~0u : 0u;
    return f;
}
// t = B * inv - o * inv with one FMA per plane; the extra rounding of o*inv is of the size of the origin's own FP32
// rounding, which the host pad covers four times over.  min/max order the two planes of an axis (no sign select).
// For an exactly axis-parallel ray (inv = +-inf) both products are NaN and the axis simply does not constrain.
// SLACK = false: the caller's box carries the host pad of the node boxes (host_bvh.cpp: node4_pad), which covers the relative
// roundings as plane displacements too -- |plane - o| is bounded by twice the scene's extent -- so that, as in node4_test, the two
// slack operations per interval end are left out (the per-primitive boxes of small scenes: four instructions of twenty-one).
template <bool SLACK = true>
__device__ __forceinline__ bool box_test32(const float* lo, const float* hi, const RayF& f, float tmin, float tmax, float& entry) {
    constexpr float kSlack = 9.53674316e-7f;  // 2^-20
    const float ax = fmaf(lo[0], f.inv[0], -f.oi[0]), bx = fmaf(hi[0], f.inv[0], -f.oi[0]);
    const float ay = fmaf(lo[1], f.inv[1], -f.oi[1]), by = fmaf(hi[1], f.inv[1], -f.oi[1]);
    const float az = fmaf(lo[2], f.inv[2], -f.oi[2]), bz = fmaf(hi[2], f.inv[2], -f.oi[2]);
    float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    if (SLACK) {
        tn = fmaf(-fabsf(tn), kSlack, tn);
        tf = fmaf(fabsf(tf), kSlack, tf);
    }
    tn = fmaxf(tn, tmin);   // a NaN interval end falls back to the ray's own range
    tf = fminf(tf, tmax);
    entry = tn;
    return tf >= tn;
}

// closest_hit_small: a primitive whose own (widened, FP32) box no lane's ray enters before its closest hit so far is
// skipped by the whole wave -- the conservative slab test of the hierarchy's boxes, applied to the primitives directly.
template <int F>
__device__ __forceinline__ bool closest_hit_small(const DScene& S, const DParams& P, const Ray& r, double tmin, double tmax, Cand& best,
                                                  Counters& C, bool any_hit, bool active = true) {
    C.rays += active ? 1u : 0u;
    best.t = tmax;
    best.prim = -1;
    best.tri = 0;
    best.bu = best.bv = 0.0;
    int best_order = 0;
    bool best_incl = false;
    bool live = active;
    const RayF rf = make_rayf(r);
    const float tmin32 = __double2float_rd(tmin);
    float tmax32 = __double2float_ru(tmax);   // follows best.t (rounded up)
    const int n = __builtin_amdgcn_readfirstlane(S.n_prims);
    for (int i = 0; i < n; ++i) {
        if (any_hit && __ballot(live) == 0ull) break;
        const int prim = P.use_bvh ? i : __builtin_amdgcn_readfirstlane(S.file_order[i]);
        bool enters = live;
        if (S.prim_boxes) {
            float e;
            if (P.count_tests) C.nodes += live ? 1u : 0u;
            enters = live && box_test32<false>(S.prim_boxes + 6 * prim, S.prim_boxes + 6 * prim + 3, rf, tmin32, tmax32, e);
        }
        if (__ballot(enters) == 0ull) continue;
        if (enters) {
            test_prim<F>(S, P, r, prim, i, tmin, best, best_order, best_incl, C);
            tmax32 = __double2float_ru(best.t);
            if (any_hit && best.prim >= 0) live = false;
        }
    }
    return best.prim >= 0;
}

// One 4-wide node (device_types.h: DNode4) pulled into registers with four 16-byte loads issued back to back.
typedef unsigned rt_u4 __attribute__((ext_vector_type(4)));
struct NodeRegs {
    float org[3], step[3];
    unsigned qlo[3], qhi[3];
    int child[4];
};
__device__ __forceinline__ NodeRegs load_node(const DNode4* n) {
    const rt_u4* p = reinterpret_cast<const rt_u4*>(n);
    const rt_u4 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
    NodeRegs r;
    r.org[0] = __uint_as_float(a0.x); r.org[1] = __uint_as_float(a0.y); r.org[2] = __uint_as_float(a0.z);
    r.step[0] = __uint_as_float(a0.w); r.step[1] = __uint_as_float(a3.z); r.step[2] = __uint_as_float(a3.w);
    r.qlo[0] = a1.x; r.qlo[1] = a1.y; r.qlo[2] = a1.z;
    r.qhi[0] = a1.w; r.qhi[1] = a2.x; r.qhi[2] = a2.y;
    r.child[0] = (int)a2.z; r.child[1] = (int)a2.w; r.child[2] = (int)a3.x; r.child[3] = (int)a3.y;
    return r;
}

// The four child boxes of a node against the ray.  Planes are org + q * 2^e on the node's grid, so the slab distance is
// t = q * (2^e * inv) + (org * inv - o * inv): two per-axis constants per node and one FMA per plane (issued as packed
// pairs: near and far plane of one axis).  Which plane is the near one follows from the sign of the direction, so the
// two plane words of an axis are swapped once per node instead of ordering every pair of distances.  Conservative like
// box_test32, but with ALL rounding errors in the host pad (host_bvh.cpp: node4_pad): a relative error eps of a slab
// distance t = (plane - o) / d is a plane moved by eps * |plane - o| <= eps * 2 * max|coordinate|, so an absolute pad
// covers it and the node loop carries no slack arithmetic (eight instructions per node less).  NaNs of axis-parallel
// rays drop out of min3/max3.  e[c] = entry distance of child c, +inf when missed/unused.
typedef float rt_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void node4_test(const NodeRegs& n, const RayF& f, float tmin, float tmax, float e[4]) {
#ifdef RT_NODE_SLACK
    constexpr float kSlack = 9.53674316e-7f;  // 2^-20 (A/B builds: the relative slack the pad replaces)
#endif
    rt_f2 a[3], b[3];
    unsigned qn[3], qf[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float ak = n.step[k] * f.inv[k];   // power of two: exact
        const float bk = fmaf(n.org[k], f.inv[k], -f.oi[k]);
        a[k] = rt_f2{ak, ak};
        b[k] = rt_f2{bk, bk};
        const unsigned d = (n.qlo[k] ^ n.qhi[k]) & f.neg[k];
        qn[k] = n.qlo[k] ^ d;
        qf[k] = n.qhi[k] ^ d;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        rt_f2 t[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const rt_f2 q{(float)((qn[k] >> (8 * c)) & 0xffu), (float)((qf[k] >> (8 * c)) & 0xffu)};
            t[k] = __builtin_elementwise_fma(q, a[k], b[k]);
        }
        float tn = fmaxf(fmaxf(t[0].x, t[1].x), t[2].x);
        float tf = fminf(fminf(t[0].y, t[1].y), t[2].y);
#ifdef RT_NODE_SLACK
        tn = fmaf(-fabsf(tn), kSlack, tn);
        tf = fmaf(fabsf(tf), kSlack, tf);
#endif
        tn = fmaxf(tn, tmin);
        tf = fminf(tf, tmax);
        // (an unused slot holds an inverted box: planes 255 / 0 on every axis, and the step is never zero, so tf < tn on every
        // axis the ray is not parallel to -- there is one -- and the slot's reference is never looked at)
        e[c] = (tf >= tn) ? tn : INFINITY;
    }
}

// One traversal step at an interior node: returns the nearest child entered and defers the other entered children, farthest
// first, so that they are popped nearest first.  When no child is entered: POP = true returns the next deferred entry
// (closest_hit), POP = false returns kNoChild to callers that pop themselves (k_traverse: ONE pop loop per step, shared by the
// lanes whose node had no child for them and the lanes that have just parked a leaf).
constexpr int kNoChild = 0x7ffffffe;
template <bool POP>
__device__ __forceinline__ int node4_step(const NodeRegs& n, const RayF& f, float tmin, float tmax, TravStack& stack, int& sp) {
    float e[4];
    node4_test(n, f, tmin, tmax, e);
    int c[4] = {n.child[0], n.child[1], n.child[2], n.child[3]};
#ifdef RT_NO_CHILD_SORT
    // Experiment (what the ordering is worth): the nearest entered child is still taken first, but the other entered
    // children are deferred in SLOT order instead of sorted by entry distance -- no 5-comparator network, no compaction of the
    // misses, three conditional pushes.  Same closest hit (order independent); the entry distances still cull at pop time.
    {
        const float m = fminf(fminf(e[0], e[1]), fminf(e[2], e[3]));
        if (m == INFINITY) return POP ? stack.pop(sp, tmax) : kNoChild;
        const int k = e[0] == m ? 0 : e[1] == m ? 1 : e[2] == m ? 2 : 3;
        int first = c[0];
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            if (j == k) first = c[j];
            else if (e[j] != INFINITY) stack.push(sp, c[j], e[j]);
        }
        return first;
    }
#endif
#define RT_CSWAP(i, j)                                                      \
    {                                                                       \
        const bool sw = e[j] < e[i];                                        \
        const float te = sw ? e[j] : e[i], tf_ = sw ? e[i] : e[j];          \
        const int tc = sw ? c[j] : c[i], td = sw ? c[i] : c[j];             \
        e[i] = te; e[j] = tf_; c[i] = tc; c[j] = td;                        \
    }
    RT_CSWAP(0, 1) RT_CSWAP(2, 3) RT_CSWAP(0, 2) RT_CSWAP(1, 3) RT_CSWAP(1, 2)
#undef RT_CSWAP
    if (e[0] == INFINITY) return POP ? stack.pop(sp, tmax) : kNoChild;
    const int deferred = (e[1] != INFINITY) + (e[2] != INFINITY) + (e[3] != INFINITY);
    stack.push_sorted(sp, deferred, c[1], e[1], c[2], e[2], c[3], e[3]);
    return c[0];
}
__device__ __forceinline__ int node4_children(const NodeRegs& n) {
    return (n.child[0] != kEmptyChild) + (n.child[1] != kEmptyChild) + (n.child[2] != kEmptyChild) + (n.child[3] != kEmptyChild);
}

// Closest hit == `world.intersect(r, tmin, tmax, rec)` (hittable_list.cpp:8-33 over bvh.cpp:101-114).
// Traversal is front-to-back with a per-lane stack of deferred children; the hit found equals the reference's
// left-then-right recursion because closest-hit selection is order independent (ties: tie_new_wins).
// any_hit: the caller only asks WHETHER something lies in (tmin, tmax) -- a shadow ray in a scene without transmitting
// materials (shading.h:52-55: an opaque occluder gives 0 whichever it is) -- so the first hit found ends the traversal.
template <int F>
__device__ bool closest_hit(const DScene& S, const DParams& P, const Ray& r, double tmin, double tmax, Cand& best, Counters& C,
                            TravStack& stack, bool any_hit = false) {
    C.rays++;
    best.t = tmax;
    best.prim = -1;
    best.tri = 0;
    best.bu = best.bv = 0.0;
    int best_order = 0;
    bool best_incl = false;
    if (S.n_prims <= 0) return false;
    if (!P.use_bvh) {  // --no-bvh: the list in scene-file order
        for (int i = 0; i < S.n_prims; ++i) {
            test_prim<F>(S, P, r, S.file_order[i], i, tmin, best, best_order, best_incl, C);
            if (any_hit && best.prim >= 0) return true;
        }
        return best.prim >= 0;
    }
    const RayF rf = make_rayf(r);
    const float tmin32 = __double2float_rd(tmin);
    float e;
    if (P.count_tests) C.nodes++;
    float tmax32 = __double2float_ru(tmax);   // follows best.t (rounded up)
    if (!box_test32(S.root_lo, S.root_hi, rf, tmin32, tmax32, e)) return false;
    int sp = 0;
    int cur = S.root_ref;
    constexpr int kDone = 0x7fffffff;
    // "while-while" traversal: every lane first descends interior nodes until it holds a leaf (or runs out of
    // work); lanes that found their leaf early idle until the whole wave has left the inner loop, so that the
    // expensive FP64 primitive tests below run with many lanes active instead of once per node step.
    while (cur != kDone) {
        while (static_cast<unsigned>(cur) < static_cast<unsigned>(kDone)) {  // 0 <= cur < kDone: interior node
            const NodeRegs n = load_node(&S.nodes[cur]);
            if (P.count_tests) { C.nodes += node4_children(n); C.visits++; }
            cur = node4_step<true>(n, rf, tmin32, tmax32, stack, sp);
        }
        if (cur < 0) {  // leaf: one or two consecutive primitives
            int code = ~cur;
            int first = code >> 1;
            test_prim<F>(S, P, r, first, first, tmin, best, best_order, best_incl, C);
            if (code & 1) test_prim<F>(S, P, r, first + 1, first + 1, tmin, best, best_order, best_incl, C);
            if (any_hit && best.prim >= 0) return true;
            tmax32 = __double2float_ru(best.t);
            cur = stack.pop(sp, tmax32);
        }
    }
    return best.prim >= 0;
}

// Scenes of a handful of primitives (the reference's own example scenes have 4 to 10): every lane of the wave tests
// primitive 0, 1, 2 ... in step, so the primitive's record is the same for all lanes -- its loads are scalar, its type
// switch is a scalar branch, nothing diverges and no hierarchy, stack or scheduling is involved.  Same closest hit as
// the traversal (the tie rule makes the result independent of the testing order); `visit` maps step i to (primitive,
// visit order) as the reference would meet them: the BVH's depth-first order, or the file order with --no-bvh.
constexpr int kSmallScenePrims = 16;
// (closest_hit_small itself stands with the box tests above)

// The whole description of such a scene fits a few KB: the kernels for small scenes copy it into LDS once per workgroup
// and hand the device functions a DScene whose arrays point there, so that the chains of dependent loads of a ray
// (primitive -> transform -> material, light) cost LDS latency instead of an L2 round trip each.
constexpr int kSmallSceneMaterials = 16, kSmallSceneLights = 8;
struct SmallSceneLds {
    float boxes[kSmallScenePrims * 6];
    DPrim prims[kSmallScenePrims];
    DXform xforms[kSmallScenePrims];
    DQuad quads[kSmallScenePrims];
    DMaterial materials[kSmallSceneMaterials];
    DLight lights[kSmallSceneLights];
};
__device__ __forceinline__ void lds_copy_words(void* dst, const void* src, int bytes) {
    unsigned* d = static_cast<unsigned*>(dst);
    const unsigned* s = static_cast<const unsigned*>(src);
    for (int i = threadIdx.x; i < bytes / 4; i += blockDim.x) d[i] = s[i];
}
// Called by every thread of the workgroup (it contains a barrier); S.n_prims <= kSmallScenePrims is the caller's business.
__device__ __forceinline__ DScene stage_small_scene(const DScene& S, SmallSceneLds& L) {
    DScene T = S;
    lds_copy_words(L.prims, S.prims, S.n_prims * (int)sizeof(DPrim));
    lds_copy_words(L.xforms, S.xforms, S.n_xforms * (int)sizeof(DXform));
    lds_copy_words(L.quads, S.quads, S.n_quads * (int)sizeof(DQuad));
    T.prims = L.prims;
    T.xforms = L.xforms;
    T.quads = L.quads;
    if (S.prim_boxes) {
        lds_copy_words(L.boxes, S.prim_boxes, S.n_prims * 6 * (int)sizeof(float));
        T.prim_boxes = L.boxes;
    }
    if (S.n_materials <= kSmallSceneMaterials) {
        lds_copy_words(L.materials, S.materials, S.n_materials * (int)sizeof(DMaterial));
        T.materials = L.materials;
    }
    if (S.n_lights <= kSmallSceneLights) {
        lds_copy_words(L.lights, S.lights, S.n_lights * (int)sizeof(DLight));
        T.lights = L.lights;
    }
    __syncthreads();
    return T;
}

template <bool SMALL> struct SceneStage;
template <> struct SceneStage<true> {
    SmallSceneLds lds;
    __device__ __forceinline__ DScene stage(const DScene& S) { return stage_small_scene(S, lds); }
};
template <> struct SceneStage<false> {
    __device__ __forceinline__ DScene stage(const DScene& S) { return S; }
};

// ------------------------------------------------------------------------------------------ hit record
struct Hit {
    double t;
    D3 p, n;
    double u, v;
    int material;
    bool front;
};
__device__ __forceinline__ void set_face_normal(Hit& h, const Ray& r, D3 outward) {  // hittable.h:35-42
    h.front = dot(r.d, outward) < 0.0;
    h.n = h.front ? outward : -outward;
}

// sphere.cpp:116-167 == cube.cpp:178-229
__device__ D3 bump_nearest(const DTexture& T, D3 N, double u, double v) {
    D3 Y = mk(0, 1, 0), Tn;
    if (fabs(dot(N, Y)) > 0.999) Tn = mk(1, 0, 0);
    else Tn = normalize(cross(Y, N));
    D3 B = normalize(cross(N, Tn));
    int w = T.width, h = T.height;
    int x = to_int(u * (w - 1)), y = to_int((1.0 - v) * (h - 1));
    double hc = tex_intensity_at(T, x, y), hu = tex_intensity_at(T, x + 1, y), hv = tex_intensity_at(T, x, y + 1);
    double bu = (hu - hc) * w, bv = (hv - hc) * h;
    return normalize(N + (Tn * bu + B * bv) * 0.0075);
}

// Everything the shapes' intersect() write into the HitRecord, recomputed for the winning candidate only.
template <int F>
__device__ void fill_hit(const DScene& S, const DParams& P, const Ray& r, const Cand& c, Hit& h) {
    const DPrim pr = S.prims[c.prim];
    const DMaterial& M = S.materials[pr.material];
    h.t = c.t;
    h.p = r.o + (c.t * r.d);  // ray.h:30-32
    h.material = pr.material;
    h.u = 0.0;
    h.v = 0.0;
    const bool need_uv = Feat<F>::tex && ((M.texture >= 0) || (M.bump >= 0));
    if (Feat<F>::sphere && pr.type == RT_SHAPE_SPHERE) {  // sphere.cpp:92-171
        const DXform& X = S.xforms[pr.data];
        ObjRay q = to_object(X, r);
        D3 on = (q.o + (c.t * q.d)) - mk(0, 0, 0);
        D3 outward = normal_to_world(X, on);
        if (need_uv) {
            sphere_uv(normalize(on), h.u, h.v);
            if (M.bump >= 0) outward = bump_nearest(S.textures[M.bump], outward, h.u, h.v);
        }
        set_face_normal(h, r, outward);
    } else if (Feat<F>::quad && pr.type == RT_SHAPE_PLANE) {  // plane.cpp:179-243
        const DQuad& Q = S.quads[pr.data];
        D3 n = ld3(Q.n), outward = n;
        if (c.tri == 0) { h.u = c.bu; h.v = c.bv; }
        else { h.u = 1.0 - c.bv; h.v = c.bu + c.bv; }
        if (Feat<F>::tex && M.bump >= 0) {
            const DTexture& T = S.textures[M.bump];
            D3 Tn = normalize(ld3(Q.e1)), B = normalize(ld3(Q.e2));
            int w = T.width, hh = T.height;
            double sx = 1.0 / w, sy = 1.0 / hh;
            double hc = px_intensity(tex_bilinear(T, h.u, h.v));
            double hu = px_intensity(tex_bilinear(T, h.u + sx, h.v));
            double hv = px_intensity(tex_bilinear(T, h.u, h.v + sy));
            double bu = (hu - hc) * w, bv = (hv - hc) * hh;
            bu = smax(-100.0, smin(100.0, bu));
            bv = smax(-100.0, smin(100.0, bv));
            outward = normalize(n + (Tn * bu + B * bv) * 0.0075);
        }
        set_face_normal(h, r, outward);
    } else if (Feat<F>::cube && pr.type == RT_SHAPE_CUBE) {  // cube.cpp:91-231
        const DXform& X = S.xforms[pr.data];
        ObjRay q = to_object(X, r);
        D3 p = q.o + q.d * c.t, on;
        double u, v;
        cube_face_uv(p, false, on, u, v);
        D3 outward = normal_to_world(X, on);
        h.u = u;
        h.v = v;
        if (Feat<F>::tex && M.bump >= 0) outward = bump_nearest(S.textures[M.bump], outward, u, v);
        set_face_normal(h, r, outward);
    } else if (Feat<F>::complex_) {  // complex_*.cpp: uv at the marched point, central-difference normal (step 0.005)
        const DXform& X = S.xforms[pr.data];
        ObjRay q = to_object(X, r);
        D3 p = q.o + q.d * c.t;
        const double D = X.disp;
        (void)sdf_complex(pr.type, S, M, D, p, &h.u, &h.v);
        const double e = 0.005;
        double gx = sdf_complex(pr.type, S, M, D, p + mk(e, 0, 0), nullptr, nullptr) - sdf_complex(pr.type, S, M, D, p - mk(e, 0, 0), nullptr, nullptr);
        double gy = sdf_complex(pr.type, S, M, D, p + mk(0, e, 0), nullptr, nullptr) - sdf_complex(pr.type, S, M, D, p - mk(0, e, 0), nullptr, nullptr);
        double gz = sdf_complex(pr.type, S, M, D, p + mk(0, 0, e), nullptr, nullptr) - sdf_complex(pr.type, S, M, D, p - mk(0, 0, e), nullptr, nullptr);
        D3 ln = normalize(mk(gx, gy, gz));
        set_face_normal(h, r, normal_to_world(X, ln));
    }
}

// ------------------------------------------------------------------------------------------ integrator (tracer.h)
__device__ __forceinline__ D3 reflect(D3 V, D3 N) { return V - 2 * dot(V, N) * N; }  // tracer.h:57-60

// camera.cpp:89-135
__device__ Ray generate_ray(const DCamera& c, float px, float py, double time, Rng& rng) {
    double uc = (0.5 - px) * c.sensor_width, vc = (py - 0.5) * c.sensor_height;
    D3 dir = normalize((c.focal_length * ld3(c.w)) + (uc * ld3(c.u)) + (vc * ld3(c.v)));
    if (c.aperture_radius <= 0.0) return Ray{ld3(c.location), dir, time};
    D3 focal_point = ld3(c.location) + dir * c.focal_distance;
    D3 disk = rng.in_unit_disk() * c.aperture_radius;
    D3 origin = ld3(c.location) + (ld3(c.u) * disk.x + ld3(c.v) * disk.y);
    return Ray{origin, normalize(focal_point - origin), time};
}

// tracer.h:23-54
__device__ D3 tonemap(D3 c, int mode) {
    if (mode == RT_TONEMAP_REINHARD) return mk(c.x / (1.0 + c.x), c.y / (1.0 + c.y), c.z / (1.0 + c.z));
    if (mode == RT_TONEMAP_ACES) {
        auto f = [](double x) { return (x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14); };
        return mk(f(c.x), f(c.y), f(c.z));
    }
    if (mode == RT_TONEMAP_FILMIC) {
        auto f = [](double x) {
            const double A = 0.15, B = 0.50, Cc = 0.10, D = 0.20, E = 0.02, F = 0.30;
            return ((x * (A * x + Cc * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
        };
        double ws = 1.0 / f(11.2);
        return mk(f(c.x) * ws, f(c.y) * ws, f(c.z) * ws);
    }
    return c;
}
__device__ __forceinline__ unsigned char quantise(double v) {  // tracer.h:297-308 (a NaN channel becomes 255)
    double c = smax(0.0, smin(1.0, v));
    return (unsigned char)(255 * c);
}

}  // namespace

}  // namespace rt
