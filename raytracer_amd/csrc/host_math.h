// Host-side FP64 vector / matrix helpers for scene construction (never used on the device).
// Semantics follow the reference's math layer: Code/utilities/vector3.h:15-103 and
// Code/utilities/matrix4x4.cpp:4-152 (row-major, column vectors, point transform with w-divide).
#pragma once
#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace rt {

struct V3 {
    double x = 0, y = 0, z = 0;
    V3() = default;
    V3(double a, double b, double c) : x(a), y(b), z(c) {}
    explicit V3(const double* p) : x(p[0]), y(p[1]), z(p[2]) {}
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double length(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
// vector3.h:92-102: lengths <= 1e-6 normalise to the zero vector.
inline V3 normalize(V3 a) {
    double len = length(a);
    if (len > 1e-6) return {a.x / len, a.y / len, a.z / len};
    return {};
}
inline void store(double* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

struct M4 {
    double m[4][4];
    M4() {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) m[i][j] = (i == j) ? 1.0 : 0.0;
    }
};

// matrix4x4.cpp:66-77 — plain triple loop, accumulation order k = 0..3.
inline M4 mul(const M4& a, const M4& b) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}

inline M4 translation(V3 t) { M4 r; r.m[0][3] = t.x; r.m[1][3] = t.y; r.m[2][3] = t.z; return r; }
inline M4 scaling(V3 s) { M4 r; r.m[0][0] = s.x; r.m[1][1] = s.y; r.m[2][2] = s.z; return r; }
inline M4 rotation_x(double a) { M4 r; double c = std::cos(a), s = std::sin(a); r.m[1][1] = c; r.m[1][2] = -s; r.m[2][1] = s; r.m[2][2] = c; return r; }
inline M4 rotation_y(double a) { M4 r; double c = std::cos(a), s = std::sin(a); r.m[0][0] = c; r.m[0][2] = s; r.m[2][0] = -s; r.m[2][2] = c; return r; }
inline M4 rotation_z(double a) { M4 r; double c = std::cos(a), s = std::sin(a); r.m[0][0] = c; r.m[0][1] = -s; r.m[1][0] = s; r.m[1][1] = c; return r; }

// Inverse by the classical adjugate: inv[j][i] = (-1)^(i+j) * minor(i,j) / det  (matrix4x4.cpp:116-152 computes the
// same adjugate with the 24-term expansion written out; results agree to rounding).  Singular => throws, as there.
inline M4 inverse(const M4& a) {
    auto minor3 = [&](int skip_r, int skip_c) {
        double s[3][3];
        int rr = 0;
        for (int r = 0; r < 4; ++r) {
            if (r == skip_r) continue;
            int cc = 0;
            for (int c = 0; c < 4; ++c) {
                if (c == skip_c) continue;
                s[rr][cc++] = a.m[r][c];
            }
            ++rr;
        }
        return s[0][0] * (s[1][1] * s[2][2] - s[1][2] * s[2][1]) - s[0][1] * (s[1][0] * s[2][2] - s[1][2] * s[2][0]) +
               s[0][2] * (s[1][0] * s[2][1] - s[1][1] * s[2][0]);
    };
    M4 adj;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double c = minor3(i, j);
            adj.m[j][i] = ((i + j) & 1) ? -c : c;
        }
    double det = a.m[0][0] * adj.m[0][0] + a.m[0][1] * adj.m[1][0] + a.m[0][2] * adj.m[2][0] + a.m[0][3] * adj.m[3][0];
    if (det == 0.0) throw std::runtime_error("Matrix is singular and cannot be inverted.");
    double inv_det = 1.0 / det;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) adj.m[i][j] *= inv_det;
    return adj;
}

// matrix4x4.cpp:80-93 — point transform: homogeneous divide whenever w is neither 0 nor 1.
inline V3 transform_point(const M4& a, V3 v) {
    double x = a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z + a.m[0][3];
    double y = a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z + a.m[1][3];
    double z = a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z + a.m[2][3];
    double w = a.m[3][0] * v.x + a.m[3][1] * v.y + a.m[3][2] * v.z + a.m[3][3];
    if (w != 0.0 && w != 1.0) return {x / w, y / w, z / w};
    return {x, y, z};
}

inline M4 from_row_major(const double* p) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = p[i * 4 + j];
    return r;
}
inline void to_row_major(const M4& a, double* p) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) p[i * 4 + j] = a.m[i][j];
}

struct Box {
    V3 lo, hi;
};
// aabb.cpp:41-56
inline Box combine(const Box& a, const Box& b) {
    return {{std::min(a.lo.x, b.lo.x), std::min(a.lo.y, b.lo.y), std::min(a.lo.z, b.lo.z)},
            {std::max(a.hi.x, b.hi.x), std::max(a.hi.y, b.hi.y), std::max(a.hi.z, b.hi.z)}};
}
// aabb.h:24-31
inline void grow(Box& b, V3 p) {
    b.lo.x = std::min(b.lo.x, p.x); b.lo.y = std::min(b.lo.y, p.y); b.lo.z = std::min(b.lo.z, p.z);
    b.hi.x = std::max(b.hi.x, p.x); b.hi.y = std::max(b.hi.y, p.y); b.hi.z = std::max(b.hi.z, p.z);
}

}  // namespace rt
