// TEST INFRASTRUCTURE ONLY (oracle). Not part of the product path.
//
// Force-included (-include) into every translation unit of the *unmodified* reference sources when
// they are compiled in place from /root/reference/Code by oracle/Makefile.  It pre-empts the include
// guard of Code/utilities/random_utils.h (the build passes -DB216602_RANDOM_UTILS_H) and supplies
// reproducible replacements for the reference's non-reproducible RNG
// (random_utils.h:13-35 uses a thread_local mt19937 seeded from random_device; camera.cpp:17-18 uses rand()).
//
// Modes (environment variable RT_ORACLE_RNG, read once per thread):
//   unset / "d0"   every uniform = 0.5, unit-ball sample = (0,0,0), lens sample = (0,0)
//                  => pixel-centre sampling, ray time = 0.5*shutter, point lights, mirror bounces.
//   "seed:<n>"     splitmix64 stream seeded with n (run single-threaded for reproducible output).
#pragma once
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <random>
#include <algorithm>
#include <numeric>
#include <string>
#include "utilities/vector3.h"

namespace oracle_shim {
struct Stream {
    bool d0 = true;
    uint64_t s = 0;
    Stream() {
        const char* e = std::getenv("RT_ORACLE_RNG");
        if (e && std::strncmp(e, "seed:", 5) == 0) { d0 = false; s = std::strtoull(e + 5, nullptr, 10) * 0x9E3779B97F4A7C15ull + 1; }
    }
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
};
inline Stream& stream() { static thread_local Stream st; return st; }
}  // namespace oracle_shim

inline double random_double() {
    auto& st = oracle_shim::stream();
    if (st.d0) return 0.5;
    return (double)(st.next() >> 11) * (1.0 / 9007199254740992.0);
}
inline double random_double(double a, double b) { return a + (b - a) * random_double(); }
inline int random_int(int a, int b) {
    auto& st = oracle_shim::stream();
    if (st.d0) return (a + b) / 2;
    return a + (int)(st.next() % (uint64_t)(b - a + 1));
}
inline Vector3 random_in_unit_sphere() {
    auto& st = oracle_shim::stream();
    if (st.d0) return Vector3(0, 0, 0);
    while (true) {
        Vector3 p(random_double() * 2 - 1, random_double() * 2 - 1, random_double() * 2 - 1);
        if (p.dot(p) < 1.0) return p;
    }
}
inline int oracle_rand() {
    auto& st = oracle_shim::stream();
    if (st.d0) return RAND_MAX / 2 + 1;  // (r/(RAND_MAX+1.0))*2-1 == 0 exactly
    return (int)(st.next() >> 33);       // 31 uniform bits, 0..RAND_MAX
}
#define rand oracle_rand
