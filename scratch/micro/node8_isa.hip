// Static instruction counts of one node step, for DESIGN section 6's comparison of the 4-wide sorted node with an 8-wide,
// octant-ordered one (never launched; compiled with scratch/kernel_regs.sh and counted with scratch/isa_mix.py).
//   k_step4: the node loop's arithmetic as it is: four 16-byte loads, node4_test, the 5-comparator sort of (entry, child).
//   k_step8: eight 16-byte loads of a 128-byte node (origin, steps, 8 x 6 plane bytes, 8 references), the same packed-FMA slab
//            test for eight children, the hit mask in visiting order (bit = slot ^ ray octant), the first child's reference.
// Neither holds the stack traffic: k_traverse's push of up to three sorted pairs and its pop are in the real kernel's count; the
// 8-wide form would push one (node, mask) pair per step and re-read a child reference at each pop.
#include <hip/hip_runtime.h>
#include "../../raytracer_amd/csrc/device_rt.h"
using namespace rt;

extern "C" __global__ void k_step0(const DNode4* nodes, const int* which, const RayF* rays, float* out_e, int* out_c) {   // the frame both share
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const RayF f = rays[i];
    out_e[i] = f.inv[0] + f.inv[1] + f.inv[2] + f.oi[0] + f.oi[1] + f.oi[2];
    out_c[i] = (int)(f.neg[0] ^ f.neg[1] ^ f.neg[2]) + which[i];
}

extern "C" __global__ void k_step4(const DNode4* nodes, const int* which, const RayF* rays, float* out_e, int* out_c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const RayF f = rays[i];
    const NodeRegs n = load_node(nodes + which[i]);
    float e[4];
    node4_test(n, f, 0.0f, 1e5f, e);
    int c[4] = {n.child[0], n.child[1], n.child[2], n.child[3]};
#define CSWAP(a, b) { const bool sw = e[b] < e[a]; const float te = sw ? e[b] : e[a], tf = sw ? e[a] : e[b]; const int tc = sw ? c[b] : c[a], td = sw ? c[a] : c[b]; e[a] = te; e[b] = tf; c[a] = tc; c[b] = td; }
    CSWAP(0, 1) CSWAP(2, 3) CSWAP(0, 2) CSWAP(1, 3) CSWAP(1, 2)
    for (int k = 0; k < 4; ++k) { out_e[4 * i + k] = e[k]; out_c[4 * i + k] = c[k]; }
}

struct alignas(128) DNode8 {
    float org[3];
    float step[3];
    unsigned qlo[3][2];   // [axis][children 0-3, 4-7]
    unsigned qhi[3][2];
    int child[8];
    unsigned pad[6];
};
static_assert(sizeof(DNode8) == 128, "");

extern "C" __global__ void k_step8(const DNode8* nodes, const int* which, const RayF* rays, unsigned* out_mask, int* out_c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const RayF f = rays[i];
    const unsigned oct = (f.neg[0] & 1u) | (f.neg[1] & 2u) | (f.neg[2] & 4u);
    const rt_u4* p = reinterpret_cast<const rt_u4*>(nodes + which[i]);
    rt_u4 w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = p[k];
    const unsigned* u = reinterpret_cast<const unsigned*>(w);
    rt_f2 a[3], b[3];
    unsigned qn[3][2], qf[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float ak = __uint_as_float(u[3 + k]) * f.inv[k];
        const float bk = fmaf(__uint_as_float(u[k]), f.inv[k], -f.oi[k]);
        a[k] = rt_f2{ak, ak};
        b[k] = rt_f2{bk, bk};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned lo = u[6 + 2 * k + h], hi = u[12 + 2 * k + h];
            const unsigned d = (lo ^ hi) & f.neg[k];
            qn[k][h] = lo ^ d;
            qf[k][h] = hi ^ d;
        }
    }
    unsigned mask = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        rt_f2 t[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const rt_f2 q{(float)((qn[k][c >> 2] >> (8 * (c & 3))) & 0xffu), (float)((qf[k][c >> 2] >> (8 * (c & 3))) & 0xffu)};
            t[k] = __builtin_elementwise_fma(q, a[k], b[k]);
        }
        float tn = fmaxf(fmaxf(t[0].x, t[1].x), t[2].x);
        float tf = fminf(fminf(t[0].y, t[1].y), t[2].y);
        tn = fmaxf(tn, 0.0f);
        tf = fminf(tf, 1e5f);
        mask |= (tf >= tn) ? (1u << c) : 0u;
    }
    // bit p of the visiting-order mask = slot p ^ oct: three conditional exchanges of bit groups
    unsigned m = mask;
    m = (oct & 1u) ? (((m & 0x55u) << 1) | ((m >> 1) & 0x55u)) : m;
    m = (oct & 2u) ? (((m & 0x33u) << 2) | ((m >> 2) & 0x33u)) : m;
    m = (oct & 4u) ? (((m & 0x0fu) << 4) | ((m >> 4) & 0x0fu)) : m;
    const int first = m ? (__ffs(m) - 1) ^ (int)oct : 0;
    int c = (int)u[18];
#pragma unroll
    for (int k = 1; k < 8; ++k) c = first == k ? (int)u[18 + k] : c;
    out_mask[i] = m;
    out_c[i] = c;
}
