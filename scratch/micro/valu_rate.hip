// VALU issue rate per SIMD against waves per SIMD: independent v_fma_f32 / v_fma_f64 / v_pk_fma_f32 / v_cvt_f32_ubyte0 streams.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ void __launch_bounds__(64) k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    const float m = 1.0000001f, c = 1e-9f;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            }
        } else if (OP == 1) {
            const double dm = 1.0000001, dc = 1e-9;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dm), "v"(dc));
            }
        } else if (OP == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte2 %2, %2\n v_cvt_f32_ubyte3 %3, %3\n"
                             "v_cvt_f32_ubyte0 %4, %4\n v_cvt_f32_ubyte1 %5, %5\n v_cvt_f32_ubyte2 %6, %6\n v_cvt_f32_ubyte3 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        } else if (OP == 3) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d4), "v"(d5));
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_max3_f32 %1, %1, %8, %9\n v_xor_b32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_max3_f32 %5, %5, %8, %9\n v_xor_b32 %6, %6, %8\n v_min_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc");
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}
template <int OP> void run(const char* name, float* out, int cus) {
    const int iters = 20000;
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        const int blocks = cus * 4 * wps;   // one wave per block: wps waves per SIMD when spread evenly
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 64.0 * wps;
        printf("%-18s %d waves/SIMD: %.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, wps, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    }
}
int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* out;
    hipMalloc(&out, (size_t)cus * 4 * 8 * 64 * sizeof(float));
    run<0>("v_fma_f32", out, cus);
    run<1>("v_fma_f64", out, cus);
    run<2>("v_cvt_f32_ubyteN", out, cus);
    run<3>("v_pk_fma_f32", out, cus);
    run<4>("cndmask/max3/xor/min", out, cus);
    hipFree(out);
    return 0;
}
