// Issue cost of single VALU instruction kinds at 4 waves per SIMD (independent streams of 8 registers).
#include <hip/hip_runtime.h>
#include <cstdio>
#define STREAM8(INS) asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m))
#define STREAM8_3(INS) asm volatile(INS " %0, %0, %8, %9\n" INS " %1, %1, %8, %9\n" INS " %2, %2, %8, %9\n" INS " %3, %3, %8, %9\n" INS " %4, %4, %8, %9\n" INS " %5, %5, %8, %9\n" INS " %6, %6, %8, %9\n" INS " %7, %7, %8, %9\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c))
template <int OP>
__global__ void __launch_bounds__(64) k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 1e-9f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) STREAM8("v_add_f32");
            if (OP == 1) STREAM8("v_mul_f32");
            if (OP == 2) STREAM8("v_max_f32");
            if (OP == 3) STREAM8_3("v_max3_f32");
            if (OP == 4) STREAM8("v_xor_b32");
            if (OP == 5) STREAM8("v_and_b32");
            if (OP == 6) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                                      "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
            if (OP == 7) asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                                      "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
            if (OP == 8) STREAM8("v_add_u32");
            if (OP == 9) STREAM8("v_lshlrev_b32");
            if (OP == 10) STREAM8_3("v_fma_f32");
            if (OP == 11) STREAM8("v_min_u32");
            if (OP == 12) STREAM8_3("v_lshl_or_b32");
            if (OP == 13) asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int OP> void run(const char* name, float* out, int cus) {
    const int iters = 20000, wps = 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(cus * 4 * wps), dim3(64), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(cus * 4 * wps), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-16s %.2f cycles per wave-instruction per SIMD (4 waves per SIMD, 2.4 GHz assumed)\n", name, ms * 1e-3 * 2.4e9 / ((double)iters * 64.0 * wps));
}
int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* out;
    hipMalloc(&out, (size_t)cus * 16 * 64 * sizeof(float));
    run<10>("v_fma_f32", out, cus); run<0>("v_add_f32", out, cus); run<1>("v_mul_f32", out, cus); run<2>("v_max_f32", out, cus); run<3>("v_max3_f32", out, cus);
    run<4>("v_xor_b32", out, cus); run<5>("v_and_b32", out, cus); run<6>("v_cndmask_b32", out, cus); run<7>("v_cmp_lt_f32", out, cus);
    run<8>("v_add_u32", out, cus); run<9>("v_lshlrev_b32", out, cus); run<11>("v_min_u32", out, cus); run<12>("v_lshl_or_b32", out, cus); run<13>("v_mov_b32", out, cus);
    return 0;
}
