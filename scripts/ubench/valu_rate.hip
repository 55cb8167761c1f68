// Microbenchmark: issue rate of f32 VALU forms on gfx950 (which ops are "full rate"?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    v2f sv = {s, s};
    unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 ^ 0x1234567;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 8 independent v_mul_f32
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 1) {  // 8 independent v_pk_mul_f32
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(sv));
        } else if (MODE == 2) {  // 8 v_cvt_f32_ubyte
            asm volatile("v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %8\n v_cvt_f32_ubyte2 %2, %8\n v_cvt_f32_ubyte3 %3, %8\n"
                         "v_cvt_f32_ubyte0 %4, %9\n v_cvt_f32_ubyte1 %5, %9\n v_cvt_f32_ubyte2 %6, %9\n v_cvt_f32_ubyte3 %7, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u0), "v"(u1));
        } else if (MODE == 3) {  // 8 v_add_f32
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 4) {  // 8 v_pk_add_f32
            asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(sv));
        } else if (MODE == 5) {  // 8 v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                         "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 6) {  // 8 v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                         "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(sv));
        } else if (MODE == 7) {  // v_med3_f32
            asm volatile("v_med3_f32 %0, %0, %8, %8\n v_med3_f32 %1, %1, %8, %8\n v_med3_f32 %2, %2, %8, %8\n v_med3_f32 %3, %3, %8, %8\n"
                         "v_med3_f32 %4, %4, %8, %8\n v_med3_f32 %5, %5, %8, %8\n v_med3_f32 %6, %6, %8, %8\n v_med3_f32 %7, %7, %8, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 8) {  // v_cvt_u32_f32
            asm volatile("v_cvt_u32_f32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_cvt_u32_f32 %2, %2\n v_cvt_u32_f32 %3, %3\n"
                         "v_cvt_u32_f32 %4, %4\n v_cvt_u32_f32 %5, %5\n v_cvt_u32_f32 %6, %6\n v_cvt_u32_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 9) {  // v_cvt_pk_u8_f32
            asm volatile("v_cvt_pk_u8_f32 %0, %8, 0, %0\n v_cvt_pk_u8_f32 %1, %8, 1, %1\n v_cvt_pk_u8_f32 %2, %8, 2, %2\n v_cvt_pk_u8_f32 %3, %8, 3, %3\n"
                         "v_cvt_pk_u8_f32 %4, %8, 0, %4\n v_cvt_pk_u8_f32 %5, %8, 1, %5\n v_cvt_pk_u8_f32 %6, %8, 2, %6\n v_cvt_pk_u8_f32 %7, %8, 3, %7\n"
                         : "+v"(u0), "+v"(u1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y + u0 + u1;
}

template <int MODE> void run(const char *name, int lane_ops_per_instr) {
    const int blocks = 256 * 8, iters = 20000;
    float *out; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 100, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)blocks * 4 /*waves*/ * iters * 8;
    double per_simd_per_s = instr / (256.0 * 4) / (ms * 1e-3);
    printf("%-18s %8.3f ms  %7.2f G wave-instr/s/SIMD  -> %.2f cycles/wave-instr @2.4GHz  | %.1f T lane-ops/s\n", name, ms,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, instr * 64 * lane_ops_per_instr / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    run<0>("v_mul_f32", 1); run<1>("v_pk_mul_f32", 2); run<2>("v_cvt_f32_ubyteN", 1); run<3>("v_add_f32", 1);
    run<4>("v_pk_add_f32", 2); run<5>("v_fma_f32", 1); run<6>("v_pk_fma_f32", 2); run<7>("v_med3_f32", 1);
    run<8>("v_cvt_u32_f32", 1); run<9>("v_cvt_pk_u8_f32", 1);
    return 0;
}
