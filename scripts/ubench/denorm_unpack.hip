// Microbenchmark: "denormal unpack".  The reference's unpack is u = float(s) * (1.0f / max) (src/picha.h:98-105),
// i.e. v_cvt_f32_ubyteN (half rate) + v_mul_f32.  A zero-extended byte/word IS the denormal s * 2^-149, so one
// v_mul_f32_sdwa src0_sel:BYTE_n by K = (1/max) * 2^127 gives u' = fl(s/max... ) * 2^-22 exactly (power-of-two
// scaling of the same rounded product), and weights pre-scaled by 2^22 give the reference's w*u bit for bit.
// This program checks both claims on the hardware for every u8/u16 sample and measures the issue rate.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define SDWA_MUL(SEL)                                                                                          \
    __device__ __forceinline__ float mul_##SEL(uint32_t raw, float k) {                                        \
        float r;                                                                                               \
        asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" #SEL            \
                     " src1_sel:DWORD"                                                                         \
                     : "=v"(r)                                                                                 \
                     : "v"(raw), "v"(k));                                                                      \
        return r;                                                                                              \
    }
SDWA_MUL(BYTE_0) SDWA_MUL(BYTE_1) SDWA_MUL(BYTE_2) SDWA_MUL(BYTE_3) SDWA_MUL(WORD_0) SDWA_MUL(WORD_1)

// out[6][65536]: the six selects applied to a dword holding the sample in the selected position, rest = 0xA5 noise
__global__ void check_kernel(float *out, float k8, float k16) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= 65536) return;
    const uint32_t b = s & 0xff;
    out[0 * 65536 + s] = mul_BYTE_0(0xA5A5A500u | b, k8);
    out[1 * 65536 + s] = mul_BYTE_1(0xA5A500A5u | (b << 8), k8);
    out[2 * 65536 + s] = mul_BYTE_2(0xA500A5A5u | (b << 16), k8);
    out[3 * 65536 + s] = mul_BYTE_3(0x00A5A5A5u | (b << 24), k8);
    out[4 * 65536 + s] = mul_WORD_0(0xA5A50000u | s, k16);
    out[5 * 65536 + s] = mul_WORD_1(0x0000A5A5u | (s << 16), k16);
}

// products: w' * u' against the reference's w * (float(s) * inv)
__global__ void product_kernel(const float *w, float *got, float *ref, int n, float k8, float inv8) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = (i * 2654435761u) >> 24;
    const float u = mul_BYTE_0(s, k8);
    const float wscaled = w[i] * 4194304.0f;  // 2^22, exact
    got[i] = wscaled * u;
    ref[i] = w[i] * (float(s) * inv8);
}

#define REP8(X) X X X X X X X X
template <int MODE> __global__ __launch_bounds__(256) void rate_kernel(float *out, int iters, uint32_t seed, float k) {
    uint32_t r0 = threadIdx.x * 2654435761u + seed, r1 = r0 ^ 0x1234567u;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)
            asm volatile(REP8("v_mul_f32_sdwa %0, %4, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
                              "v_mul_f32_sdwa %1, %4, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                              "v_mul_f32_sdwa %2, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n"
                              "v_mul_f32_sdwa %3, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(r0), "v"(r1), "v"(k));
        else if (MODE == 1)
            asm volatile(REP8("v_cvt_f32_ubyte0 %0, %4\n v_mul_f32 %0, %6, %0\n v_cvt_f32_ubyte1 %1, %4\n v_mul_f32 %1, %6, %1\n"
                              "v_cvt_f32_ubyte2 %2, %5\n v_mul_f32 %2, %6, %2\n v_cvt_f32_ubyte3 %3, %5\n v_mul_f32 %3, %6, %3\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(r0), "v"(r1), "v"(k));
        else if (MODE == 2)  // plain v_mul_f32 on denormal inputs (is the denormal path full rate?)
            asm volatile(REP8("v_mul_f32 %0, %4, %6\n v_mul_f32 %1, %5, %6\n v_mul_f32 %2, %4, %6\n v_mul_f32 %3, %5, %6\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(r0 & 0xffffu), "v"(r1 & 0xffu), "v"(k));
        else  // v_and + v_mul (no SDWA)
            asm volatile(REP8("v_and_b32 %0, 0xff, %4\n v_mul_f32 %0, %6, %0\n v_lshrrev_b32 %1, 24, %4\n v_mul_f32 %1, %6, %1\n"
                              "v_and_b32 %2, 0xff, %5\n v_mul_f32 %2, %6, %2\n v_lshrrev_b32 %3, 24, %5\n v_mul_f32 %3, %6, %3\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(r0), "v"(r1), "v"(k));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <int MODE> static void rate(const char *name, float *out, float k) {
    const int iters = 20000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<MODE><<<blocks, 256>>>(out, 10, 1u, k);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<MODE><<<blocks, 256>>>(out, iters, 1u, k);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    // 32 unpacked samples per loop iteration in every mode
    printf("%-44s %.2f ns per unpacked sample per wave per SIMD\n", name, ms * 1e6 / (iters * 32.0) / 8);
}

int main() {
    const float inv8 = 1.0f / 255.0f, inv16 = 1.0f / 65535.0f;
    const float k8 = ldexpf(inv8, 127), k16 = ldexpf(inv16, 127);
    float *d; hipMalloc(&d, 6 * 65536 * 4);
    check_kernel<<<256, 256>>>(d, k8, k16);
    std::vector<float> h(6 * 65536);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int sel = 0; sel < 6; ++sel)
        for (uint32_t s = 0; s < 65536; ++s) {
            const uint32_t v = sel < 4 ? (s & 0xff) : s;
            const float want = ldexpf(float(v) * (sel < 4 ? inv8 : inv16), -22);
            if (memcmp(&want, &h[sel * 65536 + s], 4) != 0) {
                if (bad < 5) printf("MISMATCH sel %d s %u: got %a want %a\n", sel, v, h[sel * 65536 + s], want);
                ++bad;
            }
        }
    printf("unpack check: %ld mismatches over 6 selects x 65536 samples (u' == fl(s * 1/max) * 2^-22)\n", bad);

    const int n = 1 << 22;
    std::vector<float> w(n);
    uint64_t x = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const int e = int((x >> 40) % 60) - 50;  // weights from 2^-50 to 2^9, both signs
        w[i] = ldexpf(float((x & 0xffffff) | 0x800000) / 8388608.0f, e) * ((x >> 63) ? -1.0f : 1.0f);
    }
    for (int i = 0; i < 64; ++i) w[i] = ldexpf(1.0f + i / 64.0f, -126 - (i % 20));  // denormal and near-denormal weights
    float *dw, *dg, *dr; hipMalloc(&dw, n * 4); hipMalloc(&dg, n * 4); hipMalloc(&dr, n * 4);
    hipMemcpy(dw, w.data(), n * 4, hipMemcpyHostToDevice);
    product_kernel<<<n / 256, 256>>>(dw, dg, dr, n, k8, inv8);
    std::vector<float> g(n), r(n);
    hipMemcpy(g.data(), dg, n * 4, hipMemcpyDeviceToHost); hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost);
    long pbad = 0, hbad = 0;
    for (int i = 0; i < n; ++i) {
        if (memcmp(&g[i], &r[i], 4) != 0) { if (pbad < 5) printf("PRODUCT MISMATCH w %a: got %a ref %a\n", w[i], g[i], r[i]); ++pbad; }
        const uint32_t s = (uint32_t(i) * 2654435761u) >> 24;
        volatile float u = float(s) * inv8; volatile float p = w[i] * u; float pp = p;
        if (memcmp(&pp, &r[i], 4) != 0) ++hbad;
    }
    printf("product check: %ld mismatches of %d (scaled-weight product vs w * (float(s) * inv)); device ref vs host: %ld\n", pbad, n, hbad);

    rate<1>("v_cvt_f32_ubyteN + v_mul_f32 (today)", d, inv8);
    rate<0>("v_mul_f32_sdwa BYTE_n on denormal bits", d, k8);
    rate<3>("v_and/v_lshr + v_mul_f32 on denormal bits", d, k8);
    rate<2>("v_mul_f32, denormal operand (4 per 'sample')", d, k8);
    return 0;
}
