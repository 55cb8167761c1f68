// Probe: does v_cvt_pk_u8_f32 follow MODE.round (FP32 rounding mode)?  If it truncates under round-toward-zero, the exact
// pack  T(clamp(t, 0, 255))  is ONE instruction (saturating, byte insert) instead of v_med3 + v_cvt_u32_sdwa.
// Also: issue check of v_floor_f32 + v_cvt_pk_u8_f32 as an alternative.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const float *in, unsigned *out_rne, unsigned *out_rtz, unsigned *out_floor, int n) {
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i >= n) return;
    float v = in[i];
    unsigned a = 0, b = 0, c = 0;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(a) : "v"(v));
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\ts_nop 4\n\tv_cvt_pk_u8_f32 %0, %1, 0, %0\n\ts_nop 4\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" : "+v"(b) : "v"(v));
    float f = floorf(v);
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(c) : "v"(f));
    out_rne[i] = a; out_rtz[i] = b; out_floor[i] = c;
}
int main() {
    std::vector<float> h;
    for (float v : {0.0f, 0.49f, 0.5f, 0.99f, 1.0f, 1.5f, 2.5f, 3.5f, 254.5f, 254.99f, 255.0f, 255.49f, 255.5f, 255.99f, 256.f, 300.f, 1e9f, -0.3f, -0.99f, -1.f, -1e9f, 127.99999f, 128.5f, 0.99999994f, 254.99998f})
        h.push_back(v);
    for (int k = 0; k < 4096; ++k) h.push_back(-2.0f + 260.0f * k / 4096.0f);
    int n = h.size();
    float *d; unsigned *o; hipMalloc(&d, n * 4); hipMalloc(&o, 3 * n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<(n + 63) / 64, 64>>>(d, o, o + n, o + 2 * n, n);
    std::vector<unsigned> r(3 * n); hipMemcpy(r.data(), o, 3 * n * 4, hipMemcpyDeviceToHost);
    int bad_rtz = 0, bad_floor = 0;
    for (int i = 0; i < n; ++i) {
        float t = h[i]; float c = fminf(fmaxf(t, 0.f), 255.f); unsigned e = (unsigned)c;
        if (r[n + i] != e) ++bad_rtz;
        if (r[2 * n + i] != e) ++bad_floor;
        if (i < 25) printf("v=%-12g rne=%3u rtz-mode=%3u floor+cvt=%3u trunc-clamp=%3u\n", t, r[i], r[n + i], r[2 * n + i], e);
    }
    printf("mismatches vs trunc(clamp(v,0,255)) over %d values: RTZ-mode cvt_pk_u8 %d, floor + cvt_pk_u8 %d\n", n, bad_rtz, bad_floor);
    return 0;
}
