// Memory-side ceiling for the resize access pattern: read 256 x 4K rgba images with the span
// kernel's load shape (per lane: P*2+3 pixels per row, rows marching down a column strip, double
// buffered groups of D rows), write a 1080p rgba image, with (almost) no arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int P, int D, int WG>
__global__ __launch_bounds__(WG) void k(const unsigned char *src, unsigned char *dst, int sw, int sh, int dw, int dh,
                                         int rows_per_chunk, long long spitch, long long dpitch) {
    constexpr int NW = 2 * (P - 1) + 5;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp0 = (blockIdx.x * (WG / 64) + wave) * 64;
    if (grp0 * P >= dw) return;
    const int x0 = (grp0 + lane) * P;
    const unsigned char *sb = src + blockIdx.z * spitch;
    unsigned char *db = dst + blockIdx.z * dpitch;
    auto srs = __builtin_amdgcn_make_buffer_rsrc((void *)sb, 0, sw * sh * 4, 0x00020000);
    auto drs = __builtin_amdgcn_make_buffer_rsrc((void *)db, 0, dw * dh * 4, 0x00020000);
    const int y0 = blockIdx.y * rows_per_chunk, y1 = min(dh, y0 + rows_per_chunk);
    const int lane_off = max(2 * x0 - 1, 0) * 4;
    const int r0 = max(2 * y0 - 1, 0), r1 = min(sh - 1, 2 * (y1 - 1) + 3);
    unsigned a[D][NW], b[D][NW];
    auto load = [&](unsigned (&t)[NW], int r) {
        unsigned off = min(r, r1) * sw * 4 + lane_off;
#pragma unroll
        for (int i = 0; i < NW; ++i) t[i] = __builtin_amdgcn_raw_buffer_load_b32(srs, off + 4 * i, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) load(a[d], r0 + d);
    unsigned acc[P] = {};
    int y = y0;
    auto group = [&](unsigned (&cur)[D][NW], unsigned (&nxt)[D][NW], int r) {
#pragma unroll
        for (int d = 0; d < D; ++d) load(nxt[d], r + D + d);
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int i = 0; i < NW; ++i) acc[i % P] ^= cur[d][i];
            const int row = r + d;
            while (y < y1 && 2 * y + 3 <= row) {
                unsigned off = y * dw * 4 + x0 * 4;
                if (x0 + P <= dw) {
                    if constexpr (P == 1) __builtin_amdgcn_raw_buffer_store_b32(acc[0], drs, off, 0, 0);
                    else if constexpr (P == 2) { u32x2 t = {acc[0], acc[1]}; __builtin_amdgcn_raw_buffer_store_b64(t, drs, off, 0, 0); }
                    else { u32x4 t = {acc[0], acc[1], acc[2], acc[3]}; __builtin_amdgcn_raw_buffer_store_b128(t, drs, off, 0, 0); }
                }
                ++y;
            }
        }
    };
    for (int r = r0;;) {
        group(a, b, r); r += D; if (r > r1) break;
        group(b, a, r); r += D; if (r > r1) break;
    }
    while (y < y1) {  // bottom edge rows
        unsigned off = y * dw * 4 + x0 * 4;
        if (x0 + P <= dw) { if constexpr (P == 1) __builtin_amdgcn_raw_buffer_store_b32(acc[0], drs, off, 0, 0); }
        ++y;
    }
}

template <int P, int D, int WG> void run(int chunks, unsigned char *src, unsigned char *dst, int n) {
    const int sw = 3840, sh = 2160, dw = 1920, dh = 1080;
    int groups = (dw + P - 1) / P;
    dim3 grid((groups + WG - 1) / WG, chunks, n);
    int rpc = (dh + chunks - 1) / chunks;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) k<P, D, WG><<<grid, WG>>>(src, dst, sw, sh, dw, dh, rpc, (long long)sw * sh * 4, (long long)dw * dh * 4);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) k<P, D, WG><<<grid, WG>>>(src, dst, sw, sh, dw, dh, rpc, (long long)sw * sh * 4, (long long)dw * dh * 4);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    double bytes = (double)n * (sw * sh * 4.0 + dw * dh * 4.0);
    printf("WG=%4d P=%d D=%d chunks=%2d : %.3f ms  %.2f us/img  %.0f GB/s (r+w)  frac %.3f\n", WG, P, D, chunks, ms, ms * 1e3 / n, bytes / ms / 1e6, bytes / ms / 1e6 / 8000);
}
int main() {
    const int n = 256;
    unsigned char *src, *dst;
    hipMalloc(&src, (size_t)n * 3840 * 2160 * 4); hipMalloc(&dst, (size_t)n * 1920 * 1080 * 4);
    hipMemset(src, 1, (size_t)n * 3840 * 2160 * 4);
    for (int chunks : {4, 16, 64}) {
        run<2, 2, 256>(chunks, src, dst, n);
        run<2, 2, 1024>(chunks, src, dst, n); run<2, 4, 1024>(chunks, src, dst, n);
        run<1, 2, 1024>(chunks, src, dst, n); run<1, 4, 1024>(chunks, src, dst, n);
        run<4, 2, 512>(chunks, src, dst, n);
    }
    return 0;
}
