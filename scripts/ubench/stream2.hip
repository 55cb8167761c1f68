// What limits the resize read pattern?  Variants of a 4:1 read:write streaming kernel over
// 256 x (3840x2160x4 B) images.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));

// V0: linear grid-stride copy-like: read 16 B/lane, every 4th read produces one 16 B store
__global__ __launch_bounds__(256) void v0(const u32x4 *src, u32x4 *dst, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        u32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i / 4 + (i % 4) * 0] = a ^ b ^ c ^ d;  // 1/4 of the bytes written (indices collide harmlessly)
    }
}

// V1/V2: column strips.  Wave = 64 lanes x 16 B = 1 KB of each row, marching down ROWS rows.
// MODE 0: one x4 load per row.  MODE 1: plus an overlapping x3 load at +16 B (the span kernel's shape).
// MODE 2: rows visited with 4 loads in flight issued back to back (unroll 4)
template <int MODE>
__global__ __launch_bounds__(256) void v1(const unsigned char *src, unsigned char *dst, int row_bytes, int rows, int rows_per_chunk,
                                          long long spitch, long long dpitch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = ((blockIdx.x * 4 + wave) * 64 + lane) * 16;
    if ((blockIdx.x * 4 + wave) * 1024 >= row_bytes) return;
    const unsigned char *sb = src + blockIdx.z * spitch;
    unsigned char *db = dst + blockIdx.z * dpitch;
    auto srs = __builtin_amdgcn_make_buffer_rsrc((void *)sb, 0, row_bytes * rows, 0x00020000);
    auto drs = __builtin_amdgcn_make_buffer_rsrc((void *)db, 0, row_bytes / 2 * (rows / 2), 0x00020000);
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    u32x4 acc = {0, 0, 0, 0};
    for (int r = r0; r < r1; r += 4) {
        u32x4 a[4]; u32x3 b[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            unsigned off = (r + d) * row_bytes + col;
            a[d] = __builtin_amdgcn_raw_buffer_load_b128(srs, off, 0, 0);
            if (MODE == 1) b[d] = __builtin_amdgcn_raw_buffer_load_b96(srs, off + 16, 0, 0);
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            acc ^= a[d];
            if (MODE == 1) { acc.x ^= b[d].x; acc.y ^= b[d].y; acc.z ^= b[d].z; }
            if (d & 1) {  // one 8 B store per lane per 2 rows
                unsigned off = ((r + d) / 2) * (row_bytes / 2) + col / 2;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_shufflevector(acc, acc, 0, 1), drs, off, 0, 0);
            }
        }
    }
}

int main() {
    const int n = 256, W = 3840 * 4, H = 2160;
    unsigned char *src, *dst;
    size_t sbytes = (size_t)n * W * H, dbytes = sbytes / 4;
    hipMalloc(&src, sbytes); hipMalloc(&dst, dbytes);
    hipMemset(src, 1, sbytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto report = [&](const char *name, float ms) {
        printf("%-40s %.3f ms  %.0f GB/s (r+w)  frac %.3f\n", name, ms, (sbytes + dbytes) / ms / 1e6, (sbytes + dbytes) / ms / 1e6 / 8000);
    };
    for (int blocks : {2048, 8192, 32768}) {
        for (int i = 0; i < 2; ++i) v0<<<blocks, 256>>>((const u32x4 *)src, (u32x4 *)dst, sbytes / 16);
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) v0<<<blocks, 256>>>((const u32x4 *)src, (u32x4 *)dst, sbytes / 16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        char nm[64]; snprintf(nm, 64, "V0 linear grid-stride blocks=%d", blocks); report(nm, ms / 5);
    }
    for (int chunks : {1, 5, 20}) {
        int rpc = (H + chunks - 1) / chunks; rpc = (rpc + 3) / 4 * 4;
        dim3 grid((W / 1024 + 3) / 4, (H + rpc - 1) / rpc, n);
        float ms; char nm[64];
#define RUN(MODE)                                                                                        \
        for (int i = 0; i < 2; ++i) v1<MODE><<<grid, 256>>>(src, dst, W, H, rpc, (long long)W * H, (long long)W * H / 4); \
        hipEventRecord(e0);                                                                              \
        for (int i = 0; i < 5; ++i) v1<MODE><<<grid, 256>>>(src, dst, W, H, rpc, (long long)W * H, (long long)W * H / 4); \
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);                    \
        snprintf(nm, 64, "V1 column strips mode=%d chunks=%d", MODE, chunks); report(nm, ms / 5);
        RUN(0) RUN(1)
    }
    return 0;
}
