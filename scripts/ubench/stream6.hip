// Does 16-byte alignment of buffer_load_dwordx4 matter for the column-strip stream?  Same kernel, the
// per-lane byte offset shifted by SHIFT bytes (0 = 16-B aligned, 4/8/12 = dword aligned only, 1 = byte aligned).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
template <int SHIFT, int ST>
__global__ __launch_bounds__(256) void v1(const unsigned char *src, unsigned char *dst, int row_bytes, int rows, int rows_per_chunk,
                                          long long spitch, long long dpitch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = ((blockIdx.x * 4 + wave) * 64 + lane) * 16 + SHIFT;
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
            b[d] = __builtin_amdgcn_raw_buffer_load_b96(srs, off + 16, 0, 0);
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            acc ^= a[d]; acc.x ^= b[d].x; acc.y ^= b[d].y; acc.z ^= b[d].z;
            if (ST && (d & 1)) {
                unsigned off = ((r + d) / 2) * (row_bytes / 2) + (col - SHIFT) / 2;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_shufflevector(acc, acc, 0, 1), drs, off, 0, 0);
            }
        }
    }
    if (!ST && acc.x == 0x12345) __builtin_amdgcn_raw_buffer_store_b32(acc.y, drs, 0, 0, 0);
}
int main() {
    const int n = 256, W = 3840 * 4, H = 2160;
    unsigned char *src, *dst;
    size_t sbytes = (size_t)n * W * H, dbytes = sbytes / 4;
    hipMalloc(&src, sbytes + 64); hipMalloc(&dst, dbytes); hipMemset(src, 1, sbytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(SHIFT, ST)                                                                                  \
    {                                                                                                   \
        int rpc = 432; dim3 grid(4, 5, n);                                                              \
        for (int i = 0; i < 2; ++i) v1<SHIFT, ST><<<grid, 256>>>(src, dst, W, H, rpc, (long long)W * H, (long long)W * H / 4); \
        hipEventRecord(e0);                                                                             \
        for (int i = 0; i < 5; ++i) v1<SHIFT, ST><<<grid, 256>>>(src, dst, W, H, rpc, (long long)W * H, (long long)W * H / 4); \
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5; \
        printf("lane offset shift %2d B, stores %d : %.3f ms\n", SHIFT, ST, ms);                        \
    }
    RUN(0, 0) RUN(4, 0) RUN(12, 0) RUN(1, 0) RUN(0, 1) RUN(4, 1) RUN(12, 1) RUN(1, 1)
    return 0;
}
