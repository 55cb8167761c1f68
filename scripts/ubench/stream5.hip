// Does bursting the writes help the 4:1 read:write stream?  Full-row workgroups (960 threads = 15 waves x
// 16 B = one 15,360-B source row per step); output rows are either written as produced (every 2 source
// rows) or held back and written BURST rows at a time (contiguous BURST*7680 B per workgroup).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int BURST>
__global__ __launch_bounds__(960) void k(const unsigned char *src, unsigned char *dst, int rows, int rows_per_chunk,
                                         long long spitch, long long dpitch) {
    const int row_bytes = 15360;
    const int col = threadIdx.x * 16;
    const unsigned char *sb = src + blockIdx.z * spitch;
    unsigned char *db = dst + blockIdx.z * dpitch;
    auto srs = __builtin_amdgcn_make_buffer_rsrc((void *)sb, 0, row_bytes * rows, 0x00020000);
    auto drs = __builtin_amdgcn_make_buffer_rsrc((void *)db, 0, row_bytes / 2 * (rows / 2), 0x00020000);
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    u32x4 acc = {0, 0, 0, 0};
    u32x2 hold[BURST];
    for (int r = r0; r < r1; r += 2 * BURST) {
#pragma unroll
        for (int b = 0; b < BURST; ++b) {
            u32x4 a0 = __builtin_amdgcn_raw_buffer_load_b128(srs, (r + 2 * b) * row_bytes + col, 0, 0);
            u32x4 a1 = __builtin_amdgcn_raw_buffer_load_b128(srs, (r + 2 * b + 1) * row_bytes + col, 0, 0);
            acc ^= a0 ^ a1;
            hold[b] = __builtin_shufflevector(acc, acc, 0, 1);
        }
#pragma unroll
        for (int b = 0; b < BURST; ++b)
            __builtin_amdgcn_raw_buffer_store_b64(hold[b], drs, ((r / 2) + b) * (row_bytes / 2) + col / 2, 0, 0);
    }
}

int main() {
    const int n = 256, W = 15360, H = 2160;
    unsigned char *src, *dst;
    size_t sbytes = (size_t)n * W * H, dbytes = sbytes / 4;
    hipMalloc(&src, sbytes); hipMalloc(&dst, dbytes); hipMemset(src, 1, sbytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(BURST, CHUNKS)                                                                              \
    {                                                                                                   \
        int rpc = H / CHUNKS; rpc = (rpc + 2 * BURST - 1) / (2 * BURST) * (2 * BURST);                  \
        dim3 grid(1, (H + rpc - 1) / rpc, n);                                                           \
        for (int i = 0; i < 2; ++i) k<BURST><<<grid, 960>>>(src, dst, H, rpc, (long long)W * H, (long long)W * H / 4); \
        hipEventRecord(e0);                                                                             \
        for (int i = 0; i < 5; ++i) k<BURST><<<grid, 960>>>(src, dst, H, rpc, (long long)W * H, (long long)W * H / 4); \
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5; \
        printf("burst=%2d output rows (%6d B contiguous per WG) chunks=%2d : %.3f ms  %.0f GB/s (r+w)\n", BURST, BURST * 7680, (int)grid.y, ms, (sbytes + dbytes) / ms / 1e6); \
    }
    RUN(1, 8) RUN(2, 8) RUN(4, 8) RUN(8, 8) RUN(16, 8) RUN(32, 8)
    RUN(1, 20) RUN(8, 20) RUN(16, 20)
    return 0;
}
