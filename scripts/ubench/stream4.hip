// Ceiling for a 4:1 read:write stream (the byte mix of 4K->1080p rgba) with LINEAR addressing, plus
// a write-only and read-only leg, to separate "mix" effects from "pattern" effects.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// block b handles src chunk [b*16KB, +16KB) -> dst chunk [b*4KB, +4KB); grid-stride over chunks
template <int RD, int WR, int NT, int K>
__global__ __launch_bounds__(256) void lin(const u32x4 *src, u32x4 *dst, size_t nchunks) {
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        u32x4 acc = {1, 2, 3, 4};
        if (RD) {
#pragma unroll
            for (int k = 0; k < K; ++k) acc ^= (NT & 1) ? __builtin_nontemporal_load(&src[c * (256 * K) + k * 256 + threadIdx.x]) : src[c * (256 * K) + k * 256 + threadIdx.x];
        }
        if (WR) { if (NT & 2) __builtin_nontemporal_store(acc, &dst[c * 256 + threadIdx.x]); else dst[c * 256 + threadIdx.x] = acc; }
        else if (acc.x == 0x1234567) dst[0] = acc;
    }
}
int main() {
    const size_t sbytes = (size_t)256 * 3840 * 2160 * 4, dbytes = sbytes / 4;
    unsigned char *src, *dst; hipMalloc(&src, sbytes); hipMalloc(&dst, sbytes); hipMemset(src, 1, sbytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    
#define RUN(RD, WR, BLOCKS, NT, K)                                                                                          \
    {                                                                                                                \
        for (int i = 0; i < 2; ++i) lin<RD, WR, NT, K><<<BLOCKS, 256>>>((const u32x4 *)src, (u32x4 *)dst, sbytes / (4096 * K));        \
        hipEventRecord(e0);                                                                                          \
        for (int i = 0; i < 5; ++i) lin<RD, WR, NT, K><<<BLOCKS, 256>>>((const u32x4 *)src, (u32x4 *)dst, sbytes / (4096 * K));        \
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;             \
        double b = (RD ? sbytes : 0) + (WR ? sbytes / K : 0);                                                            \
        printf("linear rd=%d wr=%d blocks=%6d nt=%d reads-per-write=%d : %.3f ms  %.0f GB/s\n", RD, WR, BLOCKS, NT, K, ms, b / ms / 1e6);            \
    }
    RUN(1, 1, 65536, 0, 4) RUN(1, 1, 65536, 2, 4) RUN(1, 1, 65536, 1, 4) RUN(1, 1, 65536, 3, 4)
    RUN(1, 1, 8192, 0, 4) RUN(1, 1, 8192, 2, 4) RUN(1, 1, 8192, 3, 4)
    RUN(1, 1, 65536, 0, 1) RUN(1, 1, 65536, 3, 1) RUN(1, 1, 8192, 0, 1)
    RUN(1, 1, 65536, 0, 8) RUN(1, 1, 65536, 2, 8)
    return 0;
}
