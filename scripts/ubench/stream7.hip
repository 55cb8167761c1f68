// Can time-phased writes beat the mixed read/write ceiling?  Linear 4:1 stream.  Every workgroup buffers its
// output in LDS and flushes it only during globally agreed time windows derived from s_memrealtime (100 MHz):
// period = RD + WR microseconds, writes allowed in the last WR microseconds (or when the buffer is full).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int THREADS, int SLOTS>
__global__ __launch_bounds__(THREADS) void k(const u32x4 *src, u32x4 *dst, size_t nsteps_total, int rd_ticks, int wr_ticks, int mode,
                                             unsigned long long *stats) {
    extern __shared__ u32x4 buf[];  // [SLOTS][THREADS]
    const size_t per_block = (nsteps_total + gridDim.x - 1) / gridDim.x;
    const size_t s0 = blockIdx.x * per_block, s1 = min(nsteps_total, s0 + per_block);
    int nbuf = 0;
    size_t first = s0;  // first buffered step
    unsigned long long forced = 0;
    for (size_t st = s0; st < s1; ++st) {
        u32x4 acc = {1, 2, 3, 4};
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) acc ^= src[(st * 4 + kq) * THREADS + threadIdx.x];
        if (mode == 0) { dst[st * THREADS + threadIdx.x] = acc; continue; }
        buf[nbuf * THREADS + threadIdx.x] = acc;
        ++nbuf;
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        const bool window = (int)(t % (unsigned)(rd_ticks + wr_ticks)) >= rd_ticks;
        if (window || nbuf == SLOTS || st + 1 == s1) {
            if (!window) ++forced;
            for (int b = 0; b < nbuf; ++b) dst[(first + b) * THREADS + threadIdx.x] = buf[b * THREADS + threadIdx.x];
            first = st + 1;
            nbuf = 0;
        }
    }
    if (threadIdx.x == 0 && forced) atomicAdd(stats, forced);
}

int main() {
    const size_t sbytes = (size_t)256 * 3840 * 2160 * 4;
    unsigned char *src, *dst; hipMalloc(&src, sbytes); hipMalloc(&dst, sbytes / 4); hipMemset(src, 1, sbytes);
    unsigned long long *stats; hipMalloc(&stats, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(THREADS, SLOTS, BLOCKS, RD_US, WR_US, MODE)                                                                    \
    {                                                                                                                      \
        size_t nsteps = sbytes / (64 * THREADS);                                                                           \
        size_t lds = (size_t)SLOTS * THREADS * 16;                                                                          \
        hipFuncSetAttribute((const void *)k<THREADS, SLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
        hipMemset(stats, 0, 8);                                                                                            \
        for (int i = 0; i < 2; ++i) k<THREADS, SLOTS><<<BLOCKS, THREADS, lds>>>((const u32x4 *)src, (u32x4 *)dst, nsteps, RD_US * 100, WR_US * 100, MODE, stats); \
        hipEventRecord(e0);                                                                                                \
        for (int i = 0; i < 5; ++i) k<THREADS, SLOTS><<<BLOCKS, THREADS, lds>>>((const u32x4 *)src, (u32x4 *)dst, nsteps, RD_US * 100, WR_US * 100, MODE, stats); \
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;                   \
        unsigned long long f; hipMemcpy(&f, stats, 8, hipMemcpyDeviceToHost);                                              \
        printf("threads=%4d slots=%2d blocks=%5d rd=%3dus wr=%2dus mode=%d : %.3f ms  %.0f GB/s   forced flushes %llu\n", THREADS, SLOTS, BLOCKS, RD_US, WR_US, MODE, ms, (sbytes * 1.25) / ms / 1e6, f); \
    }
    RUN(1024, 8, 512, 20, 5, 0)
    RUN(1024, 8, 512, 20, 5, 1) RUN(1024, 8, 512, 8, 2, 1) RUN(1024, 8, 512, 40, 10, 1) RUN(1024, 8, 512, 4, 1, 1)
    RUN(1024, 8, 256, 20, 5, 1) RUN(1024, 8, 256, 8, 2, 1)
    RUN(512, 16, 512, 20, 5, 1) RUN(512, 16, 1024, 8, 2, 1) RUN(512, 16, 512, 8, 2, 0)
    return 0;
}
