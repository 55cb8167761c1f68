// What is the HBM ceiling of this box?  MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy (read + written bytes per
// second); round 1's linear copy reached 5.2-5.3 TB/s.  This sweeps what could make the difference: grid size
// (grid-stride vs one-shot), bytes in flight per thread (UNR independent 16-byte loads issued before the first store),
// workgroup size, cache-policy hints, buffer size (inside / outside the 256 MiB Infinity Cache), the relative placement of
// source and destination, and hipMemcpyDtoD.  Then the same sweep for the 4:1 read:write mix of a 2x downscale (cfg2).
//   usage: copy_ceiling [quick]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// RPW = 16-byte reads per 16-byte write (1 = copy, 4 = cfg2's mix); UNR = writes per thread per iteration, all
// RPW*UNR loads are issued before the first store.  POL: 0 plain, 1 nt stores, 2 nt loads, 3 both.
template <int RPW, int UNR, int POL, int THREADS>
__global__ __launch_bounds__(THREADS) void stream_k(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n_out) {
    const size_t per_iter = (size_t)THREADS * UNR;  // outputs per workgroup iteration
    const size_t n_iter = n_out / per_iter;          // host makes n_out a multiple
    for (size_t it = blockIdx.x; it < n_iter; it += gridDim.x) {
        u32x4 v[UNR][RPW];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const u32x4 *p = src + ((it * UNR + u) * RPW + r) * THREADS + threadIdx.x;
                v[u][r] = (POL & 2) ? __builtin_nontemporal_load(p) : *p;
            }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            u32x4 a = v[u][0];
#pragma unroll
            for (int r = 1; r < RPW; ++r) a ^= v[u][r];
            u32x4 *q = dst + (it * UNR + u) * THREADS + threadIdx.x;
            if (POL & 1) __builtin_nontemporal_store(a, q); else *q = a;
        }
    }
}

__global__ void fill_k(unsigned *p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ seed; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
        p[i] = x;
    }
}

static hipEvent_t e0, e1;
template <int RPW, int UNR, int POL, int THREADS>
double run(const void *src, void *dst, size_t out_bytes, int blocks, int reps = 6) {
    const size_t n_out = out_bytes / 16 / ((size_t)THREADS * UNR) * ((size_t)THREADS * UNR);
    const size_t n_iter = n_out / ((size_t)THREADS * UNR);
    if (blocks <= 0 || (size_t)blocks > n_iter) blocks = (int)n_iter;
    for (int i = 0; i < 2; ++i) stream_k<RPW, UNR, POL, THREADS><<<blocks, THREADS>>>((const u32x4 *)src, (u32x4 *)dst, n_out);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) stream_k<RPW, UNR, POL, THREADS><<<blocks, THREADS>>>((const u32x4 *)src, (u32x4 *)dst, n_out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double gbs = (double)n_out * 16 * (RPW + 1) / ms / 1e6;
    printf("rpw=%d unr=%d pol=%d threads=%4d blocks=%7d out=%6.0f MB : %7.3f ms  %6.0f GB/s (read+write)\n", RPW, UNR, POL, THREADS, blocks,
           n_out * 16 / 1e6, ms, gbs);
    fflush(stdout);
    return gbs;
}

int main(int argc, char **argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t SRC = (size_t)256 * 3840 * 2160 * 4;  // cfg2's source batch, 8.49 GB
    unsigned char *src, *dst;
    CK(hipMalloc(&src, SRC + (1 << 20))); CK(hipMalloc(&dst, SRC + (1 << 20)));
    fill_k<<<4096, 256>>>((unsigned *)src, SRC / 4, 12345u);
    fill_k<<<4096, 256>>>((unsigned *)dst, SRC / 4, 999u);
    CK(hipDeviceSynchronize());

    printf("== hipMemcpyDtoD, 4.25 GB (read+write bytes counted) ==\n");
    for (int rep = 0; rep < 2; ++rep) {
        const size_t b = SRC / 2;
        CK(hipMemcpyAsync(dst, src, b, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(e0));
        for (int i = 0; i < 4; ++i) CK(hipMemcpyAsync(dst, src, b, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 4;
        printf("hipMemcpyDtoD %.0f MB : %.3f ms  %.0f GB/s\n", b / 1e6, ms, 2.0 * b / ms / 1e6);
    }

    printf("== 1:1 copy, 4.25 GB out: grid size x bytes in flight ==\n");
    const size_t OUT1 = SRC / 2;
    for (int blocks : {0, 65536, 16384, 8192, 4096, 2048, 1024}) {
        run<1, 1, 0, 256>(src, dst, OUT1, blocks);
        run<1, 4, 0, 256>(src, dst, OUT1, blocks);
        if (!quick) run<1, 8, 0, 256>(src, dst, OUT1, blocks);
    }
    printf("== 1:1 copy: workgroup size, policy ==\n");
    run<1, 4, 0, 512>(src, dst, OUT1, 2048); run<1, 4, 0, 1024>(src, dst, OUT1, 1024); run<1, 2, 0, 1024>(src, dst, OUT1, 2048);
    run<1, 4, 0, 64>(src, dst, OUT1, 16384); run<1, 8, 0, 64>(src, dst, OUT1, 8192);
    run<1, 4, 1, 256>(src, dst, OUT1, 4096); run<1, 4, 2, 256>(src, dst, OUT1, 4096); run<1, 4, 3, 256>(src, dst, OUT1, 4096);
    run<1, 4, 1, 256>(src, dst, OUT1, 0); run<1, 4, 3, 256>(src, dst, OUT1, 0);
    printf("== 1:1 copy: buffer size (Infinity Cache is 256 MiB) ==\n");
    for (size_t mb : {64, 128, 256, 512, 1024, 2048}) run<1, 4, 0, 256>(src, dst, mb << 20, 4096, 20);
    printf("== 1:1 copy: destination placement relative to the source (same allocation, offset) ==\n");
    for (size_t off_kb : {0, 4, 64, 1024}) run<1, 4, 0, 256>(src, src + (SRC / 2) + off_kb * 1024, SRC / 2 - (2 << 20), 4096);

    printf("== 4:1 (cfg2's mix: 8.49 GB read, 2.12 GB written): grid size x bytes in flight ==\n");
    const size_t OUT4 = SRC / 4;
    for (int blocks : {0, 65536, 16384, 8192, 4096, 2048}) {
        run<4, 1, 0, 256>(src, dst, OUT4, blocks);
        run<4, 2, 0, 256>(src, dst, OUT4, blocks);
        if (!quick) run<4, 4, 0, 256>(src, dst, OUT4, blocks);
    }
    run<4, 1, 1, 256>(src, dst, OUT4, 0); run<4, 2, 1, 256>(src, dst, OUT4, 0); run<4, 2, 1, 256>(src, dst, OUT4, 8192);
    run<4, 2, 3, 256>(src, dst, OUT4, 0); run<4, 2, 0, 512>(src, dst, OUT4, 4096); run<4, 2, 0, 1024>(src, dst, OUT4, 2048);
    run<4, 2, 0, 64>(src, dst, OUT4, 32768);
    printf("== read-mostly (16:1) and 8:1 ==\n");
    run<16, 1, 0, 256>(src, dst, SRC / 16, 0); run<16, 1, 0, 256>(src, dst, SRC / 16, 8192); run<8, 1, 0, 256>(src, dst, SRC / 8, 0);
    run<8, 2, 0, 256>(src, dst, SRC / 8, 8192);
    return 0;
}
