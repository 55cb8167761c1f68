// Column-strip read pattern: which workgroup shape / block order gets closest to linear streaming?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));

// ORDER 0: blockIdx.x = column block (fastest), y = chunk, z = image   (current kernel)
// ORDER 1: linear id; XCD-aware: the nx column blocks of one (image, chunk) band share id % 8
// ORDER 2: linear id; column block slowest (consecutive ids = consecutive bands)
template <int WG, int ORDER, int UNR, int ST>
__global__ __launch_bounds__(WG) void v1(const unsigned char *src, unsigned char *dst, int row_bytes, int rows, int rows_per_chunk,
                                         long long spitch, long long dpitch, int nx, int nchunks, int nimg) {
    int bx, by, bz;
    if (ORDER == 0) { bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z; }
    else {
        int id = blockIdx.x;
        if (ORDER == 1) {
            int xcd = id & 7, k = id >> 3;      // k-th block on this XCD
            int band = (k / nx) * 8 + xcd;      // bands dealt round-robin to XCDs, nx consecutive k share a band
            bx = k % nx; by = band % nchunks; bz = band / nchunks;
            if (bz >= nimg) return;
        } else { int band = id % (nchunks * nimg); bx = id / (nchunks * nimg); by = band % nchunks; bz = band / nchunks; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WPB = WG / 64;
    const int col = ((bx * WPB + wave) * 64 + lane) * 16;
    if ((bx * WPB + wave) * 1024 >= row_bytes) return;
    const unsigned char *sb = src + bz * spitch;
    unsigned char *db = dst + bz * dpitch;
    auto srs = __builtin_amdgcn_make_buffer_rsrc((void *)sb, 0, row_bytes * rows, 0x00020000);
    auto drs = __builtin_amdgcn_make_buffer_rsrc((void *)db, 0, row_bytes / 2 * (rows / 2), 0x00020000);
    const int r0 = by * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    u32x4 acc = {0, 0, 0, 0};
    for (int r = r0; r < r1; r += UNR) {
        u32x4 a[UNR]; u32x3 b[UNR];
#pragma unroll
        for (int d = 0; d < UNR; ++d) {
            unsigned off = (r + d) * row_bytes + col;
            a[d] = __builtin_amdgcn_raw_buffer_load_b128(srs, off, 0, 0);
            b[d] = __builtin_amdgcn_raw_buffer_load_b96(srs, off + 16, 0, 0);
        }
#pragma unroll
        for (int d = 0; d < UNR; ++d) {
            acc ^= a[d]; acc.x ^= b[d].x; acc.y ^= b[d].y; acc.z ^= b[d].z;
            if (ST == 1 && (d & 1)) {   // 8 B per lane every 2 rows
                unsigned off = ((r + d) / 2) * (row_bytes / 2) + col / 2;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_shufflevector(acc, acc, 0, 1), drs, off, 0, 0);
            }
            if (ST == 2 && (d & 3) == 3) {   // 16 B per lane every 4 rows (same bytes)
                unsigned off = ((r + d) / 4) * row_bytes + col;
                __builtin_amdgcn_raw_buffer_store_b128(acc, drs, off, 0, 0);
            }
            if (ST == 3 && (d & 1)) {   // 8 B per lane, non-temporal
                unsigned off = ((r + d) / 2) * (row_bytes / 2) + col / 2;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_shufflevector(acc, acc, 0, 1), drs, off, 0, 2);
            }
        }
    }
    if (ST == 0 && acc.x == 0x12345) __builtin_amdgcn_raw_buffer_store_b32(acc.y, drs, 0, 0, 0);
}

int main() {
    const int n = 256, W = 3840 * 4, H = 2160;
    unsigned char *src, *dst;
    size_t sbytes = (size_t)n * W * H, dbytes = sbytes / 4;
    hipMalloc(&src, sbytes); hipMalloc(&dst, dbytes);
    hipMemset(src, 1, sbytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto report = [&](const char *name, float ms) {
        printf("%-56s %.3f ms  %.0f GB/s (r+w)  frac %.3f\n", name, ms, (sbytes + dbytes) / ms / 1e6, (sbytes + dbytes) / ms / 1e6 / 8000);
    };
#define RUN(WG, ORDER, UNR, CHUNKS, ST)                                                                          \
    {                                                                                                        \
        int chunks = CHUNKS; int rpc = (H + chunks - 1) / chunks; rpc = (rpc + UNR - 1) / UNR * UNR; chunks = (H + rpc - 1) / rpc; \
        int nx = (W / 1024 + WG / 64 - 1) / (WG / 64);                                                        \
        dim3 grid = ORDER == 0 ? dim3(nx, chunks, n) : dim3(ORDER == 1 ? ((chunks * n + 7) / 8) * 8 * nx : nx * chunks * n); \
        for (int i = 0; i < 2; ++i) v1<WG, ORDER, UNR, ST><<<grid, WG>>>(src, dst, W, H, rpc, (long long)W * H, (long long)W * H / 4, nx, chunks, n); \
        hipEventRecord(e0);                                                                                  \
        for (int i = 0; i < 5; ++i) v1<WG, ORDER, UNR, ST><<<grid, WG>>>(src, dst, W, H, rpc, (long long)W * H, (long long)W * H / 4, nx, chunks, n); \
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);             \
        char nm[80]; snprintf(nm, 80, "WG=%d order=%d unroll=%d chunks=%d store-mode=%d", WG, ORDER, UNR, chunks, ST); report(nm, ms / 5); \
    }
    RUN(256, 0, 4, 5, 1) RUN(256, 0, 4, 5, 0) RUN(256, 0, 4, 5, 2) RUN(256, 0, 4, 5, 3)
    RUN(256, 0, 8, 5, 0) RUN(256, 0, 8, 5, 2)
    RUN(1024, 0, 4, 20, 0) RUN(1024, 0, 4, 20, 2)
    return 0;
}
