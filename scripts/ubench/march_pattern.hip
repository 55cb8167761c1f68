// march_pattern.hip (r04) — the memory side of the marching resize kernels without their arithmetic: a wave owns a column strip
// of NDW dwords per lane and row (64 x NDW x 4 contiguous bytes per wave and row), marches down a chunk of source rows with
// RIF rows in flight, XORs what it loaded and stores 16 bytes per lane every `ydiv` rows (cfg5: 4 KB per wave-row, one
// output row per four source rows).  Everything the real kernels could change about the access pattern is a parameter:
//   rif      rows in flight per wave (row buffers - 1)
//   bw       waves per workgroup (adjacent strips of the same rows)
//   chunk    source rows per chunk (+ halo rows re-read above it)
//   order    0 plain grid (strip fastest, then chunk, then image)   1 XCD-contiguous (each XCD a contiguous range of the same order)
//            2 chunk fastest: the strips of ALL chunks of one image before the next strip block
//   pf       touch-prefetch distance in rows (one dword per 64 bytes of the row `pf` rows ahead, 1 load per wave and row), 0 off
//   occ      cap on workgroups per CU through a dummy LDS allocation (0 = none)
// usage: march_pattern <n_images> <row_bytes> <rows> [ydiv [halo]]   -> a table over the parameter sets below
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Params {
    const unsigned char *src; unsigned char *dst;
    long long spitch, dpitch;
    int row_bytes, rows, out_rows, out_row_bytes;
    int strips;        // waves per row
    int chunk, halo, n_chunks, n_images, ydiv, order, pf, bw;
    int stores;   // 0 none (one store per wave at the very end), 1 non-temporal, 2 plain
    int sync;     // 1: s_barrier per row keeps a workgroup's waves on the same row
    int phase_ticks; // > 0: time-phased stores — a wave keeps up to 8 finished 16-byte outputs per lane in registers and stores them when
                     // s_memrealtime / phase_ticks changes (100 MHz ticks, the same on the whole chip: every wave flushes in the same few us)
    int lpol, spol;  // cache policy (buffer aux bits: 1 sc0, 2 nt, 16 sc1) of the row loads / of the stores when stores == 3
};
template <int AUX> __device__ __forceinline__ unsigned ld(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, AUX); }

template <int NDW, int RIF, int LPOL = 0, int SPOL = -1>
__global__ __launch_bounds__(1024) void march(Params p) {
    extern __shared__ unsigned dummy_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nsb = (p.strips + p.bw - 1) / p.bw;
    unsigned lin = blockIdx.x;
    const unsigned total = nsb * p.n_chunks * p.n_images;
    if (p.order == 1) { const unsigned full = total & ~7u; if (lin < full) lin = (lin & 7u) * (full >> 3) + (lin >> 3); }
    unsigned bx, by, bz;
    if (p.order == 2) { by = lin % p.n_chunks; unsigned t = lin / p.n_chunks; bx = t % nsb; bz = t / nsb; }
    else { bx = lin % nsb; unsigned t = lin / nsb; by = t % p.n_chunks; bz = t / p.n_chunks; }
    const int strip = bx * p.bw + wave;
    if (strip >= p.strips && !p.sync) return;
    const bool live = strip < p.strips;
    const unsigned char *sb = p.src + bz * p.spitch;
    unsigned char *db = p.dst + bz * p.dpitch;
    auto srs = __builtin_amdgcn_make_buffer_rsrc((void *)sb, 0, (unsigned)(p.row_bytes * (long long)p.rows), 0x00020000);
    auto drs = __builtin_amdgcn_make_buffer_rsrc((void *)db, 0, (unsigned)(p.out_row_bytes * (long long)p.out_rows), 0x00020000);
    const int r0 = max(0, (int)by * p.chunk - p.halo), r1 = min(p.rows, ((int)by + 1) * p.chunk) - 1;
    const unsigned lane_off = (strip * 64 + lane) * NDW * 4;
    const unsigned pf_off = strip * 64 * NDW * 4 + lane * 64;   // one dword of every 64 bytes of the wave's run (NDW = 16: 4 KB = 64 lanes)
    unsigned raw[RIF + 1][NDW];
    auto load = [&](unsigned (&t)[NDW], int r) __attribute__((always_inline)) {
        const unsigned off = live ? unsigned(min(r, r1)) * p.row_bytes + lane_off : 0xfffffff0u;
#pragma unroll
        for (int i = 0; i < NDW; ++i) t[i] = ld<LPOL>(srs, off + 4 * i);
    };
#pragma unroll
    for (int k = 0; k < RIF; ++k) load(raw[k], r0 + k);
    unsigned acc[4] = {0, 0, 0, 0}, pfacc = 0;
    u32x4 obuf[8];
    int nbuf = 0, first_out = 0;
    unsigned phase = p.phase_ticks > 0 ? unsigned(__builtin_amdgcn_s_memrealtime() / (unsigned long long)p.phase_ticks) : 0u;
    for (int r = r0; r <= r1; r += RIF + 1) {
#pragma unroll
        for (int k = 0; k < RIF + 1; ++k) {
            load(raw[(k + RIF) % (RIF + 1)], r + k + RIF);
            if (p.pf) {
                unsigned v = 0;
                for (int j = 0; j < NDW / 16; ++j)
                    v ^= __builtin_amdgcn_raw_buffer_load_b32(srs, unsigned(min(r + k + p.pf, r1)) * p.row_bytes + pf_off + j * 4096, 0, 0);
                pfacc ^= v;
            }
            const int row = r + k;
            if (p.sync) __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < NDW; ++i) acc[i & 3] ^= raw[k][i];
            if (p.phase_ticks > 0) {
                if (live && row <= r1 && row >= (int)by * p.chunk && (row % p.ydiv) == p.ydiv - 1) {
                    const u32x4 t = {acc[0], acc[1], acc[2], acc[3] ^ pfacc};
                    switch (nbuf) {   // wave-uniform
                    case 0: obuf[0] = t; break; case 1: obuf[1] = t; break; case 2: obuf[2] = t; break; case 3: obuf[3] = t; break;
                    case 4: obuf[4] = t; break; case 5: obuf[5] = t; break; case 6: obuf[6] = t; break; default: obuf[7] = t; break;
                    }
                    if (nbuf == 0) first_out = row / p.ydiv;
                    ++nbuf;
                }
                const unsigned ph = unsigned(__builtin_amdgcn_s_memrealtime() / (unsigned long long)p.phase_ticks);
                if (nbuf == 8 || (nbuf > 0 && (ph != phase || row >= r1))) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (q < nbuf) __builtin_amdgcn_raw_buffer_store_b128(obuf[q], drs, unsigned(first_out + q) * p.out_row_bytes + (strip * 64 + lane) * 16, 0, 2);
                    nbuf = 0;
                }
                phase = ph;
            } else
            if (live && p.stores && row <= r1 && row >= (int)by * p.chunk && (row % p.ydiv) == p.ydiv - 1) {
                u32x4 t = {acc[0], acc[1], acc[2], acc[3] ^ pfacc};
                const unsigned doff = unsigned(row / p.ydiv) * p.out_row_bytes + (strip * 64 + lane) * 16;
                if (p.stores == 1) __builtin_amdgcn_raw_buffer_store_b128(t, drs, doff, 0, 2);
                else if (p.stores == 2) __builtin_amdgcn_raw_buffer_store_b128(t, drs, doff, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(t, drs, doff, 0, SPOL < 0 ? 0 : SPOL);
            }
        }
    }
    if (live && !p.stores && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3] ^ pfacc) == 0x12345u)   // keeps the loads alive, (almost) never stores
        __builtin_amdgcn_raw_buffer_store_b32(acc[0], drs, (strip * 64 + lane) * 16, 0, 0);
}

template <int NDW, int RIF, int LPOL = 0, int SPOL = -1> float run(const Params &p, int occ, int reps) {
    const unsigned nsb = (p.strips + p.bw - 1) / p.bw;
    const unsigned grid = nsb * p.n_chunks * p.n_images;
    const size_t lds = occ > 0 ? (size_t)(160 * 1024 / occ) / 256 * 256 : 0;
    if (lds > 65536) CK(hipFuncSetAttribute((const void *)march<NDW, RIF, LPOL, SPOL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) march<NDW, RIF, LPOL, SPOL><<<grid, 64 * p.bw, lds>>>(p);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) march<NDW, RIF, LPOL, SPOL><<<grid, 64 * p.bw, lds>>>(p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

float dispatch(int ndw, int rif, const Params &p, int occ, int reps) {
    if (p.lpol != 0 || p.stores == 3) {   // cache-policy variants: 4 KB wave-rows, one row in flight
#define PQ(L, S) if (ndw == 16 && rif == 1 && p.lpol == L && (p.stores == 3 ? p.spol : -1) == S) return run<16, 1, L, S>(p, occ, reps);
        PQ(0, 1) PQ(0, 3) PQ(0, 16) PQ(0, 17) PQ(0, 18) PQ(0, 19) PQ(2, -1) PQ(2, 3) PQ(2, 16) PQ(2, 18) PQ(16, -1) PQ(16, 16) PQ(16, 18) PQ(1, -1) PQ(18, -1) PQ(18, 18)
#undef PQ
        return -1.0f;
    }
#define C(N, R) if (ndw == N && rif == R) return run<N, R>(p, occ, reps);
    C(16, 1) C(16, 2) C(16, 3) C(16, 4) C(32, 1) C(32, 2) C(8, 1) C(8, 2) C(8, 4) C(4, 2) C(4, 4) C(4, 8)
#undef C
    return -1.0f;
}

int main(int argc, char **argv) {
    // usage: march_pattern <n_images> <row_bytes> <rows> <ydiv> <halo> <pitch_pad> <variant>...
    //   variant = ndw,rif,bw,chunk,order,pf,occ,stores,sync[,lpol,spol,place,gap,phase_us]
    //   place: 0 source and destination in two allocations, 1 one allocation [all sources | gap | all destinations],
    //          2 one allocation, per image [source i | gap | destination i]; 3 one allocation [all destinations | gap | all sources]; gap in KB
    if (argc < 8) { printf("usage: march_pattern n row_bytes rows ydiv halo pitch_pad variant...\n"); return 1; }
    // pitch_pad may be "<pad>" or "<pad>:<plan>" with plan = the order of allocations of a fresh process, e.g. S,P16,D = source,
    // a 16 GB spacer, destination (place 0 then uses these; D is sized for 16-dword strips)
    const int n = atoi(argv[1]), row_bytes = atoi(argv[2]), rows = atoi(argv[3]), ydiv = atoi(argv[4]), halo = atoi(argv[5]), pad = atoi(argv[6]);
    const char *plan = strchr(argv[6], ':') ? strchr(argv[6], ':') + 1 : nullptr;
    const int pitch = row_bytes + pad;
    const size_t sbytes = (size_t)pitch * rows;
    unsigned char *src = nullptr, *dst, *big = nullptr, *plan_dst = nullptr;
    size_t plan_dst_bytes = 0;
    if (plan) {
        const size_t dmax = (size_t)(rows / ydiv + 1) * row_bytes * n;   // enough for 4-dword strips (one 16-byte store per 16 source bytes and ydiv rows)
        for (const char *q = plan; *q;) {
            if (*q == 'S') { CK(hipMalloc(&src, sbytes * n)); ++q; }
            else if (*q == 'D') { CK(hipMalloc(&plan_dst, dmax)); plan_dst_bytes = dmax; ++q; }
            else if (*q == 'P') { void *sp; CK(hipMalloc(&sp, (size_t)atof(q + 1) * ((size_t)1 << 30) / 1)); while (*q && *q != ',') ++q; printf("spacer %p\n", sp); }
            if (*q == ',') ++q;
        }
        printf("plan %s: src %p dst %p (dst - src = %.3f GB)\n", plan, (void *)src, (void *)plan_dst, ((double)(plan_dst - src)) / (1u << 30));
    }
    if (!src) CK(hipMalloc(&src, sbytes * n));
    const size_t big_bytes = (sbytes + (size_t)rows / ydiv * row_bytes) * n + ((size_t)34 << 30);   // room for gaps of up to 16 GB on either side
    if (!plan) CK(hipMalloc(&big, big_bytes));
    std::vector<unsigned> h(sbytes / 4);
    {   // pseudo-random pixels (the clock a chip holds depends on the data)
        unsigned s = 12345u;
        for (auto &v : h) { s = s * 1664525u + 1013904223u; v = s; }
        for (int i = 0; i < n; ++i) CK(hipMemcpy(src + i * sbytes, h.data(), sbytes, hipMemcpyHostToDevice));
    }
    printf("march_pattern: %d images x %d rows x %d bytes, row pitch %d (%.2f GB), one 16-byte store per lane every %d rows, halo %d rows\n", n, rows, row_bytes, pitch, (double)row_bytes * rows * n / 1e9, ydiv, halo);
    for (int rnd = 0; rnd < 2; ++rnd)
    for (int a = 7; a < argc; ++a) {
        int ndw, rif, bw, chunk, order, pf, occ, stores, sync, lpol = 0, spol = 0, place = 0, gap = 0, phase_us = 0;
        if (sscanf(argv[a], "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d", &ndw, &rif, &bw, &chunk, &order, &pf, &occ, &stores, &sync, &lpol, &spol, &place, &gap, &phase_us) < 9) { printf("bad variant %s\n", argv[a]); continue; }
        Params p{};
        p.strips = row_bytes / (64 * ndw * 4);
        if (p.strips * 64 * ndw * 4 != row_bytes) { printf("variant %s: strips do not tile the row\n", argv[a]); continue; }
        p.out_rows = rows / ydiv; p.out_row_bytes = p.strips * 64 * 16;
        const size_t dbytes = (size_t)p.out_rows * p.out_row_bytes;
        if (plan_dst && dbytes * n > plan_dst_bytes) { printf("variant %s: destination larger than the planned allocation\n", argv[a]); continue; }
        if (plan_dst) dst = plan_dst; else CK(hipMalloc(&dst, dbytes * n));
        p.src = src; p.dst = dst; p.spitch = (long long)sbytes; p.dpitch = (long long)dbytes;
        p.lpol = lpol; p.spol = spol; p.phase_ticks = phase_us * 100;
        if (place && !big) { printf("variant %s needs the big allocation (no plan)\n", argv[a]); continue; }
        if (place == 1) {
            p.src = big; p.dst = big + sbytes * n + (size_t)gap * 1024;
            for (int i = 0; i < n; ++i) CK(hipMemcpy(big + i * sbytes, h.data(), sbytes, hipMemcpyHostToDevice));
        } else if (place == 3) {   // destination BEFORE the source: [all destinations | gap | all sources]
            unsigned char *sb2 = big + dbytes * n + (size_t)gap * 1024;
            p.dst = big; p.src = sb2;
            for (int i = 0; i < n; ++i) CK(hipMemcpy(sb2 + i * sbytes, h.data(), sbytes, hipMemcpyHostToDevice));
        } else if (place == 2) {
            const size_t per = sbytes + (size_t)gap * 1024 + dbytes;
            p.src = big; p.dst = big + sbytes + (size_t)gap * 1024; p.spitch = p.dpitch = (long long)per;
            for (int i = 0; i < n; ++i) CK(hipMemcpy(big + i * per, h.data(), sbytes, hipMemcpyHostToDevice));
        }
        p.row_bytes = pitch; p.rows = rows; p.chunk = chunk; p.halo = halo; p.n_chunks = (rows + chunk - 1) / chunk;
        p.n_images = n; p.ydiv = ydiv; p.order = order; p.pf = pf; p.bw = bw; p.stores = stores; p.sync = sync;
        const float ms = dispatch(ndw, rif, p, occ, 10);
        const double rd = (double)row_bytes * rows * n * (1.0 + (double)halo / chunk), wr = stores ? (double)dbytes * n : 0.0;
        printf("rnd %d  ndw %2d rif %d bw %2d chunk %4d order %d pf %d occ %d stores %d sync %d lpol %2d spol %2d place %d gap %5d KB phase %3d us : %7.3f ms  read %5.2f TB/s (with halo rows)  read+write %5.2f TB/s\n",
               rnd, ndw, rif, bw, chunk, order, pf, occ, stores, sync, lpol, spol, place, gap, phase_us, ms, rd / ms / 1e9, (rd + wr) / ms / 1e9);
        if (!plan_dst) CK(hipFree(dst));
    }
    return 0;
}
