// kernels_rows.hip — "row-staged" downscale kernel: every source pixel is converted to float ONCE per row, by the
// workgroup, into LDS; the lanes then read their tap windows from LDS (src/resize.cc:99-132; with the colour convert of
// src/colorconvert.cc:136-152 fused in front when the source format differs).
//
// Why: in the accumulator / span kernels a lane converts every pixel of its own window, so with long windows (large
// downscales: a 7.5x thumbnail has 30 taps at a step of 7.5 pixels) each source sample is fetched and converted by ~4 lanes —
// that redundancy, not HBM, is what bounds them (cfg3b: 0.32 of the roof).  Here the conversion — and, in the fused form, the
// whole rgb -> luma -> quantise -> unpack chain, ~20 VALU slots per pixel — happens once per source pixel:
//   stage     the 256 threads of a workgroup load the row's source span with coalesced dword loads (prefetched one row
//             ahead in registers), convert, and write floats to LDS (one 16-byte ds_write per thread and piece: no bank
//             conflicts); two LDS row buffers, ONE barrier per source row;
//   resample  a thread owns P (1 or 2) ADJACENT outputs: it reads the union of their windows from LDS (8- or 16-byte
//             reads) and applies its lane-resident weights in ascending tap order — for P = 2 the second window starts a
//             uniform S pixels after the first (zero-padded taps absorb the fractional part of the scale), so both
//             windows index one register array at compile-time offsets;
//   fold      as resize_acc_kernel: NACC output rows pending in registers, per-source-row weight records through the
//             scalar cache, a completed row is packed and stored.
// Numerics: FMA == false is the reference arithmetic, bit-exact (each pixel is unpacked with the reference's
// float(s) * (1/max); fused form: the intermediate is quantised to the destination sample type exactly as the reference's
// colorConvert writes it, then unpacked again); FMA == true: fused multiply-adds, 1/max folded into the x weights.
#define PICHA_SPAN_ONLY 1
#include "kernels_impl.inc"

namespace picha_hip {
namespace {

template <int N> using IC = std::integral_constant<int, N>;

constexpr int rows_lds_floats(int ch) { return ch == 3 ? 4 : ch; }   // rgb pixels are padded to float4 in LDS
// source pixels a thread stages per piece: whole dwords of source bytes, and (where the format allows) 4 floats = one
// 16-byte LDS write
constexpr int rows_stage_px(int src_bpp, int lds_floats) {
    return lds_floats == 1 ? 4 : (lds_floats == 2 ? ((2 * src_bpp) % 4 == 0 ? 2 : 4) : ((src_bpp % 4 == 0) ? 1 : (src_bpp % 2 == 0 ? 2 : 4)));
}

template <int N> __device__ __forceinline__ void rows_load_dwords(__amdgpu_buffer_rsrc_t rs, uint32_t off, uint32_t *w) {
    if constexpr (N == 1) w[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
    else if constexpr (N == 2) { const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0); w[0] = v.x; w[1] = v.y; }
    else if constexpr (N == 3) { const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, off, 0, 0); w[0] = v.x; w[1] = v.y; w[2] = v.z; }
    else { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
           if constexpr (N > 4) rows_load_dwords<N - 4>(rs, off + 16, w + 4); }
}

// SCH/SS16: source format; CH/S16: destination format (== the format the resample runs in); P outputs per thread, the
// second window S pixels after the first; TXB taps per output (zero padded); KST stage pieces per thread and row.
template <int SCH, bool SS16, int CH, bool S16, int P, int S, int TXB, int KST, int NACC, bool FMA>
__global__ __launch_bounds__(256) void resize_rows_kernel(RowsParams p) {
    constexpr bool CONV = (SCH != CH) || (SS16 != S16);
    constexpr int SBPP = SCH * Sample<SS16>::kBytes, BPP = CH * Sample<S16>::kBytes;
    constexpr int CHL = rows_lds_floats(CH);
    constexpr int PXT = rows_stage_px(SBPP, CHL);
    constexpr int SW = PXT * SBPP / 4;             // source dwords per stage piece
    constexpr int FW = PXT * CHL;                  // floats per stage piece
    constexpr int SPANPX = S * (P - 1) + TXB;      // pixels a thread reads per row
    static_assert((PXT * SBPP) % 4 == 0, "stage pieces are whole dwords");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    float *lds = reinterpret_cast<float *>(lds_raw);
    const int buf_floats = KST * 256 * FW;          // one row buffer

    PICHA_PROBE_BEGIN
    const int t = threadIdx.x;
    const int strip = blockIdx.x;
    const int img = blockIdx.z;
    const uint8_t *sbase = p.src_tab ? static_cast<const uint8_t *>(p.src_tab[img]) : p.src + img * p.src_pitch;
    uint8_t *dbase = p.dst_tab ? static_cast<uint8_t *>(p.dst_tab[img]) : p.dst + img * p.dst_pitch;
    // the source descriptor ends at the dword boundary at or above the last payload byte (rows are dword aligned)
    const __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(sbase), 0, (p.src_bytes + 3u) & ~3u, 0x00020000);
    const __amdgpu_buffer_rsrc_t dst_rs = __builtin_amdgcn_make_buffer_rsrc(dbase, 0, p.dst_bytes, 0x00020000);

    const int y0 = blockIdx.y * p.rows_per_chunk;
    const int y1 = min(p.dst_h, y0 + p.rows_per_chunk);
    if (y0 >= y1) return;  // workgroup-uniform

    // this strip's staged source span [strip_left, strip_left + KST*256*PXT) and this thread's outputs
    const int strip_left = p.strip_left[strip];
    const int g = min(strip * 256 + t, p.x_groups - 1);   // surplus threads duplicate the last group, never store
    const int x0 = (strip * 256 + t) * P;
    float wx[P][TXB];
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
        for (int j = 0; j < TXB; ++j) wx[q][j] = p.x_weight[(size_t)j * p.xw_stride + g * P + q];
    const int win = (p.x_left[g] - strip_left) * CHL;      // first float of this thread's window in a row buffer (host: aligned)

    const_i32_ptr yrange = (const_i32_ptr)(p.yrow_range);
    const_i32_ptr yrec = (const_i32_ptr)(p.yrow_rec);
    const int r0 = yrange[2 * y0];
    const int r1 = yrange[2 * (y1 - 1) + 1];

    float acc[NACC][P][CH];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int q = 0; q < P; ++q)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[a][q][c] = 0.0f;

    // PF rows of raw source bytes in registers: the row PF - 1 ahead is requested before the current one is staged (a
    // workgroup finishes a row in well under the HBM latency: one row ahead leaves the loads exposed)
    constexpr int PF = (KST * SW <= 8) ? 4 : 3;
    uint32_t raw[PF][KST][SW];
    auto load_row = [&](uint32_t (&dst)[KST][SW], int row) __attribute__((always_inline)) {
        const uint32_t rowoff = uint32_t(row) * uint32_t(p.src_stride);
#pragma unroll
        for (int k = 0; k < KST; ++k) {
            // pieces left of the image start (negative only through alignment) or past the row's end read the neighbouring
            // row or zeros: they only ever feed taps of weight 0
            const int px = strip_left + (k * 256 + t) * PXT;
            rows_load_dwords<SW>(src_rs, rowoff + uint32_t(max(px, 0)) * SBPP, dst[k]);
        }
    };
    // raw piece -> floats in the reference's arithmetic
    auto stage = [&](const uint32_t (&rw)[KST][SW], float *buf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < KST; ++k) {
            float f[FW];
#pragma unroll
            for (int q = 0; q < PXT; ++q) {
                float sv[SCH];
#pragma unroll
                for (int c = 0; c < SCH; ++c) {
                    const int kk = q * SCH + c;
                    const uint32_t s = SS16 ? ((rw[k][kk / 2] >> (16 * (kk % 2))) & 0xffffu) : ((rw[k][kk / 4] >> (8 * (kk % 4))) & 0xffu);
                    if constexpr (CONV) sv[c] = unpack_sample<SS16>(s);
                    else sv[c] = FMA ? float(s) : unpack_sample<SS16>(s);
                }
                if constexpr (CONV) {
                    // ColorConverter::op (src/colorconvert.cc:143-150): unpack -> channel map -> pack to the destination
                    // sample type; then the resample's own unpack of that sample (src/resize.cc:112)
                    float dv[CH];
                    channel_map<SCH, CH>(sv, dv, p.r, p.g, p.b);
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        // pack_sample then unpack_sample without leaving float: T(t) truncates a clamped, non-negative t
                        // (src/picha.h:107-114), and float(T(t)) == trunc(t) exactly
                        float tq = dv[c] * Sample<S16>::kMax;
                        tq = tq + 0.5f;
                        tq = __builtin_truncf(__builtin_amdgcn_fmed3f(tq, 0.0f, Sample<S16>::kMax));
                        f[q * CHL + c] = FMA ? tq : tq * Sample<S16>::kInv;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < CH; ++c) f[q * CHL + c] = sv[c];
                }
                if constexpr (CHL > CH) f[q * CHL + 3] = 0.0f;
            }
            float *dst = buf + (k * 256 + t) * FW;
            if constexpr (FW % 4 == 0) {
#pragma unroll
                for (int i = 0; i < FW; i += 4) *reinterpret_cast<float4 *>(dst + i) = make_float4(f[i], f[i + 1], f[i + 2], f[i + 3]);
            } else {
                static_assert(FW == 2, "8-byte stage pieces");
                *reinterpret_cast<float2 *>(dst) = make_float2(f[0], f[1]);
            }
        }
    };

    auto one_row = [&](int row, const uint32_t (&rw)[KST][SW]) __attribute__((always_inline)) {
        float *buf = lds + ((row - r0) & 1) * buf_floats;
        stage(rw, buf);
        // this row's record first: the scalar loads complete under the horizontal pass
        int rec[2 * NACC];
#pragma unroll
        for (int i = 0; i < 2 * NACC; ++i) rec[i] = yrec[(size_t)row * (2 * NACC) + i];
        // LDS writes of every wave visible; one barrier per row is enough with two buffers (a wave can only run one
        // row ahead of the slowest one).  No vmcnt wait: the prefetched row stays in flight across the barrier.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ---- horizontal pass (src/resize.cc:110-117): the union of the P windows, every pixel read once (8- or 16-byte
        // LDS reads) and applied to the outputs whose window holds it; taps ascend per output ----
        float h[P][CH];
        const float *wp = buf + win;
        auto apply = [&](int i, const float *u) __attribute__((always_inline)) {  // span pixel i (constant after unrolling)
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const int j = i - S * q;
                if (j >= 0 && j < TXB) {
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        if (j == 0) h[q][c] = wx[q][0] * u[c];
                        else if constexpr (FMA) h[q][c] = __builtin_fmaf(wx[q][j], u[c], h[q][c]);
                        else h[q][c] = h[q][c] + wx[q][j] * u[c];
                    }
                }
            }
        };
        if constexpr (CHL == 1) {
            // 8-byte reads, windows aligned to 8 bytes.  Two alternatives were measured on cfg3b and lost (profiles/r02_ab_rows_*.txt):
            // plain ds_read_b64 issued by hand with counted waits instead of the compiler's ds_read2_b64 pairs (10 % slower: the
            // hand-placed waits expose the LDS latency the compiler's interleaving hides), and 16-byte aligned windows read as
            // float4 (4 more padded taps per output: -6 % against the accumulator kernel instead of -11 %)
#pragma unroll
            for (int i = 0; i < SPANPX; i += 2) {  // an odd span reads one float past its end: inside the buffer's slack
                const float2 u = *reinterpret_cast<const float2 *>(wp + i);
                apply(i, &u.x);
                if (i + 1 < SPANPX) apply(i + 1, &u.y);
            }
        } else if constexpr (CHL == 2) {
#pragma unroll
            for (int i = 0; i < SPANPX; ++i) {
                const float2 u = *reinterpret_cast<const float2 *>(wp + 2 * i);
                const float uu[2] = {u.x, u.y};
                apply(i, uu);
            }
        } else {
#pragma unroll
            for (int i = 0; i < SPANPX; ++i) {
                const float4 u = *reinterpret_cast<const float4 *>(wp + 4 * i);
                const float uu[4] = {u.x, u.y, u.z, u.w};
                apply(i, uu);
            }
        }
        // ---- fold the row into the pending outputs (src/resize.cc:125-129), emit the rows it completes ----
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            float w;
            asm volatile("v_mov_b32 %0, %1" : "=v"(w) : "s"(rec[a]));
#pragma unroll
            for (int q = 0; q < P; ++q)
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    if constexpr (FMA) acc[a][q][c] = __builtin_fmaf(w, h[q][c], acc[a][q][c]);
                    else acc[a][q][c] = acc[a][q][c] + w * h[q][c];
                }
        }
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            const int y = rec[NACC + a];
            if (y >= 0) {  // workgroup-uniform
#ifdef PICHA_ROWS_NOSTORE   // measurement builds only: everything but the output stores (one row per chunk is stored)
                if (y == y0) {
#else
                if (y >= y0 && y < y1) {
#endif
#pragma unroll
                    for (int q = 0; q < P; ++q)
                        if (x0 + q < p.dst_w) {
                            uint32_t pw[2];
                            pack_px_words<CH, S16, FMA>(acc[a][q], pw);
                            store_px_words<CH, S16>(dst_rs, uint32_t(y * p.dst_stride + (x0 + q) * BPP), pw);
                        }
                }
#pragma unroll
                for (int q = 0; q < P; ++q)
#pragma unroll
                    for (int c = 0; c < CH; ++c) acc[a][q][c] = 0.0f;
            }
        }
    };

#pragma unroll
    for (int d = 0; d < PF - 1; ++d) load_row(raw[d], min(r0 + d, r1));
    for (int row = r0; row <= r1; row += PF) {
        auto run = [&](auto self, auto uc) __attribute__((always_inline)) -> void {
            constexpr int u = decltype(uc)::value;
            if constexpr (u < PF) {
                if (u == 0 || row + u <= r1) {  // workgroup-uniform
                    load_row(raw[(u + PF - 1) % PF], min(row + u + PF - 1, r1));  // refill the buffer staged one row ago
                    one_row(row + u, raw[u]);
                    self(self, IC<u + 1>{});
                }
            }
        };
        run(run, IC<0>{});
    }
    PICHA_PROBE_END
}

template <int SCH, bool SS16, int CH, bool S16, int P, int S, int TXB, int KST, int NACC>
hipError_t launch_rows_t(const RowsParams &p, hipStream_t s, const char **name) {
    constexpr int CHL = rows_lds_floats(CH);
    constexpr int PXT = rows_stage_px(SCH * (SS16 ? 2 : 1), CHL);
    size_t lds = (size_t)2 * KST * 256 * PXT * CHL * sizeof(float) + 32;  // + slack for the last read of a span that is not a whole number of 8 / 16-byte pieces
#ifdef PICHA_ROWS_LDS_MIN   // measurement builds: a floor under the workgroup's LDS caps the workgroups per CU (160 KB / floor)
    if (lds < PICHA_ROWS_LDS_MIN) lds = PICHA_ROWS_LDS_MIN;
#endif
    auto kf = p.fma ? resize_rows_kernel<SCH, SS16, CH, S16, P, S, TXB, KST, NACC, true> : resize_rows_kernel<SCH, SS16, CH, S16, P, S, TXB, KST, NACC, false>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    dim3 grid(p.n_strips, p.n_chunks, p.n_images);
    kf<<<grid, 256, lds, s>>>(p);
    snprintf(g_kernel_name, sizeof g_kernel_name, "resize_rows<src ch%d%s,ch%d,%s,P%d,S%d,TX%d,K%d,NACC%d,%s>", SCH, SS16 ? "u16" : "u8", CH,
             S16 ? "u16" : "u8", P, S, TXB, KST, NACC, p.fma ? "fma" : "exact");
    if (name) *name = g_kernel_name;
    return hipGetLastError();
}

template <int SCH, bool SS16, int CH, bool S16, int P, int S, int TXB, int KST>
hipError_t launch_rows_nacc(const RowsParams &p, hipStream_t s, const char **name) {
    if (p.nacc == 4) return launch_rows_t<SCH, SS16, CH, S16, P, S, TXB, KST, 4>(p, s, name);
    if (p.nacc == 6) return launch_rows_t<SCH, SS16, CH, S16, P, S, TXB, KST, 6>(p, s, name);
    return hipErrorInvalidValue;
}
template <int SCH, bool SS16, int CH, bool S16, int P, int S, int TXB>
hipError_t launch_rows_k(const RowsParams &p, hipStream_t s, const char **name) {
    switch (p.stage_pieces) {
    case 2: return launch_rows_nacc<SCH, SS16, CH, S16, P, S, TXB, 2>(p, s, name);
    case 4: return launch_rows_nacc<SCH, SS16, CH, S16, P, S, TXB, 4>(p, s, name);
    case 8: return launch_rows_nacc<SCH, SS16, CH, S16, P, S, TXB, 8>(p, s, name);
    }
    return hipErrorInvalidValue;
}

// X(P, S, TXB): one output per thread with 16 / 24 / 32 / 40 taps; two adjacent outputs whose windows are S pixels apart
#define PICHA_ROWS_SHAPES(X) X(1, 1, 16) X(1, 1, 24) X(1, 1, 32) X(1, 1, 40) \
    X(2, 3, 12) X(2, 3, 16) X(2, 4, 16) X(2, 4, 20) X(2, 5, 20) X(2, 5, 24) X(2, 6, 24) X(2, 6, 28) X(2, 7, 24) X(2, 7, 32)

template <int SCH, bool SS16, int CH, bool S16>
bool rows_shape_supported(int per_lane, int step, int taps) {
#define X(P, S, TXB) if (per_lane == P && step == S && taps == TXB) return true;
    PICHA_ROWS_SHAPES(X)
#undef X
    return false;
}
template <int SCH, bool SS16, int CH, bool S16>
hipError_t launch_rows_shape(const RowsParams &p, hipStream_t s, const char **name) {
#define X(P, S, TXB) if (p.x_per_lane == P && p.x_step == S && p.x_taps == TXB) return launch_rows_k<SCH, SS16, CH, S16, P, S, TXB>(p, s, name);
    PICHA_ROWS_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace

// Variants (source format -> resample format), one translation unit each (PICHA_ROWS_VARIANT):
//   0 grey -> grey   1 rgba -> rgba   2 rgb -> grey (fused colour convert)   3 rgba -> grey (fused)   4 greya -> greya   5 r16g16b16a16
//   r03, fused colour convert in front of the resample: 6 rgb -> greya   7 rgba -> greya   8 r16g16b16 -> r16   9 r16g16b16a16 -> r16
//   10 r16g16b16 -> grey   11 r16g16b16a16 -> grey
#ifndef PICHA_ROWS_VARIANT
#error "compile with -DPICHA_ROWS_VARIANT=0..11"
#endif
#if PICHA_ROWS_VARIANT == 0
#define ROWS_T 1, false, 1, false
#elif PICHA_ROWS_VARIANT == 1
#define ROWS_T 4, false, 4, false
#elif PICHA_ROWS_VARIANT == 2
#define ROWS_T 3, false, 1, false
#elif PICHA_ROWS_VARIANT == 3
#define ROWS_T 4, false, 1, false
#elif PICHA_ROWS_VARIANT == 4
#define ROWS_T 2, false, 2, false
#elif PICHA_ROWS_VARIANT == 5
#define ROWS_T 4, true, 4, true
#elif PICHA_ROWS_VARIANT == 6
#define ROWS_T 3, false, 2, false
#elif PICHA_ROWS_VARIANT == 7
#define ROWS_T 4, false, 2, false
#elif PICHA_ROWS_VARIANT == 8
#define ROWS_T 3, true, 1, true
#elif PICHA_ROWS_VARIANT == 9
#define ROWS_T 4, true, 1, true
#elif PICHA_ROWS_VARIANT == 10
#define ROWS_T 3, true, 1, false
#else
#define ROWS_T 4, true, 1, false
#endif
#define PICHA_CAT2(a, b) a##b
#define PICHA_CAT(a, b) PICHA_CAT2(a, b)
PICHA_PROBE_HOST(PICHA_CAT(picha_probe_rows_v, PICHA_ROWS_VARIANT))
bool PICHA_CAT(rows_supported_variant, PICHA_ROWS_VARIANT)(int per_lane, int step, int taps) { return rows_shape_supported<ROWS_T>(per_lane, step, taps); }
hipError_t PICHA_CAT(launch_rows_variant, PICHA_ROWS_VARIANT)(const RowsParams &p, hipStream_t s, const char **name) {
    return launch_rows_shape<ROWS_T>(p, s, name);
}

}  // namespace picha_hip
