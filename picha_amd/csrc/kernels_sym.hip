// kernels_sym.hip — streaming resize kernel for integer scale factors whose tap tables are UNIFORM AND SYMMETRIC
// (every interior output has the same weights w[0..T), and w[j] == w[T-1-j] bit for bit: true for all six picha
// filters whenever the scale is an integer, src/resize.cc:19-50 — the tap centre then lands on an integer or
// half-integer position and the filters are even functions of |offset|).
//
// What that buys over resize_stream_kernel (same decomposition: a wave owns 64*P adjacent output columns of a row chunk
// and marches down the source rows; no LDS, no barrier; vertical accumulators in registers; src/resize.cc:99-132):
//   * the weights are wave-uniform: (T+1)/2 VGPRs per axis instead of P*TX lane-resident ones, which is what makes three
//     or four outputs per lane affordable (fewer pixel conversions per output: a lane converts S*(P-1)+TX pixels for P
//     outputs);
//   * products are SHARED between outputs without moving a rounding: source pixel q is tap j of output o and tap j' of
//     output o' with w[j] == w[j'] bitwise, so fl(w[j] * u) is computed once and added into both sums — each sum still
//     adds the reference's products in the reference's (ascending) order.  The same holds vertically: resampled row r
//     is tap j of output row m and tap j' of row m - i.  The sharing is left to common-subexpression elimination: the
//     code below just names the weight by its symmetric index min(j, T-1-j).
// Edge handling: waves that contain an output whose window hangs over the left/right image edge (different, renormalised
// weights) run the general per-lane-weight horizontal pass; steps that touch an output row at the top/bottom edge
// run the general vertical fold with that step's own weights.  Both are wave-uniform decisions.
#define PICHA_SPAN_ONLY 1
#include "kernels_impl.inc"

namespace picha_hip {
namespace {

template <int N> using IC = std::integral_constant<int, N>;
// Measurement builds only (scripts/build_variant.sh; results differ from the reference — never shipped; profiles/r05_ab_cfg2_ablate.txt):
//   -DPICHA_SYM_ABLATE=1  no destination stores (and no pack)     =2  no source loads inside the march (it computes on the rows of the prologue)
//   -DPICHA_SYM_ABLATE=3  loads and stores only: the arithmetic of a row is one XOR over its registers
//   any non-zero value also reads PICHA_SYM_LDS (environment): bytes of unused dynamic LDS per workgroup, which holds a build at a chosen number of waves per SIMD
#ifndef PICHA_SYM_ABLATE
#define PICHA_SYM_ABLATE 0
#endif
constexpr int ce_min(int a, int b) { return a < b ? a : b; }

// NBUF row buffers in registers (NBUF - 1 rows in flight per wave): 3 unless a row is more than 12 dwords per lane
template <int CH, bool S16, int P, int SX, int TX, bool XCH> constexpr int sym_nbuf() {
    return (XCH ? XchCfg<CH, S16, P, SX, TX>::NOWN + XchCfg<CH, S16, P, SX, TX>::NE : SpanCfg<CH, S16, P, SX, TX>::NW) <= 12 ? 3 : 2;
}

// EPI (r04): the colour convert of a resize -> colorConvert chain as a STORE EPILOGUE (as on the accumulator kernel since r03,
// epilogue_store in kernels_impl.inc): the finished pixel is packed to the resize's own sample type exactly as src/resize.cc:130
// writes it, unpacked, channel-mapped and packed to p.epi_pixel (src/colorconvert.cc:143-150); dst_* describe THAT image.
template <int CH, bool S16, int P, int SX, int TX, int SY, int TY, bool FMA, bool XCH = false, bool EPI = false>
#ifdef PICHA_SYM_WAVES   // measurement builds: the register budget of N waves per SIMD
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PICHA_SYM_WAVES, PICHA_SYM_WAVES))) void resize_sym_kernel(StreamParams p) {
#else
__global__ __launch_bounds__(256) void resize_sym_kernel(StreamParams p) {
#endif
    using C = SpanCfg<CH, S16, P, SX, TX>;
    using X = XchCfg<CH, S16, P, SX, TX>;
    static_assert(!XCH || X::OK, "neighbour exchange: dword pixels, halo within one lane's pixels");
    constexpr int NRAW = XCH ? X::NRAW : C::NW;  // registers per row buffer
    constexpr int NBUF = sym_nbuf<CH, S16, P, SX, TX, XCH>();
    constexpr int NPEND = (TY - 1) / SY + 1;
    constexpr int ICLOSE = (TY - 1) / SY, PHCLOSE = (TY - 1) % SY;
    constexpr int YW = (NPEND * SY + 1 + 3) / 4 * 4;  // dwords per step record: NPEND*SY weights, flags in the last dword
    constexpr int TXH = (TX + 1) / 2, TYH = (TY + 1) / 2;
    PICHA_PROBE_BEGIN

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto run_item = [&](int bx, int y0, int y1, int img) __attribute__((always_inline)) {
    const int grp0 = (bx * (blockDim.x >> 6) + wave) * kWave;  // 1..4 waves per workgroup (host's choice)
    if (grp0 * P >= p.dst_w) return;  // wave-uniform; the march has no barriers
    const int x0 = (grp0 + lane) * P;

    const uint8_t *sbase = uniform_ptr(p.src_tab ? static_cast<const uint8_t *>(p.src_tab[img]) : p.src + img * p.src_pitch);
    uint8_t *dbase = uniform_ptr(p.dst_tab ? static_cast<uint8_t *>(p.dst_tab[img]) : p.dst + img * p.dst_pitch);
    const int sadj = C::BROT ? int(reinterpret_cast<uintptr_t>(sbase) & 3u) : 0;
    const __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(sbase) - sadj, 0, C::BROT ? ((p.src_bytes + sadj + 3u) & ~3u) : p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dst_rs = __builtin_amdgcn_make_buffer_rsrc(dbase, 0, p.dst_bytes, 0x00020000);

    if (y0 >= y1) return;
    const int m_end = y1 - 1 + ICLOSE;  // last step that completes an output row of this chunk
    const_i32_ptr ystep = (const_i32_ptr)(p.y_step_weight);

    // every output of this wave has the canonical weights and window start SX*x + x_base?
    const bool interior = grp0 * P >= p.cx0 && (grp0 + kWave) * P <= p.cx1;

    auto march = [&](auto xc_c, auto nb_c) __attribute__((always_inline)) {
        constexpr bool XC = decltype(xc_c)::value;
        constexpr int NB = decltype(nb_c)::value;               // row buffers in registers: NB - 1 rows are in flight per wave
        constexpr int BPER = NB / ce_gcd(NB, SY);                   // steps after which the row-buffer roles repeat
        constexpr int USTEPS = NPEND * BPER / ce_gcd(NPEND, BPER);  // ... and the accumulator roles too: unrolled together
        // horizontal weights: canonical (wave-uniform, indexed by symmetric id) or lane-resident (edge waves)
        float wx[XC ? 1 : P][XC ? TXH : TX];
        int32_t lane_off;
        bool neg_lane = false;
        if constexpr (XC) {
#pragma unroll
            for (int id = 0; id < TXH; ++id) asm volatile("v_mov_b32 %0, %1" : "=v"(wx[0][id]) : "s"(p.x_canon[id]));
            lane_off = (SX * x0 + p.x_base) * C::BPP + sadj;
        } else {
            const int grp = min(grp0 + lane, p.x_groups - 1);
#pragma unroll
            for (int q = 0; q < P; ++q)
#pragma unroll
                for (int j = 0; j < TX; ++j) wx[q][j] = p.x_weight[(size_t)j * p.xw_stride + grp * P + q];
            lane_off = p.x_left[grp] * C::BPP + sadj;
            neg_lane = __any(lane_off < 0);
        }
        // XCH: the lane's own pixels start at SX * x0 (the table is uniform-base: every window starts HL pixels before them)
        if constexpr (XCH) { lane_off = SX * x0 * C::BPP; neg_lane = false; }
        const bool edge_lane = lane == 0 || lane == kWave - 1;
        const int32_t edge_off = lane == 0 ? -X::M * C::BPP : X::OWN * C::BPP;
        float wyc[TYH];
#pragma unroll
        for (int id = 0; id < TYH; ++id) asm volatile("v_mov_b32 %0, %1" : "=v"(wyc[id]) : "s"(p.y_canon[id]));

#if PICHA_SYM_ABLATE == 3
        uint32_t xacc = 0;
#endif
        float acc[NPEND][P][CH];  // static roles: at step index K of the unrolled group, output m - i lives in slot (i - K) mod NPEND
        uint32_t raw[NB][NRAW];
        if constexpr (XCH) {  // halo registers of the lanes that never load them: defined once, kept by the masked loads
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int k = X::NOWN; k < NRAW; ++k) raw[b][k] = 0u;
        }

        auto load_row = [&](uint32_t (&t)[NRAW], int row) __attribute__((always_inline)) {
            const int rr = (row >= p.src_h) ? p.src_h : row;  // rows outside the image read 0 (outside the descriptor)
            const int32_t off = rr * p.src_stride + lane_off;
            if constexpr (XCH) {
                xch_load_row<X>(src_rs, off, edge_lane, edge_off, t);
            } else if constexpr (XC) {
                load_span<C, false>(src_rs, off, t);
            } else {
                // a lane whose span starts before the image (left edge): in row 0 a wrapped offset would void the whole
                // merged load, so that row is fetched dword by dword with clamped offsets (zero-weight taps only)
                if (neg_lane && rr <= 0) load_span<C, true>(src_rs, off, t);
                else load_span<C, false>(src_rs, off, t);
            }
        };

        auto one_step = [&](int m, auto k_c) __attribute__((always_inline)) {
            constexpr int K = decltype(k_c)::value;
            int wb[YW];
#pragma unroll
            for (int t = 0; t < YW; ++t) wb[t] = ystep[(size_t)m * YW + t];
            const bool ycanon = (wb[YW - 1] & 2) != 0;  // all NPEND outputs of this step exist and are canonical
            // optional: keep the waves of a workgroup (adjacent column strips) on the same step, so the partial lines at
            // their common edge are fetched once and hit in the CU's L1 (waves that left the kernel do not take part)
            if (p.sync_steps) __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int ph = 0; ph < SY; ++ph) {
                const int T = K * SY + ph;                 // row index inside the unrolled group (a constant after unrolling)
                const int bi = T % NB;                     // buffer holding this row
                const int bn = (T + NB - 1) % NB;          // buffer consumed by the previous row: refill it, NB-1 rows ahead
                const int row = SY * m + p.y_base + ph;
                if (PICHA_SYM_ABLATE != 2 || p.src_h < 0) load_row(raw[bn], row + NB - 1);
#if PICHA_SYM_ABLATE == 3
                {
                    // xacc runs over ALL rows: a per-row value would be dead on the rows that complete no output, and the compiler then deletes their loads
#pragma unroll
                    for (int k = 0; k < NRAW; ++k) xacc ^= raw[bi][k];
                    if (ph == PHCLOSE && m - ICLOSE >= y0 && x0 + P <= p.dst_w) {
                        uint32_t dw[P * C::BPP / 4];
#pragma unroll
                        for (int k = 0; k < P * C::BPP / 4; ++k) dw[k] = xacc + k;
                        store_dwords<P * C::BPP / 4, kStoreNT>(dst_rs, uint32_t((m - ICLOSE) * p.dst_stride + x0 * C::BPP), dw);
                    }
                    continue;
                }
#endif
                // ---- horizontal pass (src/resize.cc:110-117) ----
                float h[P][CH];
                uint32_t win[XCH ? X::NWIN : (C::BROT ? C::NWIN : 1)];
                if constexpr (XCH) xch_window<X>(raw[bi], win);
                if constexpr (C::BROT)
                    rotate_span<C>(raw[bi], uint32_t(((row >= p.src_h) ? p.src_h : row) * p.src_stride + lane_off) & 3u, win);
#pragma unroll
                for (int q = 0; q < C::SPAN; ++q) {
                    float u[CH];
#ifdef PICHA_SYM_ABLATE_HALO   // measurement builds only (wrong results): the halo pixels of a lane's window cost no unpack — the upper bound of what handing
                    // unpacked pixels between lanes could save
                    constexpr int AHL = (TX - SX - 1) / 2, AOWN = SX * P;
                    const int qq = q < AHL ? AHL : q >= AHL + AOWN ? AHL + AOWN - 1 : q;
                    if constexpr (C::BROT || XCH) unpack_span_px<C, CH, S16, FMA>(win, qq, u);
                    else unpack_span_px<C, CH, S16, FMA>(raw[bi], qq, u);
#else
                    if constexpr (C::BROT || XCH) unpack_span_px<C, CH, S16, FMA>(win, q, u);
                    else unpack_span_px<C, CH, S16, FMA>(raw[bi], q, u);
#endif
#pragma unroll
                    for (int o = 0; o < P; ++o) {
                        const int j = q - SX * o;  // tap index of span pixel q for output o
                        if (j >= 0 && j < TX) {
                            float w;
                            if constexpr (XC) w = wx[0][ce_min(j, TX - 1 - j)];
                            else w = wx[o][j];
#pragma unroll
                            for (int c = 0; c < CH; ++c) {
                                if (j == 0) h[o][c] = w * u[c];
                                else if constexpr (FMA) h[o][c] = __builtin_fmaf(w, u[c], h[o][c]);
                                else h[o][c] = h[o][c] + w * u[c];  // identical products (same w bits, same u) are computed once
                            }
                        }
                    }
                }
                // ---- fold into the pending outputs (src/resize.cc:125-129, one tap per pending output) ----
                auto fold = [&](auto yc_c) __attribute__((always_inline)) {
                    constexpr bool YC = decltype(yc_c)::value;
#pragma unroll
                    for (int i = 0; i < NPEND; ++i) {
                        const int j = SY * i + ph;
                        if (j < TY) {
                            const int sl = ((i - K) % NPEND + NPEND) % NPEND;
                            float w;
                            if constexpr (YC) w = wyc[ce_min(j, TY - 1 - j)];
                            else asm volatile("v_mov_b32 %0, %1" : "=v"(w) : "s"(wb[i * SY + ph]));
#pragma unroll
                            for (int o = 0; o < P; ++o)
#pragma unroll
                                for (int c = 0; c < CH; ++c) {
                                    if (j == 0) acc[sl][o][c] = w * h[o][c];
                                    else if constexpr (FMA) acc[sl][o][c] = __builtin_fmaf(w, h[o][c], acc[sl][o][c]);
                                    else acc[sl][o][c] = acc[sl][o][c] + w * h[o][c];
                                }
                        }
                    }
                };
                if (ycanon) fold(std::true_type{});
                else fold(std::false_type{});
                if (ph == PHCLOSE) {
                    const int y = m - ICLOSE;
                    if (y >= y0 && (PICHA_SYM_ABLATE != 1 || p.dst_w < 0)) {  // outputs above the chunk belong to the previous chunk
                        const int sl = ((ICLOSE - K) % NPEND + NPEND) % NPEND;
                        if constexpr (EPI) {
#pragma unroll
                            for (int o = 0; o < P; ++o)
                                if (x0 + o < p.dst_w) epilogue_store<CH, S16>(acc[sl][o], p.epi_pixel, p.epi_r, p.epi_g, p.epi_b, dst_rs, y, x0 + o, p.dst_stride);
                        } else {
                            store_lane_outputs<C, CH, S16, P, FMA>(acc[sl], dst_rs, uint32_t(y * p.dst_stride + x0 * C::BPP), x0, p.dst_w);
                        }
                    }
                }
            }
        };

#pragma unroll
        for (int d = 0; d < NB - 1; ++d) load_row(raw[d], SY * y0 + p.y_base + d);
        for (int m = y0; m <= m_end; m += USTEPS) {
            auto run = [&](auto self, auto kc) __attribute__((always_inline)) -> void {
                constexpr int k = decltype(kc)::value;
                if constexpr (k < USTEPS) {
                    if (k == 0 || m + k <= m_end) {
                        one_step(m + k, IC<k>{});
                        self(self, IC<k + 1>{});
                    }
                }
            };
            run(run, IC<0>{});
        }
    };
    // edge waves carry P*TX lane-resident weights: they give up one row of prefetch so that the kernel's register
    // allocation (the maximum over both versions) is set by the interior version
    if (interior) march(std::true_type{}, IC<NBUF>{});
    else march(std::false_type{}, IC<(NBUF > 2 ? NBUF - 1 : 2)>{});
    };  // run_item
    run_work_items(p.work, [&](int &ibx, int &y0, int &y1, int &img) __attribute__((always_inline)) {
        unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
        if (p.xcd_remap) xcd_contiguous_block(bx, by, bz);
        ibx = int(bx); y0 = int(by) * p.rows_per_chunk; y1 = min(p.dst_h, y0 + p.rows_per_chunk); img = int(bz);
    }, run_item);
    PICHA_PROBE_END
}

template <int CH, bool S16, int P, int SX, int TX, int SY, int TY, bool XCH = false, bool EPI = false>
hipError_t launch_sym_t(const StreamParams &p, hipStream_t s, const char **name) {
    const int bw = p.block_waves >= 1 && p.block_waves <= 4 ? p.block_waves : 2;
    dim3 grid((p.x_groups + 64 * bw - 1) / (64 * bw), p.n_chunks, p.n_images);
    if (p.work.items) grid = dim3(p.work.queue ? (unsigned)p.n_chunks : 8u * (unsigned)p.work.items_per_xcd, 1, 1);  // persistent: n_chunks = resident workgroups
    size_t lds = 0;
#if PICHA_SYM_ABLATE
    if (const char *e = getenv("PICHA_SYM_LDS")) lds = (size_t)atol(e);
#endif
    if (p.fma) resize_sym_kernel<CH, S16, P, SX, TX, SY, TY, true, XCH, EPI><<<grid, 64 * bw, lds, s>>>(p);
    else resize_sym_kernel<CH, S16, P, SX, TX, SY, TY, false, XCH, EPI><<<grid, 64 * bw, lds, s>>>(p);
    char epi[40] = "";
    if (EPI) snprintf(epi, sizeof epi, ",epilogue -> %s", kPixelNameAll[p.epi_pixel & 7]);
    snprintf(g_kernel_name, sizeof g_kernel_name, "resize_sym<ch%d,%s,P%d,SX%d,TX%d,SY%d,TY%d,%s,bw%d%s%s%s>", CH, S16 ? "u16" : "u8", P, SX, TX, SY,
             TY, p.fma ? "fma" : "exact", bw, XCH ? ",xch" : "", epi, p.work.queue ? ",persistent" : p.work.items ? ",items" : "");
    if (name) *name = g_kernel_name;
    return hipGetLastError();
}

template <int CH, bool S16, int P, int SX, int TX, int SY, int TY, bool XCH = false>
int sym_resident_t(const StreamParams &p) {
    const int bw = p.block_waves >= 1 && p.block_waves <= 4 ? p.block_waves : 2;
    int per_cu = 0, dev = 0, cus = 0;
    const void *k = p.fma ? reinterpret_cast<const void *>(resize_sym_kernel<CH, S16, P, SX, TX, SY, TY, true, XCH>)
                          : reinterpret_cast<const void *>(resize_sym_kernel<CH, S16, P, SX, TX, SY, TY, false, XCH>);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 64 * bw, 0) != hipSuccess) return 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return per_cu * cus;
}

// Instantiations per pixel format: X(P, SX, TX, SY, TY) — the (format, scale, tap count, outputs per lane) combinations that
// beat the general streaming / accumulator kernels in either numerics flavour (r02 survey of all 8 formats x 7 integer-scale
// cases, profiles/r02_survey_sym_{exact,fma}.txt; the planner's preference table is sym_preferred_per_lane in capi.cpp),
// plus rgba at 2x with 2..4 outputs per lane for the parity tests.  Tap counts of picha's filters at integer scales (all
// symmetric): 2x: 3 (triangle), 5 (default options), 7 (cubic / mitchel / catmulrom); 3x: 8, 12; 4x: 11, 15.  lanczos and
// box at even integer scales alias the reference's ring (src/resize.cc:52-64) and stay on the span kernel.
#define PICHA_SYM_LIST_0(X) /* rgb */ X(4, 2, 5, 2, 5) X(4, 3, 8, 3, 8) X(4, 4, 11, 4, 11) X(4, 4, 15, 4, 15)
#define PICHA_SYM_LIST_1(X) /* rgba */ X(2, 2, 5, 2, 5) X(3, 2, 5, 2, 5) X(4, 2, 5, 2, 5) X(2, 2, 7, 2, 7) X(3, 2, 7, 2, 7) \
    X(2, 3, 8, 3, 8) X(2, 4, 11, 4, 11) X(2, 4, 15, 4, 15)
#define PICHA_SYM_LIST_2(X) /* grey */ X(4, 2, 5, 2, 5) X(4, 2, 7, 2, 7)
#define PICHA_SYM_LIST_3(X) /* greya */ X(2, 2, 3, 2, 3) X(4, 2, 5, 2, 5) X(4, 2, 7, 2, 7) X(2, 3, 8, 3, 8) X(2, 3, 12, 3, 12)
#define PICHA_SYM_LIST_4(X) /* r16 */ X(2, 2, 3, 2, 3) X(4, 2, 3, 2, 3) X(4, 2, 5, 2, 5) X(4, 2, 7, 2, 7) X(2, 3, 8, 3, 8) \
    X(2, 3, 12, 3, 12) X(2, 4, 11, 4, 11)
#define PICHA_SYM_LIST_5(X) /* r16g16 */ X(2, 2, 3, 2, 3) X(2, 2, 5, 2, 5) X(4, 2, 5, 2, 5) X(3, 2, 7, 2, 7) X(4, 2, 7, 2, 7) \
    X(2, 3, 12, 3, 12) X(2, 4, 15, 4, 15)
#define PICHA_SYM_LIST_6(X) /* r16g16b16 */ X(2, 2, 3, 2, 3) X(2, 2, 5, 2, 5) X(2, 2, 7, 2, 7) X(2, 3, 12, 3, 12)
#define PICHA_SYM_LIST_7(X) /* r16g16b16a16 */ X(2, 2, 3, 2, 3) X(2, 2, 5, 2, 5) X(2, 2, 7, 2, 7) X(2, 3, 12, 3, 12)

// Neighbour-exchange instantiations: Y(P, SX, TX, SY, TY), dword-sized pixels only.  The form never beat the plain one in this
// kernel (profiles/r03_ab_xch.txt, r03_survey_xch.txt: +1 .. +21 %), so only the 2x / 5-tap shapes stay, for the parity tests
// and as a measuring point; 7-tap and 3x shapes were measured and dropped.
#define PICHA_SYMX_LIST_0(Y)
#define PICHA_SYMX_LIST_1(Y) /* rgba */ Y(2, 2, 5, 2, 5) Y(3, 2, 5, 2, 5) Y(4, 2, 5, 2, 5)
#define PICHA_SYMX_LIST_2(Y)
#define PICHA_SYMX_LIST_3(Y)
#define PICHA_SYMX_LIST_4(Y)
#define PICHA_SYMX_LIST_5(Y) /* r16g16 */ Y(2, 2, 5, 2, 5) Y(4, 2, 5, 2, 5)
#define PICHA_SYMX_LIST_6(Y)
#define PICHA_SYMX_LIST_7(Y) /* r16g16b16a16 */ Y(2, 2, 5, 2, 5)

// Colour-convert epilogue instantiations: E(P, SX, TX, SY, TY) — resize -> colorConvert of 8-bit rgba (and rgb in the fma flavour, whose
// planner choice at 2x is this kernel) at 2x, picha's default 5 taps and the 7 taps of {filter: ...}: the shapes whose two-kernel chain
// ran resize_sym + a convert (r03: the epilogue existed on the one-output accumulator kernel only)
#define PICHA_SYME_LIST_0(E) /* rgb */ E(4, 2, 5, 2, 5)
#define PICHA_SYME_LIST_1(E) /* rgba */ E(3, 2, 5, 2, 5)   /* 7 taps (P2, fma) measured 2-5 % behind the two-kernel chain: not instantiated */
#define PICHA_SYME_LIST_2(E)
#define PICHA_SYME_LIST_3(E)
#define PICHA_SYME_LIST_4(E)
#define PICHA_SYME_LIST_5(E)
#define PICHA_SYME_LIST_6(E)
#define PICHA_SYME_LIST_7(E) /* r16g16b16a16 */ E(2, 2, 5, 2, 5)

}  // namespace

#ifndef PICHA_SYM_PIXEL
#error "compile with -DPICHA_SYM_PIXEL=0..7"
#endif
#define PICHA_CAT2(a, b) a##b
#define PICHA_CAT(a, b) PICHA_CAT2(a, b)
#define PICHA_SYM_LIST PICHA_CAT(PICHA_SYM_LIST_, PICHA_SYM_PIXEL)
#define PICHA_SYMX_LIST PICHA_CAT(PICHA_SYMX_LIST_, PICHA_SYM_PIXEL)
#define PICHA_SYME_LIST PICHA_CAT(PICHA_SYME_LIST_, PICHA_SYM_PIXEL)

bool PICHA_CAT(sym_epi_supported_pixel, PICHA_SYM_PIXEL)(int per_lane, int sx, int tx, int sy, int ty) {
#define E(P, SX, TX, SY, TY) if (per_lane == P && sx == SX && tx == TX && sy == SY && ty == TY) return true;
    PICHA_SYME_LIST(E)
#undef E
    return false;
}

PICHA_PROBE_HOST(PICHA_CAT(picha_probe_sym_p, PICHA_SYM_PIXEL))

bool PICHA_CAT(sym_supported_pixel, PICHA_SYM_PIXEL)(int per_lane, int sx, int tx, int sy, int ty, int xch) {
#define X(P, SX, TX, SY, TY) if (!xch && per_lane == P && sx == SX && tx == TX && sy == SY && ty == TY) return true;
    PICHA_SYM_LIST(X)
#undef X
#define Y(P, SX, TX, SY, TY) if (xch && per_lane == P && sx == SX && tx == TX && sy == SY && ty == TY) return true;
    PICHA_SYMX_LIST(Y)
#undef Y
    return false;
}

int PICHA_CAT(sym_resident_pixel, PICHA_SYM_PIXEL)(const StreamParams &p) {
#define Y(P, SX, TX, SY, TY)                                                                                          \
    if (p.xch && p.x_per_lane == P && p.x_step == SX && p.x_taps == TX && p.y_step == SY && p.y_taps == TY)           \
        return sym_resident_t<pixel_channels(PICHA_SYM_PIXEL), pixel_is16(PICHA_SYM_PIXEL), P, SX, TX, SY, TY, true>(p);
    PICHA_SYMX_LIST(Y)
#undef Y
    if (p.xch) return 0;
#define X(P, SX, TX, SY, TY)                                                                                          \
    if (p.x_per_lane == P && p.x_step == SX && p.x_taps == TX && p.y_step == SY && p.y_taps == TY)                     \
        return sym_resident_t<pixel_channels(PICHA_SYM_PIXEL), pixel_is16(PICHA_SYM_PIXEL), P, SX, TX, SY, TY>(p);
    PICHA_SYM_LIST(X)
#undef X
    return 0;
}

hipError_t PICHA_CAT(launch_sym_pixel, PICHA_SYM_PIXEL)(const StreamParams &p, hipStream_t s, const char **name) {
#define E(P, SX, TX, SY, TY)                                                                                          \
    if (p.epi_on && !p.xch && p.x_per_lane == P && p.x_step == SX && p.x_taps == TX && p.y_step == SY && p.y_taps == TY) \
        return launch_sym_t<pixel_channels(PICHA_SYM_PIXEL), pixel_is16(PICHA_SYM_PIXEL), P, SX, TX, SY, TY, false, true>(p, s, name);
    PICHA_SYME_LIST(E)
#undef E
    if (p.epi_on) return hipErrorInvalidValue;
#define Y(P, SX, TX, SY, TY)                                                                                          \
    if (p.xch && p.x_per_lane == P && p.x_step == SX && p.x_taps == TX && p.y_step == SY && p.y_taps == TY)           \
        return launch_sym_t<pixel_channels(PICHA_SYM_PIXEL), pixel_is16(PICHA_SYM_PIXEL), P, SX, TX, SY, TY, true>(p, s, name);
    PICHA_SYMX_LIST(Y)
#undef Y
    if (p.xch) return hipErrorInvalidValue;
#define X(P, SX, TX, SY, TY)                                                                                          \
    if (p.x_per_lane == P && p.x_step == SX && p.x_taps == TX && p.y_step == SY && p.y_taps == TY)                     \
        return launch_sym_t<pixel_channels(PICHA_SYM_PIXEL), pixel_is16(PICHA_SYM_PIXEL), P, SX, TX, SY, TY>(p, s, name);
    PICHA_SYM_LIST(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace picha_hip
