// kernels_run.hip — downscale kernel for PERIODIC tap tables with long windows ("run" kernel, r03): a lane owns a run of P
// consecutive outputs of one row and resamples them from one contiguous window of the source row held in registers.
//
// Where it applies: the scale src/dst is a ratio PX/Q of small integers (7.5 = 15/2: a 4K frame to 512 columns), so that
// the reference's tap table (src/resize.cc:19-50) repeats every Q outputs — output x + Q has the weights of output x, bit
// for bit, and its window starts PX source pixels later — for every output whose window does not hang over an image edge.
// Then
//   * the weights are WAVE-UNIFORM: Q x TX registers for the whole wave instead of P x TX lane-resident ones (the
//     accumulator kernel needs 21-32 per output), which is what makes runs of 4-8 outputs per lane affordable;
//   * a lane loads and converts the union of its outputs' windows ONCE per row: floor((P-1) PX / Q) + TX pixels for P
//     outputs — at 7.5x / 21 taps 10.8 pixels per output for P = 4 against 21-24 for the one-output accumulator kernel
//     and 15 + LDS traffic for the row-staged kernel (kernels_rows.hip), whose 8-pixel window alignment also rounds 21
//     taps up to 32 multiply-adds;
//   * no LDS, no barrier; the vertical part is the accumulator kernel's (per-source-row records, NACC pending output rows
//     in registers), so every sum adds the reference's products in the reference's order: bit-exact in the reference flavour
//     (src/resize.cc:110-117, 125-129).
// The few columns whose windows are clipped by an image edge (renormalised weights: 4 of 512 for cfg3b) come out of the march
// with the canonical weights and are then OVERWRITTEN by resize_edges_kernel (same stream, launched after it): the reference's
// double sum with those columns' own weights, from a stash of raw pixels that the two lanes holding such columns leave behind
// (edge_pass below; how the other forms of this step measured: profiles/r03_ab_run_kernel.txt, LAB.md 5.4b).
#define PICHA_SPAN_ONLY 1
#include "kernels_impl.inc"
#include <cstring>
#ifndef PICHA_RUN_NB
#define PICHA_RUN_NB 3
#endif
#ifndef PICHA_RUN_XNB
#define PICHA_RUN_XNB 2   // row buffers of the neighbour-exchange form
#endif
// measurement builds only (scripts/r04_ablate.sh; results are wrong by construction): 1 = loads and stores without the arithmetic,
// 2 = arithmetic and stores without the loads in the loop, 3 = no halo exchange (the lane's own pixels stand in for its neighbours'),
// 4 = everything but the output stores (one row per chunk is stored), 5 = window pixels outside the lane's own cost no unpack
#ifndef PICHA_RUN_ABLATE
#define PICHA_RUN_ABLATE 0
#endif
// measurement builds: -DPICHA_RUN_WAVES=N makes the compiler fit resize_run_kernel into the registers of N waves per SIMD
// ... and -DPICHA_RUN_WS_AS=2 launches the shapes listed with one scalar phase with BOTH phases' weights in SGPRs
#ifdef PICHA_RUN_WS_AS
#define PICHA_RUN_WS_OF(WS) ((WS) == 1 ? PICHA_RUN_WS_AS : (WS))
#else
#define PICHA_RUN_WS_OF(WS) (WS)
#endif
#ifdef PICHA_RUN_WAVES
#define PICHA_RUN_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(PICHA_RUN_WAVES, PICHA_RUN_WAVES)))
#else
#define PICHA_RUN_WAVES_ATTR
#endif

namespace picha_hip {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// first pixel of output o's window relative to the window of output 0 of the same lane
// (RC = 1: the offsets inside a period round up instead of down — 7.5x with 30 taps: windows at 0, 8, 15, 23 need no padding tap)
constexpr int run_off(int Q, int PX, int o, int RC = 0) { return (o / Q) * PX + ((o % Q) * PX + RC * (Q - 1)) / Q; }

template <int CH, bool S16, int Q, int PX, int TX, int P, int RC = 0> struct RunCfg {
    static constexpr int BPS = Sample<S16>::kBytes;
    static constexpr int BPP = CH * BPS;
    static constexpr int SPAN = run_off(Q, PX, P - 1, RC) + TX;  // source pixels a lane reads per row
    static constexpr bool WORDS = (BPP % 4 == 0);
    static constexpr bool BYTESPAN = !WORDS;
    static constexpr bool BROT = !WORDS;                     // sub-dword pixels: aligned dwords + v_alignbyte by the lane's byte phase
    static constexpr int NWIN = (SPAN * BPP + 3) / 4;
    static constexpr int NW = WORDS ? SPAN * BPP / 4 : NWIN + 1;
};

// window of one edge-column pixel: TXE >= the table's taps, whole dwords
template <int CH, bool S16, int TXE> struct EdgeCfg {
    static constexpr int BPS = Sample<S16>::kBytes;
    static constexpr int BPP = CH * BPS;
    static constexpr int SPAN = TXE;
    static constexpr bool WORDS = (BPP % 4 == 0);
    static constexpr bool BYTESPAN = !WORDS;
    static constexpr bool BROT = !WORDS;
    static constexpr int NWIN = (SPAN * BPP + 3) / 4;
    static constexpr int NW = WORDS ? SPAN * BPP / 4 : NWIN + 1;
};

// Colour convert in front of the resize (r04; convert -> upscale as ONE kernel): span pixel q of a window loaded in the SOURCE format Cs goes
// through the reference's per-pixel convert (src/colorconvert.cc:136-152: unpack, ChannelConvertOp by channel count, pack) — its result is the
// intermediate image's QUANTISED sample, which is what picha.resize would then read — and comes out unpacked the way the kernel's own format
// CH / S16 unpacks (unpack_span_px: plain integer for fma, value x 1/max, or the 2^-22-scaled denormal form).  Pairs of equal sample depth that
// only move samples around (no luma) skip the float round trip: pack(unpack(s)) == s for every sample, pack(1.0f) == max, pack(0.0f) == 0.
template <typename Cs, int SCH, bool SS16, int CH, bool S16, bool FMA>
__device__ __forceinline__ void convert_span_px(const uint32_t *raw, int q, float *u, float r, float g, float b) {
    uint32_t sv[SCH];
#pragma unroll
    for (int c = 0; c < SCH; ++c) {
        const int bpos = q * Cs::BPP + c * Cs::BPS;   // byte position inside the span: a constant after unrolling
        sv[c] = (raw[bpos / 4] >> (8 * (bpos % 4))) & (SS16 ? 0xffffu : 0xffu);
    }
    uint32_t dv[CH];
    constexpr bool kLuma = SCH >= 3 && CH <= 2;
    if constexpr (SS16 == S16 && !kLuma) {
        constexpr uint32_t kOne = S16 ? 65535u : 255u;
        if constexpr (SCH == CH) {
#pragma unroll
            for (int c = 0; c < CH; ++c) dv[c] = sv[c];
        } else if constexpr (SCH == 1) {
            dv[0] = sv[0];
            if constexpr (CH == 2) dv[1] = kOne;
            if constexpr (CH >= 3) { dv[1] = sv[0]; dv[2] = sv[0]; }
            if constexpr (CH == 4) dv[3] = kOne;
        } else if constexpr (SCH == 2) {
            dv[0] = sv[0];
            if constexpr (CH == 1) { }
            if constexpr (CH == 3) { dv[1] = sv[1]; dv[2] = 0u; }
            if constexpr (CH == 4) { dv[1] = sv[0]; dv[2] = sv[0]; dv[3] = sv[1]; }
        } else {  // 3 -> 4 or 4 -> 3
            dv[0] = sv[0]; dv[1] = sv[1]; dv[2] = sv[2];
            if constexpr (CH == 4) dv[3] = kOne;
        }
    } else {
        float fs[SCH], fd[CH];
#pragma unroll
        for (int c = 0; c < SCH; ++c) fs[c] = unpack_sample<SS16>(sv[c]);
        channel_map<SCH, CH>(fs, fd, r, g, b);
#pragma unroll
        for (int c = 0; c < CH; ++c) dv[c] = pack_sample<S16>(fd[c]);
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        if constexpr (FMA) u[c] = float(dv[c]);
        else if constexpr (exact_unpack_scaled(CH, S16)) u[c] = mul_denorm_bits<6>(dv[c], Sample<S16>::kDen);
        else u[c] = unpack_sample<S16>(dv[c]);
    }
}

// Edge columns [xa0, xa0 + na) of one side of one image for the output rows [y0, y1), by one wave; the reference's double sum
// (src/resize.cc:110-117 into a zeroed row, :125-129 over the rows the ring holds for the window's taps: y_eff) with the columns' own weights from the generic tables, reference arithmetic in
// both numerics flavours.  The source pixels come from the STASH the march wrote: per (image, side, source row) the NRAW raw
// dwords of the lane that holds the side's edge columns (stash_rs describes this image-side's rows, st_stride bytes apart; the
// side's lane window starts st_org bytes into the image row, a multiple of 4, possibly negative).  Through kEdgeFloats floats of LDS:
//   0  the columns' x weights and the block's y weights go to LDS (coalesced loads);
//   A  one lane per (edge column, SOURCE row of the block): the horizontal sum; lanes read consecutive stash rows;
//   B  one lane per (edge column, output row): the vertical sum over the stored horizontal sums, pack, store.
// Why a stash: fetched from the image itself every window costs a 128-byte line of HBM for 30 bytes of payload, once more than
// the march already paid (141-283 MB for cfg3b: 0.03-0.07 ms, alone or as an epilogue of the march's waves, whatever the form
// of the loop — r03, profiles/r03_ab_run_kernel.txt); the march has those bytes in registers.
constexpr int kEdgeFloats = 2048;
template <int CH, bool S16, int TXE, int SRCP = -1, int GOV = 0>
__device__ __forceinline__ void edge_pass(const RunParams &p, __amdgpu_buffer_rsrc_t stash_rs, int st_stride, int st_org, __amdgpu_buffer_rsrc_t dst_rs,
                                          int lane, float *lds, int xa0, int na, int y0, int y1) {
    constexpr int SCH = SRCP < 0 ? CH : pixel_channels(SRCP < 0 ? 0 : SRCP);
    constexpr bool SS16 = SRCP < 0 ? S16 : pixel_is16(SRCP < 0 ? 0 : SRCP);
    using E = EdgeCfg<SCH, SS16, TXE>;   // the windows as they are loaded (SRCP >= 0: the source format of a convert in front)
    using ED = EdgeCfg<CH, S16, TXE>;
    constexpr bool SC = exact_unpack_scaled(CH, S16);  // such formats unpack through the denormal form: weights times 2^22 (exact)
    const_i32_ptr xl = (const_i32_ptr)(p.x_left);
    const_i32_ptr yrange = (const_i32_ptr)(p.yrow_range);
    const int ncol = na;
    float *lds_wx = lds;                       // [ncol][TXE]
    for (int ya = y0; ya < y1; ya += p.edge_rows) {
        const int yb = min(y1, ya + p.edge_rows), rows = yb - ya;
        float *lds_wy = lds_wx + ncol * TXE;   // [rows][y_taps]
        float *lds_h = lds_wy + rows * p.y_taps;  // [ncol][nsr][CH]
        // source rows of the block (the march's table: first row needed by ya or any later row, last row needed up to yb - 1)
        const int ra = yrange[2 * ya], rb = yrange[2 * (yb - 1) + 1];
        const int nsr = rb - ra + 1;  // ncol * nsr * CH floats from lds_h on: host-sized blocks (capi.cpp, run_group)
        // ---- 0: weights to LDS ----
        for (int i = lane; i < ncol * TXE; i += kWave) {
            const int c = i / TXE, j = i % TXE;
            const int x = xa0 + c;
            const float w = p.x_weight[(size_t)x * p.x_wtaps + min(j, p.x_wtaps - 1)];
            lds_wx[i] = j < p.x_count[x] ? (SC ? w * kWeightScale : w) : 0.0f;
        }
#pragma unroll 8
        for (int i = lane; i < rows * p.y_taps; i += kWave) lds_wy[i] = p.y_weight[(size_t)ya * p.y_taps + i];
        const int yl = min(ya + lane, yb - 1);     // phase B's row of this lane (block rows <= 64)
        const int ny_l = p.y_count[yl];
        const int32_t *eff_l = p.y_eff + (size_t)yl * p.y_taps;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- A: horizontal sums, one column after the other; G groups of 64 source rows (one window per lane) in flight ----
        // (no run-time integer division anywhere in this kernel: its expansion uses v_fmac, which the ISA test reserves for the fma flavour)
        constexpr int G = GOV > 0 ? GOV : E::NW <= 12 ? 4 : 2;   // (GOV: the pass inside the march's kernel must stay under the march's register count)
        for (int c = 0; c < ncol; ++c) {
            const int32_t bo = xl[xa0 + c] * E::BPP - st_org;   // window start inside the stash row: >= 0 (host-checked containment)
            const uint32_t phase = uint32_t(bo) & 3u;
            for (int s0 = 0; s0 < nsr; s0 += G * kWave) {
                uint32_t raw[G][E::NW];
#pragma unroll
                for (int g = 0; g < G; ++g)
                    load_span<E, false>(stash_rs, (ra + min(s0 + g * kWave + lane, nsr - 1)) * st_stride + (bo & ~3), raw[g]);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (s0 + g * kWave < nsr) {  // wave-uniform
                        const int sr = s0 + g * kWave + lane;
                        uint32_t win[E::BROT ? E::NWIN : 1];
                        if constexpr (E::BROT) rotate_span<E>(raw[g], phase, win);
                        float h[CH];
#pragma unroll
                        for (int t = 0; t < TXE; ++t) {
                            float u[CH];
                            if constexpr (SRCP >= 0) convert_span_px<E, SCH, SS16, CH, S16, false>(E::BROT ? win : raw[g], t, u, p.cv_r, p.cv_g, p.cv_b);
                            else if constexpr (E::BROT) unpack_span_px<E, CH, S16, false>(win, t, u);
                            else unpack_span_px<E, CH, S16, false>(raw[g], t, u);
                            const float w = lds_wx[c * TXE + t];  // every lane the same address
#pragma unroll
                            for (int ch = 0; ch < CH; ++ch) h[ch] = t == 0 ? w * u[ch] : h[ch] + w * u[ch];
                        }
                        if (sr < nsr) {
#pragma unroll
                            for (int ch = 0; ch < CH; ++ch) lds_h[(c * nsr + sr) * CH + ch] = h[ch];
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- B: vertical sums; lane = output row, one column after the other ----
        int ny_max = ny_l;  // wave maximum
#pragma unroll
        for (int m = 1; m < kWave; m <<= 1) ny_max = max(ny_max, __shfl_xor(ny_max, m));
        ny_max = __builtin_amdgcn_readfirstlane(ny_max);
        for (int c = 0; c < ncol; ++c) {
            const int x = xa0 + c;
            float acc[CH];
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) acc[ch] = 0.0f;
            for (int j = 0; j < ny_max; ++j) {
                const float wj = j < ny_l ? lds_wy[(yl - ya) * p.y_taps + j] : 0.0f;
                const int sr = min(max(eff_l[min(j, ny_l - 1)] - ra, 0), nsr - 1);   // (the row the reference's ring holds for this tap)
#pragma unroll
                for (int ch = 0; ch < CH; ++ch) acc[ch] = acc[ch] + wj * lds_h[(c * nsr + sr) * CH + ch];
            }
            if (ya + lane < yb) {
                uint32_t pw[2];
                pack_px_words<CH, S16, false>(acc, pw);
                store_px_words<CH, S16>(dst_rs, uint32_t(yl * p.dst_stride + x * ED::BPP), pw);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// One wave per (block of p.edge_rows output rows, side of the image, image): launched after the march on the same stream, it
// overwrites what the march stored in those columns with the canonical weights.  (As an epilogue of the march's waves the pass
// cost 25 VGPRs on top of the march's and 5-12 % of its time.)
template <int CH, bool S16, int TXE, int SRCP = -1>
__global__ __launch_bounds__(64) void resize_edges_kernel(RunParams p) {
    extern __shared__ __attribute__((aligned(16))) float edge_lds[];
    const int lane = threadIdx.x;
    const int img = blockIdx.z, side = blockIdx.y;
    const int y0 = blockIdx.x * p.edge_rows, y1 = min(p.dst_h, y0 + p.edge_rows);
    const int xa0 = side == 0 ? 0 : p.cx1;
    const int na = side == 0 ? p.cx0 : p.dst_w - p.cx1;
    if (y0 >= y1 || na <= 0) return;
    uint8_t *dbase = uniform_ptr(p.dst_tab ? static_cast<uint8_t *>(p.dst_tab[img]) : p.dst + img * p.dst_pitch);
    const __amdgpu_buffer_rsrc_t dst_rs = __builtin_amdgcn_make_buffer_rsrc(dbase, 0, p.dst_bytes, 0x00020000);
    if (p.edge_image) {
        // (r04) no stash: the windows come from the image itself.  The descriptor starts on the dword at or below the image and ends on
        // the one at or above its last byte; a window that runs past its row meets zero weights (or, past the image, reads 0)
        const uint8_t *sbase = uniform_ptr(p.src_tab ? static_cast<const uint8_t *>(p.src_tab[img]) : p.src + img * p.src_pitch);
        const int sadj = int(reinterpret_cast<uintptr_t>(sbase) & 3u);
        const __amdgpu_buffer_rsrc_t img_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(sbase) - sadj, 0, (p.src_bytes + sadj + 3u) & ~3u, 0x00020000);
        edge_pass<CH, S16, TXE, SRCP>(p, img_rs, p.src_stride, -sadj, dst_rs, lane, edge_lds, xa0, na, y0, y1);
        return;
    }
    if constexpr (SRCP >= 0) return;   // (a convert in front always reads the image)
    const unsigned st_stride = unsigned(p.stash_words) * 4u, st_bytes = st_stride * unsigned(p.src_h);
    const __amdgpu_buffer_rsrc_t stash_rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<uint8_t *>(p.stash) + (size_t)((p.stash_img0 + img) * 2 + side) * st_bytes, 0, st_bytes, 0x00020000);
    edge_pass<CH, S16, TXE>(p, stash_rs, int(st_stride), p.stash_org[side], dst_rs, lane, edge_lds, xa0, na, y0, y1);
}

// (r05, third session) the same pass as one wave of a workgroup INSIDE the march's grid (RunParams::edge_fused): always from the image
template <int CH, bool S16, int TXE>
__device__ __forceinline__ void edge_block_from_image(const RunParams &p, float *lds, int lane, int img, int side, int y0, int y1) {
    const int xa0 = side == 0 ? 0 : p.cx1;
    const int na = side == 0 ? p.cx0 : p.dst_w - p.cx1;
    if (y0 >= y1 || na <= 0) return;
    uint8_t *dbase = uniform_ptr(p.dst_tab ? static_cast<uint8_t *>(p.dst_tab[img]) : p.dst + img * p.dst_pitch);
    const __amdgpu_buffer_rsrc_t dst_rs = __builtin_amdgcn_make_buffer_rsrc(dbase, 0, p.dst_bytes, 0x00020000);
    const uint8_t *sbase = uniform_ptr(p.src_tab ? static_cast<const uint8_t *>(p.src_tab[img]) : p.src + img * p.src_pitch);
    const int sadj = int(reinterpret_cast<uintptr_t>(sbase) & 3u);
    const __amdgpu_buffer_rsrc_t img_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(sbase) - sadj, 0, (p.src_bytes + sadj + 3u) & ~3u, 0x00020000);
    edge_pass<CH, S16, TXE, -1, 1>(p, img_rs, p.src_stride, -sadj, dst_rs, lane, lds, xa0, na, y0, y1);
}
// shapes of resize_run_kernel that carry the edge blocks in their own grid: cfg3b's shared-unpack shape (grey 7.5x, 30 taps, four outputs per lane)
__host__ __device__ constexpr bool run_edges_fused(int ch, bool s16, int q, int px, int tx, int p, int xhl) { return kRunEfuseBuilt && ch == 1 && !s16 && q == 2 && px == 15 && tx == 30 && p == 4 && xhl == -2; }

// XHL >= 0 (r04): NEIGHBOUR EXCHANGE for dword-sized pixels.  A lane loads only the (P / Q) * PX pixels it owns — the wave reads one
// contiguous run per row, no byte is requested twice (cfg5's plain form asks for every source byte 4.75 times and is bound by the L1's
// request rate, profiles/r03_cfg5_before_rocprofv3_summary.txt) — and takes the XHL pixels in front of them from lane - 1 and the
// SPAN - OWN - XHL behind them from lane + 1 with DPP wave shifts of the RAW dwords (before conversion: products and sums are the
// reference's, bit for bit).  What lanes 0 and 63 miss — the wave's own halo, NH <= 64 dwords — is loaded ONE dword per lane with
// the row (one more register per row buffer, where the r03 accumulator form kept 12 in all 64 lanes for what two of them use) and
// handed to the two edge lanes through 256 bytes of wave-private LDS when the row is resampled.  The columns an image edge clips
// come out with the canonical weights here and are overwritten by resize_edges_kernel reading the IMAGE (no stash: p.edge_image).
// WS (r04): the weights of the last WS phases of the period stay in SGPRs (alone an SGPR-operand product issues at half rate, inside a multiply-add
// stream it costs the same slot as any other: profiles/r05_ubench_valu_issue.txt; Q x TX lane-uniform VGPRs are what kept cfg3b at 3 waves per SIMD).
template <int CH, bool S16, int Q, int PX, int TX, int P, int XHL> struct RunXch {
    static constexpr int BPP = CH * Sample<S16>::kBytes;
    static constexpr int DPX = BPP / 4;
    static constexpr int OWN = (P / Q) * PX;                       // pixels a lane loads itself
    static constexpr int SPAN = run_off(Q, PX, P - 1, 0) + TX;
    static constexpr int HL = XHL < 0 ? 0 : XHL, HR = SPAN - OWN - HL;
    static constexpr int NOWN = OWN * DPX, NHL = HL * DPX, NHR = HR * DPX, NH = NHL + NHR;
    static constexpr bool OK = BPP % 4 == 0 && HR >= 0 && HL <= OWN && HR <= OWN && NH <= kWave;
};
// XHL == -2 (r05): SHARED UNPACK for 8-bit sub-dword pixels (grey, greya).  In the plain form a lane converts its whole window — at 7.5x / 30 taps 53
// pixels for four outputs — although the 23 behind its own 30 are the FIRST 23 of lane + 1's window, which that lane converts too (src/resize.cc:110-117
// redoes the unpack per tap; any arrangement that yields the same float per pixel is the same arithmetic).  Here a lane loads and converts only the
// OWN = (P / Q) * PX pixels its window starts with, leaves the first NF = (SPAN - OWN) * CH floats of them in wave-private LDS (16-byte chunks,
// chunk-major: conflict-free b128 writes and reads) and takes the rest of its window from lane + 1's slots.  Lane 63's neighbour is a virtual lane 64:
// the NF samples behind the wave's run are loaded one BYTE per lane with the row (lane k: sample k), converted by the lanes that loaded them and
// written to lane 64's slots.  No stash (the lane no longer holds its whole window): the clipped columns come from the image (p.edge_image).
// Measured upper bound before it was built: -7.7 % on cfg3b (profiles/r04_ablate_run_kernel.txt, PICHA_RUN_ABLATE=5).
template <int CH, bool S16, int Q, int PX, int TX, int P, int RC> struct RunShu {
    static constexpr int BPP = CH * Sample<S16>::kBytes;
    static constexpr int OWN = (P / Q) * PX;
    static constexpr int SPAN = run_off(Q, PX, P - 1, RC) + TX;
    static constexpr int HR = SPAN - OWN;            // window pixels held by the lanes behind this one
    // (r05) windows that reach over TWO neighbours (lanczos thumbnails: 36 taps at 8x with two outputs per lane): lane + 1 gives all it owns, lane + 2 the rest
    static constexpr int HR1 = HR < OWN ? HR : OWN, HR2 = HR - HR1;
    static constexpr int NF = HR1 * CH;              // floats taken from lane + 1 (= what every lane leaves in its slots)
    static constexpr int NFC = (NF + 3) / 4;         // ... in 16-byte chunks
    static constexpr int NF2 = HR2 * CH, NFC2 = (NF2 + 3) / 4;   // ... from lane + 2
    static constexpr int VL = HR2 > 0 ? 2 : 1;       // virtual lanes behind the wave
    static constexpr int LANES = kWave + VL;         // slots per chunk
    // the samples behind the wave's run: one BYTE per lane (lane k: sample k) where they are at most 64, else one PIXEL per lane
    static constexpr bool HALO_PX = HR * CH > kWave;
    static constexpr bool OK = !S16 && BPP < 4 && HR > 0 && HR <= 2 * OWN && HR <= kWave && NFC * 4 <= OWN * CH + 3;
};
// shapes of resize_run_kernel that can produce the clipped edge columns themselves (EIN): the neighbour-exchange forms with all weights in lane registers
__host__ __device__ constexpr bool run_edges_inside(int xhl, int ws, int tx, int p) { return xhl >= 0 && ws == 0 && p * tx <= 32; }
// PH (r04): TIME-PHASED STORES.  A finished output row is packed and kept in registers (up to kPhaseRows rows per lane); the wave stores
// what it holds when the chip-wide 100 MHz counter (s_memrealtime) crosses a multiple of 2^phase_shift ticks — every wave of
// every workgroup in the same few microseconds — or when it holds kPhaseRows rows, or at the end of its chunk.  Why: in cfg5's march the
// stores are 6 % of the bytes but cost 20-30 % of the time (profiles/r04_ubench_march_pattern.txt: the read pattern alone streams at
// 6.1 TB/s, with the stores at 4.9-5.7 depending on where the destination was allocated; even 0.7 % of stores cost 10 %): a thin,
// steady trickle of writes makes every HBM channel turn its bus around for a line or two at a time.  Bunched into chip-wide bursts
// every ~40 us the same bytes cost half as much.  (At cfg2's 4:1 read:write ratio the write queues are never thin and phasing loses.)
#ifndef PICHA_RUN_PHASE_ROWS
#define PICHA_RUN_PHASE_ROWS 8
#endif
constexpr int kPhaseRows = PICHA_RUN_PHASE_ROWS;
template <int CH, bool S16, int Q, int PX, int TX, int P, int NACC, bool FMA, int RC = 0, int XHL = -1, int WS = 0, int PH = 0>
__global__ __launch_bounds__(256) PICHA_RUN_WAVES_ATTR void resize_run_kernel(RunParams p) {
    using C = RunCfg<CH, S16, Q, PX, TX, P, RC>;
    using X = RunXch<CH, S16, Q, PX, TX, P, XHL>;
    constexpr bool XCH = XHL >= 0;
    constexpr bool SHU = XHL == -2;
    using S = RunShu<CH, S16, Q, PX, TX, P, RC>;
    using CO = EdgeCfg<CH, S16, S::OWN>;   // the lane's own pixels as a byte span (shared-unpack form)
    static_assert(!SHU || (S::OK && !PH), "shared unpack: 8-bit sub-dword pixels, the neighbour's share within its own pixels");
    static_assert(P % Q == 0, "a lane owns whole periods");
    static_assert(!XCH || (X::OK && RC == 0), "neighbour exchange: dword pixels, halo within one lane's pixels");
    static_assert(WS >= 0 && WS <= Q, "phases with scalar weights");
    static_assert(!PH || C::WORDS, "phased stores: dword-sized pixels");
    static_assert(!SHU || (S::NFC - 1) * 4 < S::OWN * CH, "shared unpack: chunks of own floats");
    constexpr int ODW = PH ? P * (C::BPP / 4) : 1;   // dwords of a lane's packed output row
    constexpr int NRAW = XCH ? X::NOWN + 1 : SHU ? CO::NW + 1 : C::NW;
    constexpr int NB = XCH ? PICHA_RUN_XNB : SHU ? 2 : (NRAW <= 16 ? PICHA_RUN_NB : 2);  // row buffers
    PICHA_PROBE_BEGIN
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bw = __builtin_amdgcn_readfirstlane(blockDim.x >> 6);
    constexpr bool EFUSE = run_edges_fused(CH, S16, Q, PX, TX, P, XHL);
    if constexpr (EFUSE) {
        if (p.edge_fused && int(blockIdx.x) >= p.edge_bx0) {   // (workgroup-uniform) the two edge blocks of this (row chunk, image): wave 0 works, in the workgroup's exchange slots
            extern __shared__ __attribute__((aligned(16))) float efuse_lds[];
            if (wave == 0)
                edge_block_from_image<CH, S16, (TX + 3) / 4 * 4>(p, efuse_lds, lane, blockIdx.z, int(blockIdx.x) - p.edge_bx0, blockIdx.y * p.rows_per_chunk,
                                                                 min(p.dst_h, (int(blockIdx.y) + 1) * p.rows_per_chunk));
            return;
        }
    }
    auto run_item = [&](int bx, int y0, int y1, int img) __attribute__((always_inline)) {
    const int grp0 = (bx * bw + wave) * kWave;
    if (grp0 * P >= p.dst_w) return;
    const int x0 = (grp0 + lane) * P;

    const uint8_t *sbase = uniform_ptr(p.src_tab ? static_cast<const uint8_t *>(p.src_tab[img]) : p.src + img * p.src_pitch);
    uint8_t *dbase = uniform_ptr(p.dst_tab ? static_cast<uint8_t *>(p.dst_tab[img]) : p.dst + img * p.dst_pitch);
    const int sadj = C::BROT ? int(reinterpret_cast<uintptr_t>(sbase) & 3u) : 0;
    const __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(sbase) - sadj, 0, C::WORDS ? p.src_bytes : ((p.src_bytes + sadj + 3u) & ~3u), 0x00020000);
    const __amdgpu_buffer_rsrc_t dst_rs = __builtin_amdgcn_make_buffer_rsrc(dbase, 0, p.dst_bytes, 0x00020000);
    if (y0 >= y1) return;

    // canonical weights of the Q outputs of a period: scalar loads, then VGPRs (an SGPR operand halves the VALU rate)
    // (all scalar loads first, as wide as they merge, then the moves: one load and one wait per weight cost ~10 us per wave)
    const_i32_ptr xc = (const_i32_ptr)(p.x_canon);
    int32_t wbits[Q * TX];
#pragma unroll
    for (int j = 0; j < Q * TX; ++j) wbits[j] = xc[j];
    // EIN (r05, as in resize_ring_kernel): the neighbour-exchange shapes hold their weights in lane registers anyway — one set per OUTPUT of the lane (<= 32 registers)
    // lets the lanes at the image's sides take the clipped columns' own zero-padded weights, and the march produces every column itself (cfg5: the edge kernel was 2 % of the launch)
    constexpr bool EIN = run_edges_inside(XHL, WS, TX, P);
    constexpr int NWX = EIN ? P : Q;
    float wx[NWX][TX];
#pragma unroll
    for (int q = 0; q < NWX; ++q)
#pragma unroll
        for (int j = 0; j < TX; ++j) {
            if (EIN || q < Q - WS) asm("v_mov_b32 %0, %1" : "=v"(wx[q][j]) : "s"(wbits[(q % Q) * TX + j]));
            else wx[q][j] = __int_as_float(wbits[q * TX + j]);   // stays scalar: an SGPR operand of its products
        }
    if constexpr (EIN) {
        if (p.edge_in) {
#pragma unroll
            for (int o = 0; o < P; ++o) {
                const int x = x0 + o;
                if (x < p.dst_w && (x < p.cx0 || x >= p.cx1)) {   // (lane-varying: only the lanes at the image's sides)
                    const int first = p.x_left[x] - (p.x_base + (x0 / Q) * PX + run_off(Q, PX, o, RC));   // host-checked: 0 <= first, first + count <= TX
                    const int count = p.x_count[x];
#pragma unroll
                    for (int j = 0; j < TX; ++j) {
                        const int t = j - first;
                        float w = (t >= 0 && t < count) ? p.x_weight[(size_t)x * p.x_wtaps + min(max(t, 0), p.x_wtaps - 1)] : 0.0f;
                        if constexpr (FMA) w = w * Sample<S16>::kInv;
                        else if constexpr (exact_unpack_scaled(CH, S16)) w = w * kWeightScale;
                        wx[o][j] = w;
                    }
                }
            }
        }
    }
    // the lane's window starts at x_base + PX * (x0 / Q): x0 is a multiple of P, P of Q
    // (neighbour exchange: the lane's OWN pixels, XHL pixels into the window; x_base == -XHL is host-checked)
    const int32_t lane_off = XCH ? (grp0 + lane) * X::OWN * C::BPP : (p.x_base + (grp0 + lane) * ((P / Q) * PX)) * C::BPP + sadj;
    // a window that starts before the image (left-most lane) must not wrap a merged load in row 0
    const bool careful = !XCH && __any(lane_off < 0);
    // neighbour exchange: lane k < NHL fetches dword k of the NHL dwords in front of the wave's run, lane NHL + k dword k of the NHR
    // behind it (the other lanes repeat the last one); in front of the image the offset wraps and the load reads 0 (row 0) or the
    // end of the previous row: only clipped columns see it, and resize_edges_kernel overwrites those
    const int32_t halo_k = min(lane, X::NH - 1);
    const int32_t halo_off = halo_k < X::NHL ? grp0 * X::OWN * C::BPP - X::NHL * 4 + 4 * halo_k
                                             : (grp0 + kWave) * X::OWN * C::BPP + 4 * (halo_k - X::NHL);
    __shared__ uint32_t halo_lds[XCH ? 4 * kWave : 1];
    // shared unpack: lane k fetches sample k of the NF behind the wave's run (the other lanes repeat the last one)
    [[maybe_unused]] const int32_t shu_halo_off = S::HALO_PX ? (p.x_base + (grp0 + kWave) * S::OWN + min(lane, S::HR - 1)) * C::BPP + sadj
                                                             : (p.x_base + (grp0 + kWave) * S::OWN) * C::BPP + sadj + min(lane, S::HR * CH - 1);
    auto load_halo = [&](int row) __attribute__((always_inline)) -> uint32_t {
        const uint32_t off = uint32_t(row * p.src_stride + shu_halo_off);
        uint32_t v = __builtin_amdgcn_raw_buffer_load_b8(src_rs, off, 0, 0);
        if constexpr (S::HALO_PX && C::BPP >= 2) v |= uint32_t(__builtin_amdgcn_raw_buffer_load_b8(src_rs, off + 1, 0, 0)) << 8;
        if constexpr (S::HALO_PX && C::BPP >= 3) v |= uint32_t(__builtin_amdgcn_raw_buffer_load_b8(src_rs, off + 2, 0, 0)) << 16;
        return v;
    };

    const_i32_ptr yrange = (const_i32_ptr)(p.yrow_range);
    const_i32_ptr yrec = (const_i32_ptr)(p.yrow_rec);
    const int r0 = yrange[2 * y0];
    const int r1 = yrange[2 * (y1 - 1) + 1];
    // ring-aliased vertical tables (r05; lanczos at integer scales, as in resize_acc_kernel): r / r0 / r1 are positions in the list p.yrow_src of source rows, in which the
    // row an aliased first tap reads appears once more ahead of its window; src_of() gives the source row (identity without a list).  Host-checked: position 0 is row 0's only place.
    const_i32_ptr ysrc = (const_i32_ptr)(p.yrow_src);
    auto src_of = [&](int i) __attribute__((always_inline)) { return ysrc != nullptr ? int(ysrc[i]) : i; };

    float acc[NACC][P][CH];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int q = 0; q < P; ++q)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[a][q][c] = 0.0f;

    // NB row buffers in registers, NB - 1 rows in flight.  Only the loads BEFORE the loop can meet row 0 with a negative offset
    // (clamped dword by dword, load_span<.., true>); the loop's own loads are rows >= 1, where a negative lane offset lands in
    // the previous row: one merged, branch-free load sequence — a second load path inside the loop made the compiler wait for
    // the prefetch it had just issued.
    auto load_row = [&](uint32_t (&t)[NRAW], int row) __attribute__((always_inline)) {
        if constexpr (XCH) {
            const uint32_t off = uint32_t(row * p.src_stride + lane_off);
#pragma unroll
            for (int k = 0; k < X::NOWN; ++k) t[k] = __builtin_amdgcn_raw_buffer_load_b32(src_rs, off + 4 * k, 0, 0);
            t[X::NOWN] = __builtin_amdgcn_raw_buffer_load_b32(src_rs, uint32_t(row * p.src_stride + halo_off), 0, 0);
        } else if constexpr (SHU) {
            load_span<CO, false>(src_rs, row * p.src_stride + lane_off, t);
            t[CO::NW] = load_halo(row);
        } else {
            load_span<C, false>(src_rs, row * p.src_stride + lane_off, t);
        }
    };
    uint32_t raw[NB][NRAW];
#pragma unroll
    for (int k = 0; k < NB - 1; ++k) {
        const int row = src_of(min(r0 + k, r1));
        if constexpr (XCH) { load_row(raw[k], row); }
        else if constexpr (SHU) {
            if (careful && row <= 0) {
                load_span<CO, true>(src_rs, row * p.src_stride + lane_off, raw[k]);
                raw[k][CO::NW] = load_halo(row);
            } else load_row(raw[k], row);
        } else {
            const int32_t off = row * p.src_stride + lane_off;
            if (careful && row <= 0) load_span<C, true>(src_rs, off, raw[k]);
            else load_span<C, false>(src_rs, off, raw[k]);
        }
    }

    // The lanes that hold the image's first / last columns keep their raw dwords of every row for resize_edges_kernel: into a
    // wave-private LDS buffer of 64 rows per side, flushed to the global stash with one coalesced store sequence per 64 rows.
    // (Stored straight to memory, one lane and four store instructions per row, the march ran 10-17 % longer: on gfx9 stores
    // count in vmcnt like loads, so every counted wait for a row's pixels also waited for the stash stores issued before it.)
    extern __shared__ __attribute__((aligned(16))) uint32_t stash_lds[];   // [waves][2 sides][64 rows][NRAW]
    const int g_last = (p.dst_w - 1) / P;
    const int st_side_lane = p.stash == nullptr ? -1 : (grp0 + lane == 0) ? 0 : (grp0 + lane == g_last) ? 1 : -1;
    const bool st_has0 = __any(st_side_lane == 0), st_has1 = __any(st_side_lane == 1);
    const bool st_wave = st_has0 || st_has1;
    constexpr int NRP = (NRAW + 3) / 4 * 4;   // LDS pitch of a stash row: 16-byte stores
    uint32_t *st_buf = stash_lds + wave * (2 * kWave * NRP);
    const unsigned st_stride = unsigned(p.stash_words) * 4u, st_bytes = st_stride * unsigned(p.src_h);
    const __amdgpu_buffer_rsrc_t stash_rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<uint8_t *>(p.stash) + (size_t)(p.stash_img0 + img) * 2 * st_bytes, 0, p.stash ? 2 * st_bytes : 0, 0x00020000);
    auto stash_row = [&](uint32_t (&cur)[NRAW], int row) __attribute__((always_inline)) {
        const int slot = (row - r0) & (kWave - 1);
        if (st_side_lane >= 0) {
            u32x4 *d4 = reinterpret_cast<u32x4 *>(st_buf + (st_side_lane * kWave + slot) * NRP);
#pragma unroll
            for (int k4 = 0; k4 < NRP / 4; ++k4) {
                u32x4 v = {cur[4 * k4], 4 * k4 + 1 < NRAW ? cur[4 * k4 + 1] : 0u, 4 * k4 + 2 < NRAW ? cur[4 * k4 + 2] : 0u, 4 * k4 + 3 < NRAW ? cur[4 * k4 + 3] : 0u};
                d4[k4] = v;
            }
        }
        if (slot == kWave - 1 || row == r1) {  // wave-uniform: flush rows [row - slot, row]
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                if (side == 0 ? st_has0 : st_has1) {
                    if (lane <= slot) {
                        uint32_t t[NRAW];
#pragma unroll
                        for (int k = 0; k < NRAW; ++k) t[k] = st_buf[(side * kWave + lane) * NRP + k];
                        store_dwords<NRAW>(stash_rs, uint32_t(side) * st_bytes + uint32_t(row - slot + lane) * st_stride, t);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    };
    uint32_t obuf[PH ? kPhaseRows : 1][ODW];
    int nbuf = 0, first_y = 0;   // wave-uniform
    [[maybe_unused]] unsigned phase = PH ? unsigned(__builtin_amdgcn_s_memrealtime() >> p.phase_shift) : 0u;
    auto flush_outputs = [&]() __attribute__((always_inline)) {
        if constexpr (PH) {
#pragma unroll
            for (int q = 0; q < kPhaseRows; ++q) {
                if (q < nbuf) {
                    const uint32_t doff = uint32_t((first_y + q) * p.dst_stride + x0 * C::BPP);
                    if (x0 + P <= p.dst_w) store_dwords<ODW, kStoreNT>(dst_rs, doff, obuf[q]);
                    else {
#pragma unroll
                        for (int o = 0; o < P; ++o)
                            if (x0 + o < p.dst_w) store_dwords<ODW / P, kStoreNT>(dst_rs, doff + o * C::BPP, &obuf[q][o * (ODW / P)]);
                    }
                }
            }
            nbuf = 0;
        }
    };
    auto process_row = [&](uint32_t (&cur)[NRAW], int row) __attribute__((always_inline)) {
        if constexpr (!XCH && !SHU) {
            if (st_wave && row <= r1) stash_row(cur, row);  // wave-uniform
        }
        int rec[2 * NACC];
#pragma unroll
        for (int t = 0; t < 2 * NACC; ++t) rec[t] = yrec[(size_t)row * (2 * NACC) + t];
        // phased stores: the chip-wide clock is read here, with the record, so that its scalar-memory latency passes under the horizontal pass
        [[maybe_unused]] unsigned long long realtime = 0;
        if constexpr (PH) realtime = __builtin_amdgcn_s_memrealtime();
        // ---- horizontal pass (src/resize.cc:110-117): every window pixel converted once, used by all outputs it is a tap of ----
        float h[P][CH];
#if PICHA_RUN_ABLATE == 1
        {
            uint32_t x = 0;
#pragma unroll
            for (int k = 0; k < NRAW; ++k) x ^= cur[k];
#pragma unroll
            for (int o = 0; o < P; ++o)
#pragma unroll
                for (int c = 0; c < CH; ++c) h[o][c] = __uint_as_float(x & 0x3fffffffu);
        }
#else
        uint32_t win[SHU ? CO::NWIN : XCH ? C::NW : (C::BROT ? C::NWIN : 1)];
        [[maybe_unused]] f32x4 *xl = reinterpret_cast<f32x4 *>(stash_lds) + wave * (S::NFC * S::LANES);   // (shared unpack: the wave's exchange slots)
        [[maybe_unused]] float uown[SHU ? S::NFC * 4 > S::OWN * CH ? S::NFC * 4 : S::OWN * CH : 1], unext[SHU ? S::NFC * 4 : 1], unext2[SHU && S::HR2 > 0 ? S::NFC2 * 4 : 1];
        if constexpr (SHU) {
            rotate_span<CO>(cur, uint32_t(src_of(min(row, r1)) * p.src_stride + lane_off) & 3u, win);
#pragma unroll
            for (int q = 0; q < S::OWN; ++q) unpack_span_px<CO, CH, S16, FMA>(win, q, &uown[q * CH]);
#pragma unroll
            for (int k = S::OWN * CH; k < S::NFC * 4; ++k) uown[k] = 0.0f;   // (the last chunk's padding)
#pragma unroll
            for (int c = 0; c < S::NFC; ++c) {
                f32x4 v = {uown[4 * c], uown[4 * c + 1], uown[4 * c + 2], uown[4 * c + 3]};
                xl[c * S::LANES + lane] = v;
            }
            if constexpr (S::HALO_PX) {   // the lane's halo PIXEL k (k < HR): pixel k % OWN of virtual lane 64 + k / OWN
                const int hk = min(lane, S::HR - 1);
                const int vl = hk >= S::OWN ? 1 : 0, hp = hk - vl * S::OWN;
                float uh[CH];
                const uint32_t hw[1] = {cur[CO::NW]};
                unpack_span_px<EdgeCfg<CH, S16, 1>, CH, S16, FMA>(hw, 0, uh);
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int f = hp * CH + c;
                    reinterpret_cast<float *>(xl)[((f >> 2) * S::LANES + kWave + vl) * 4 + (f & 3)] = uh[c];
                }
            } else {
                const float uh = FMA ? float(cur[CO::NW]) : unpack_sample<S16>(cur[CO::NW]);
                const int hk = min(lane, S::HR * CH - 1);   // sample hk of the samples behind the run: float hk % (OWN * CH) of virtual lane 64 + hk / (OWN * CH)
                const int vl = hk >= S::OWN * CH ? 1 : 0, f = hk - vl * S::OWN * CH;
                reinterpret_cast<float *>(xl)[((f >> 2) * S::LANES + kWave + vl) * 4 + (f & 3)] = uh;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if constexpr (S::HR2 == 0) {   // (windows over two neighbours read theirs where the tap loop reaches them: 130 live floats otherwise)
#pragma unroll
                for (int c = 0; c < S::NFC; ++c) {
                    const f32x4 v = xl[c * S::LANES + lane + 1];
                    unext[4 * c] = v[0]; unext[4 * c + 1] = v[1]; unext[4 * c + 2] = v[2]; unext[4 * c + 3] = v[3];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else
        if constexpr (C::BROT) rotate_span<C>(cur, uint32_t(src_of(min(row, r1)) * p.src_stride + lane_off) & 3u, win);
        if constexpr (XCH && PICHA_RUN_ABLATE == 3) {
#pragma unroll
            for (int k = 0; k < C::NW; ++k) win[k] = cur[k % X::NOWN];
        } else if constexpr (XCH) {
            // the wave's halo: lane k wrote dword k; every lane reads all of it back (same address in all lanes: a broadcast), lanes 0 /
            // 63 keep it where the wave shift has no source lane (a DPP move leaves its destination alone there)
            uint32_t *hl = halo_lds + wave * kWave;
            hl[lane] = cur[X::NOWN];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            uint32_t hb[X::NH > 0 ? X::NH : 1];
#pragma unroll
            for (int k = 0; k < X::NH; ++k) hb[k] = hl[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < X::NHL; ++k)
                win[k] = __builtin_amdgcn_update_dpp(hb[k], cur[X::NOWN - X::NHL + k], kDppWaveShr1, 0xf, 0xf, false);
#pragma unroll
            for (int k = 0; k < X::NOWN; ++k) win[X::NHL + k] = cur[k];
#pragma unroll
            for (int k = 0; k < X::NHR; ++k)
                win[X::NHL + X::NOWN + k] = __builtin_amdgcn_update_dpp(hb[X::NHL + k], cur[k], kDppWaveShl1, 0xf, 0xf, false);
        }
#pragma unroll
        for (int q = 0; q < C::SPAN; ++q) {
            float u[CH];
#if PICHA_RUN_ABLATE == 5   // measurement only (wrong results): the window pixels outside the lane's own (P / Q) * PX cost no unpack — the upper bound of what handing
            // running sums (or unpacked pixels) between neighbouring lanes could save
            constexpr int AOWN = (P / Q) * PX, AOFF = (C::SPAN - AOWN) / 2;
            const int qa = q < AOFF ? AOFF : q >= AOFF + AOWN ? AOFF + AOWN - 1 : q;
            if constexpr (C::BROT || XCH) unpack_span_px<C, CH, S16, FMA>(win, qa, u);
            else unpack_span_px<C, CH, S16, FMA>(cur, qa, u);
#else
            if constexpr (SHU && S::HR2 > 0) {   // two neighbours: each one's floats are read when the tap loop reaches them (the compiler may not hoist the reads: "memory")
                if (q == S::OWN) {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int c = 0; c < S::NFC; ++c) {
                        const f32x4 v = xl[c * S::LANES + lane + 1];
                        unext[4 * c] = v[0]; unext[4 * c + 1] = v[1]; unext[4 * c + 2] = v[2]; unext[4 * c + 3] = v[3];
                    }
                    asm volatile("" ::: "memory");
                }
                if (q == 2 * S::OWN) {
#pragma unroll
                    for (int c = 0; c < S::NFC2; ++c) {
                        const f32x4 v = xl[c * S::LANES + lane + 2];
                        unext2[4 * c] = v[0]; unext2[4 * c + 1] = v[1]; unext2[4 * c + 2] = v[2]; unext2[4 * c + 3] = v[3];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the next row's writes stay behind these reads)
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if constexpr (SHU) {
#pragma unroll
                for (int c = 0; c < CH; ++c) u[c] = q < S::OWN ? uown[q * CH + c] : q < 2 * S::OWN ? unext[(q - S::OWN) * CH + c] : unext2[(q - 2 * S::OWN < 0 ? 0 : q - 2 * S::OWN) * CH + c];
            }
            else if constexpr (C::BROT || XCH) unpack_span_px<C, CH, S16, FMA>(win, q, u);
            else unpack_span_px<C, CH, S16, FMA>(cur, q, u);
#endif
#pragma unroll
            for (int o = 0; o < P; ++o) {
                const int j = q - run_off(Q, PX, o, RC);  // tap index of window pixel q for output o
                if (j >= 0 && j < TX) {
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        if (j == 0) h[o][c] = wx[o % NWX][0] * u[c];
                        else if constexpr (FMA) h[o][c] = __builtin_fmaf(wx[o % NWX][j], u[c], h[o][c]);
                        else h[o][c] = h[o][c] + wx[o % NWX][j] * u[c];
                    }
                }
            }
        }
#endif
        // ---- fold the row into the pending output rows (src/resize.cc:125-129) ----
#if PICHA_RUN_ABLATE == 1
#pragma unroll
        for (int o = 0; o < P; ++o)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[0][o][c] = h[o][c];
#else
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            float w;
            asm volatile("v_mov_b32 %0, %1" : "=v"(w) : "s"(rec[a]));
#pragma unroll
            for (int o = 0; o < P; ++o)
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    if constexpr (FMA) acc[a][o][c] = __builtin_fmaf(w, h[o][c], acc[a][o][c]);
                    else acc[a][o][c] = acc[a][o][c] + w * h[o][c];
                }
        }
#endif
        // ---- output rows whose window ends with this source row: pack, store (whole dwords: host-checked alignment), clear ----
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            const int y = rec[NACC + a];
            if (y >= 0) {  // wave-uniform
                if constexpr (PH) {
                    if (y >= y0 && y < y1) {   // pack now, store with the next burst (rows of a chunk complete in ascending order)
                        uint32_t pk[ODW];
#pragma unroll
                        for (int o = 0; o < P; ++o) {
                            uint32_t t2[2];
                            pack_px_words<CH, S16, FMA>(acc[a][o], t2);
                            pk[o * (ODW / P)] = t2[0];
                            if constexpr (ODW / P == 2) pk[o * 2 + 1] = t2[1];
                        }
                        // (several rows can complete with one source row — at the bottom edge every clipped window ends with the last
                        // row — so the buffer may fill up here; rows held are consecutive by construction, checked all the same)
                        if (nbuf == kPhaseRows || (nbuf > 0 && y != first_y + nbuf)) flush_outputs();
                        if (nbuf == 0) first_y = y;
#pragma unroll
                        for (int q = 0; q < kPhaseRows; ++q)
                            if (nbuf == q) {
#pragma unroll
                                for (int k = 0; k < ODW; ++k) obuf[q][k] = pk[k];
                            }
                        ++nbuf;
                    }
                } else {
#if PICHA_RUN_ABLATE == 4   // no output stores (one row per chunk keeps the arithmetic alive)
                    if (y == y0)
#else
                    if (y >= y0 && y < y1)
#endif
                    {
                        bool side_lane = false;
                        if constexpr (EFUSE) side_lane = p.edge_fused && (x0 < p.cx0 || x0 + P > p.cx1);
                        if (side_lane) {   // (edge blocks in the grid: the clipped columns are theirs, in no order with this store)
#pragma unroll
                            for (int q = 0; q < P; ++q)
                                if (x0 + q >= p.cx0 && x0 + q < p.cx1) {
                                    uint32_t pw[2];
                                    pack_px_words<CH, S16, FMA>(acc[a][q], pw);
                                    store_px_words<CH, S16>(dst_rs, uint32_t(y * p.dst_stride + (x0 + q) * C::BPP), pw);
                                }
                        } else
                        store_lane_outputs<C, CH, S16, P, FMA>(acc[PICHA_RUN_ABLATE == 1 ? 0 : a], dst_rs, uint32_t(y * p.dst_stride + x0 * C::BPP), x0, p.dst_w);
                    }
                }
#pragma unroll
                for (int o = 0; o < P; ++o)
#pragma unroll
                    for (int c = 0; c < CH; ++c) acc[a][o][c] = 0.0f;
            }
        }
        if constexpr (PH) {
            const unsigned ph = unsigned(realtime >> p.phase_shift);
            if (nbuf == kPhaseRows || (nbuf > 0 && ph != phase)) flush_outputs();
            phase = ph;
        }
    };

    // rows past r1 inside the last group of NB are ghosts: they re-read row r1 and use the records of rows > r1, in which every
    // output of this chunk has weight 0 and none of them completes
    for (int r = r0; r <= r1; r += NB) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
#if PICHA_RUN_ABLATE != 2
            load_row(raw[(k + NB - 1) % NB], src_of(max(XCH ? 0 : 1, min(r + k + NB - 1, r1))));
#endif
            process_row(raw[k], r + k);
        }
    }
    flush_outputs();
    };  // run_item
    run_work_items(p.work, [&](int &bx, int &y0, int &y1, int &img) __attribute__((always_inline)) {
        bx = blockIdx.x; y0 = blockIdx.y * p.rows_per_chunk; y1 = min(p.dst_h, y0 + p.rows_per_chunk); img = blockIdx.z;
    }, run_item);
    PICHA_PROBE_END
}

// register budget of resize_ring_kernel in waves per SIMD (r05): lanczos 2x on rgba compiles to 131 VGPRs — three waves per SIMD, and an odd number of co-resident waves
// issues at 2.74 cycles per instruction instead of 2.05 (profiles/r05_ubench_valu_issue.txt); held to 128 (five spilled dwords) it runs 3.3 % faster, fma 2.1 %
// (profiles/r05_ab_reg_budgets.txt).  Every other shape keeps what the compiler picks (the 4x upscales would spill 80-100 dwords under 128).
constexpr int ring_min_waves(int ch, bool s16, int q, int px, int tx, int p, int k, int srcp) {
#ifdef PICHA_RING_WAVES
    return PICHA_RING_WAVES;   // measurement builds
#else
    return (ch == 4 && !s16 && q == 1 && px == 2 && tx == 9 && p == 2 && k == 8 && srcp < 0) ? 4 : 1;
#endif
}

// shapes of resize_ring_kernel that can produce the clipped edge columns themselves (EIN): upscales (more outputs than source pixels per period) whose
// per-output weight sets are at most 32 registers
__host__ __device__ constexpr bool ring_edges_inside(int q, int px, int tx, int p) { return q > px && p * tx <= 32; }

// compile-time helpers of resize_ring_kernel: call f(integral_constant<i>) for the run-time i in [0, N) / for every i in [0, N)
template <int N, int I = 0, typename Fn> __device__ __forceinline__ void switch_slot(int i, Fn &&f) {
    if constexpr (I + 1 == N) { f(std::integral_constant<int, I>{}); }
    else { if (i == I) f(std::integral_constant<int, I>{}); else switch_slot<N, I + 1>(i, f); }
}
template <int N, int I = 0, typename Fn> __device__ __forceinline__ void unroll_steps(int r, Fn &&f) {
    if constexpr (I < N) { f(r, std::integral_constant<int, I>{}); unroll_steps<N, I + 1>(r, f); }
}

// ---------------------------------------------------------------------------------------------
// Ring-in-registers kernel (r03): the same horizontal scheme — a periodic tap table (Q outputs per PX source pixels), a lane
// resamples a run of P outputs from one window with wave-uniform weights — and a vertical part that IS the reference's ring
// (src/resize.cc:52-64,121-132): the last K resampled rows stay in REGISTERS, slot = row mod K, written at compile-time
// positions of a loop unrolled over K rows, and every output row is summed from them as soon as the last row the reference has
// resampled by then (its `need`, src/resize.cc:104) is in — taps in the reference's order, slot of tap j = (left + j) mod K.
//   * integer UPSCALES (Q = factor, PX = 1, K = 4: picha's upscale windows are at most three rows, four for the 4-tap filters):
//     no LDS ring (the span kernel of 5.2 reads count x P cells per output row and carries per-tap branches), no per-row records
//     over NACC accumulators of which half hold zeros (the accumulator form of 5.3c);
//   * RING-ALIASED downscales (lanczos at 2x / 4x: windows of 9 / 17 rows on the reference's ring of K = 8 / 16): with K equal
//     to the reference's ring the aliasing — the first tap reads the slot that already holds the window's last row — happens by
//     itself, as in the LDS-ring kernel, without its LDS traffic and per-tap slot loads.
// Edge columns as for the run kernel: stash + resize_edges_kernel.
// ---------------------------------------------------------------------------------------------
//   * SRCP >= 0 (r04): a colour convert in front (picha.colorConvert, then picha.resize, as ONE kernel): the windows are loaded in the source
//     format SRCP and every window pixel is converted and quantised in registers (convert_span_px) before it meets its weights; the clipped
//     columns are recomputed from the image by resize_edges_kernel<..., SRCP> (no stash).
template <int CH, bool S16, int Q, int PX, int TX, int P, int K, int TYM, bool FMA, int RC, int SRCP = -1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ring_min_waves(CH, S16, Q, PX, TX, P, K, SRCP)))) void resize_ring_kernel(RunParams p) {
    using C = RunCfg<CH, S16, Q, PX, TX, P, RC>;
    constexpr int SCH = SRCP < 0 ? CH : pixel_channels(SRCP < 0 ? 0 : SRCP);
    constexpr bool SS16 = SRCP < 0 ? S16 : pixel_is16(SRCP < 0 ? 0 : SRCP);
    using CS = RunCfg<SCH, SS16, Q, PX, TX, P, RC>;   // the windows as they are loaded
    constexpr int F = Q;
    static_assert(P % F == 0, "a lane owns whole periods");
    constexpr int NRAW = CS::NW;
    constexpr int NB = 4;  // row buffers; K is a multiple of it: one unrolled group of K rows
    PICHA_PROBE_BEGIN
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bw = __builtin_amdgcn_readfirstlane(blockDim.x >> 6);
    const int bx = blockIdx.x, img = blockIdx.z;
    const int y0 = blockIdx.y * p.rows_per_chunk, y1 = min(p.dst_h, y0 + p.rows_per_chunk);
    const int grp0 = (bx * bw + wave) * kWave;
    if (grp0 * P >= p.dst_w || y0 >= y1) return;
    const int x0 = (grp0 + lane) * P;

    const uint8_t *sbase = uniform_ptr(p.src_tab ? static_cast<const uint8_t *>(p.src_tab[img]) : p.src + img * p.src_pitch);
    uint8_t *dbase = uniform_ptr(p.dst_tab ? static_cast<uint8_t *>(p.dst_tab[img]) : p.dst + img * p.dst_pitch);
    const int sadj = CS::BROT ? int(reinterpret_cast<uintptr_t>(sbase) & 3u) : 0;
    const __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(sbase) - sadj, 0, CS::WORDS ? p.src_bytes : ((p.src_bytes + sadj + 3u) & ~3u), 0x00020000);
    const __amdgpu_buffer_rsrc_t dst_rs = __builtin_amdgcn_make_buffer_rsrc(dbase, 0, p.dst_bytes, 0x00020000);

    const_i32_ptr xc = (const_i32_ptr)(p.x_canon);
    int32_t wbits[F * TX];
#pragma unroll
    for (int j = 0; j < F * TX; ++j) wbits[j] = xc[j];
    // EIN (r05): EDGE COLUMNS INSIDE — short runs of short windows (integer upscales) keep one weight set per OUTPUT of the lane instead of one per phase; with
    // p.edge_in the lanes that hold clipped columns load those columns' own weights (src/resize.cc:19-50: fewer taps, renormalised), zero-padded into the
    // canonical window, and the march produces every column itself: no stash, no resize_edges_kernel, which cost the upscale cells 5-15 % of their time
    // (profiles/r05_ab_edge_cost.txt).  A padded tap multiplies whatever lies beside the image by 0.0f and stands at the START or END of the ascending sum, so
    // the sum is the reference's (+-0 aside, which packs to the same sample).
    constexpr bool EIN = ring_edges_inside(Q, PX, TX, P);
    constexpr int NWX = EIN ? P : F;
    float wx[NWX][TX];
#pragma unroll
    for (int q = 0; q < NWX; ++q)
#pragma unroll
        for (int j = 0; j < TX; ++j) asm("v_mov_b32 %0, %1" : "=v"(wx[q][j]) : "s"(wbits[(q % F) * TX + j]));
    if constexpr (EIN) {
        if (p.edge_in) {
#pragma unroll
            for (int o = 0; o < P; ++o) {
                const int x = x0 + o;
                if (x < p.dst_w && (x < p.cx0 || x >= p.cx1)) {   // (lane-varying: only the lanes at the image's sides)
                    const int first = p.x_left[x] - (p.x_base + (x0 / Q) * PX + run_off(Q, PX, o, RC));   // host-checked: 0 <= first, first + count <= TX
                    const int count = p.x_count[x];
#pragma unroll
                    for (int j = 0; j < TX; ++j) {
                        const int t = j - first;
                        float w = (t >= 0 && t < count) ? p.x_weight[(size_t)x * p.x_wtaps + min(max(t, 0), p.x_wtaps - 1)] : 0.0f;
                        if constexpr (FMA) w = w * Sample<S16>::kInv;
                        else if constexpr (exact_unpack_scaled(CH, S16)) w = w * kWeightScale;
                        wx[o][j] = w;
                    }
                }
            }
        }
    }
    static_assert(K % NB == 0 && TYM <= K + 1, "ring");
    const int32_t lane_off = (p.x_base + (grp0 + lane) * ((P / F) * PX)) * CS::BPP + sadj;
    const bool careful = __any(lane_off < 0);

    const_i32_ptr yrange = (const_i32_ptr)(p.yrow_range);
    const_i32_ptr yneed = (const_i32_ptr)(p.y_info);   // int4 per output row: {., need, ., .}
    const_i32_ptr yleft = (const_i32_ptr)(p.y_left), ycount = (const_i32_ptr)(p.y_count), ywgt = (const_i32_ptr)(p.y_weight);
    const int r0 = yrange[2 * y0], r1 = yrange[2 * (y1 - 1) + 1];

    // stash of the edge lanes' raw rows (see resize_run_kernel)
    extern __shared__ __attribute__((aligned(16))) uint32_t stash_lds[];
    constexpr int NRP = (NRAW + 3) / 4 * 4;
    const int g_last = (p.dst_w - 1) / P;
    const int st_side_lane = p.stash == nullptr ? -1 : (grp0 + lane == 0) ? 0 : (grp0 + lane == g_last) ? 1 : -1;
    const bool st_has0 = __any(st_side_lane == 0), st_has1 = __any(st_side_lane == 1);
    const bool st_wave = st_has0 || st_has1;
    uint32_t *st_buf = stash_lds + wave * (2 * kWave * NRP);
    const unsigned st_stride = unsigned(p.stash_words) * 4u, st_bytes = st_stride * unsigned(p.src_h);
    const __amdgpu_buffer_rsrc_t stash_rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<uint8_t *>(p.stash) + (size_t)(p.stash_img0 + img) * 2 * st_bytes, 0, p.stash ? 2 * st_bytes : 0, 0x00020000);
    auto stash_row = [&](uint32_t (&cur)[NRAW], int row) __attribute__((always_inline)) {
        const int slot = (row - r0) & (kWave - 1);
        if (st_side_lane >= 0) {
#pragma unroll
            for (int k = 0; k < NRAW; ++k) st_buf[(st_side_lane * kWave + slot) * NRP + k] = cur[k];
        }
        if (slot == kWave - 1 || row == r1) {  // wave-uniform: flush rows [row - slot, row]
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                if (side == 0 ? st_has0 : st_has1) {
                    if (lane <= slot) {
                        uint32_t t[NRAW];
#pragma unroll
                        for (int k = 0; k < NRAW; ++k) t[k] = st_buf[(side * kWave + lane) * NRP + k];
                        store_dwords<NRAW>(stash_rs, uint32_t(side) * st_bytes + uint32_t(row - slot + lane) * st_stride, t);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    };

    uint32_t raw[NB][NRAW];
#pragma unroll
    for (int k = 0; k < NB - 1; ++k) {
        const int row = min(r0 + k, r1);
        const int32_t off = row * p.src_stride + lane_off;
        if (careful && row <= 0) load_span<CS, true>(src_rs, off, raw[k]);
        else load_span<CS, false>(src_rs, off, raw[k]);
    }

    float ring[K][P][CH];   // resampled rows, slot = (row - r0) mod K
#pragma unroll
    for (int a = 0; a < K; ++a)
#pragma unroll
        for (int q = 0; q < P; ++q)
#pragma unroll
            for (int c = 0; c < CH; ++c) ring[a][q][c] = 0.0f;

    // the next output row to emit and its window, fetched ahead through the scalar cache
    int y = y0;
    int yl = yleft[y0], yc = ycount[y0], yn_row = yneed[4 * y0 + 1];
    int wb[TYM];
#pragma unroll
    for (int j = 0; j < TYM; ++j) wb[j] = ywgt[(size_t)y0 * p.y_taps + min(j, p.y_taps - 1)];

    auto emit = [&](auto base_c) __attribute__((always_inline)) {
        constexpr int B = decltype(base_c)::value;   // ring slot of the window's first row
        float o[P][CH];
#pragma unroll
        for (int j = 0; j < TYM; ++j) {
            if (j < yc) {   // wave-uniform
                float w;
                asm volatile("v_mov_b32 %0, %1" : "=v"(w) : "s"(wb[j]));
#pragma unroll
                for (int q = 0; q < P; ++q)
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const float v = ring[(B + j) % K][q][c];
                        if (j == 0) o[q][c] = w * v;
                        else if constexpr (FMA) o[q][c] = __builtin_fmaf(w, v, o[q][c]);
                        else o[q][c] = o[q][c] + w * v;
                    }
            }
        }
        store_lane_outputs<C, CH, S16, P, FMA>(o, dst_rs, uint32_t(y * p.dst_stride + x0 * C::BPP), x0, p.dst_w);
    };

    // wave-uniform dispatch on the slot of the window's first row: K copies of the sum, each with compile-time ring indices
    auto emit_at = [&](int b0, auto self_c) __attribute__((always_inline)) {
        (void)self_c;
        switch_slot<K>(b0, [&](auto bc) __attribute__((always_inline)) { emit(bc); });
    };
    auto process_row = [&](uint32_t (&cur)[NRAW], int row, auto slot_c) __attribute__((always_inline)) {
        constexpr int SL = decltype(slot_c)::value;   // == (row - r0) mod K
        if (st_wave && row <= r1) stash_row(cur, row);
        // ---- horizontal pass (src/resize.cc:110-117) ----
        uint32_t win[CS::BROT ? CS::NWIN : 1];
        if constexpr (CS::BROT) rotate_span<CS>(cur, uint32_t(min(row, r1) * p.src_stride + lane_off) & 3u, win);
        float h[P][CH];
#pragma unroll
        for (int q = 0; q < C::SPAN; ++q) {
            float u[CH];
            if constexpr (SRCP >= 0) convert_span_px<CS, SCH, SS16, CH, S16, FMA>(CS::BROT ? win : cur, q, u, p.cv_r, p.cv_g, p.cv_b);
            else if constexpr (C::BROT) unpack_span_px<C, CH, S16, FMA>(win, q, u);
            else unpack_span_px<C, CH, S16, FMA>(cur, q, u);
#pragma unroll
            for (int o = 0; o < P; ++o) {
                const int j = q - run_off(Q, PX, o, RC);
                if (j >= 0 && j < TX) {
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        if (j == 0) h[o][c] = wx[o % NWX][0] * u[c];
                        else if constexpr (FMA) h[o][c] = __builtin_fmaf(wx[o % NWX][j], u[c], h[o][c]);
                        else h[o][c] = h[o][c] + wx[o % NWX][j] * u[c];
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < P; ++q)
#pragma unroll
            for (int c = 0; c < CH; ++c) ring[SL][q][c] = h[q][c];
        // ---- every output row the reference produces once this row is resampled (src/resize.cc:104,121-132) ----
        while (y < y1 && yn_row <= row) {   // wave-uniform
            const int b0 = (yl - r0) % K;   // ring slot of the window's first row (yl >= r0)
            emit_at(b0, std::integral_constant<int, 0>{});
            ++y;
            const int yn = min(y, p.dst_h - 1);
            yl = yleft[yn]; yc = ycount[yn]; yn_row = yneed[4 * yn + 1];
#pragma unroll
            for (int j = 0; j < TYM; ++j) wb[j] = ywgt[(size_t)yn * p.y_taps + min(j, p.y_taps - 1)];
        }
    };
    // The unrolled group starts at r0, so position k of a group is ring slot (row - r0) mod K
    auto step = [&](int r, auto k_c) __attribute__((always_inline)) {
        constexpr int KK = decltype(k_c)::value;
        load_span<CS, false>(src_rs, max(1, min(r + KK + NB - 1, r1)) * p.src_stride + lane_off, raw[(KK + NB - 1) % NB]);
        if (r + KK <= r1) process_row(raw[KK % NB], r + KK, k_c);
    };
    for (int r = r0; r <= r1; r += K) unroll_steps<K>(r, step);
    PICHA_PROBE_END
}

template <int CH, bool S16, int Q, int PX, int TX, int P, int RC, int K, int TYM, int SRCP = -1>
hipError_t launch_ring_t(const RunParams &p, hipStream_t s, const char **name) {
    const int bw = p.block_waves >= 1 && p.block_waves <= 4 ? p.block_waves : 4;
    const size_t lds = p.stash ? sizeof(uint32_t) * bw * 2 * kWave * ((RunCfg<CH, S16, Q, PX, TX, P, RC>::NW + 3) / 4 * 4) : 0;
    if (SRCP >= 0 && p.stash) return hipErrorInvalidValue;   // a convert in front: the edge columns come from the image
    if (p.edge_in && !ring_edges_inside(Q, PX, TX, P)) return hipErrorInvalidValue;
    dim3 grid((p.x_groups + 64 * bw - 1) / (64 * bw), p.n_chunks, p.n_images);
    if (p.fma) resize_ring_kernel<CH, S16, Q, PX, TX, P, K, TYM, true, RC, SRCP><<<grid, 64 * bw, lds, s>>>(p);
    else resize_ring_kernel<CH, S16, Q, PX, TX, P, K, TYM, false, RC, SRCP><<<grid, 64 * bw, lds, s>>>(p);
    char srcs[24] = "";
    if (SRCP >= 0) snprintf(srcs, sizeof srcs, "src ch%d%s,", pixel_channels(SRCP < 0 ? 0 : SRCP), pixel_is16(SRCP < 0 ? 0 : SRCP) ? "u16" : "u8");
    if (Q > PX) snprintf(g_kernel_name, sizeof g_kernel_name, "resize_up<%sch%d,%s,F%d,TX%d,P%d,%s,bw%d%s%s>", srcs, CH, S16 ? "u16" : "u8", Q, TX, P, p.fma ? "fma" : "exact", bw, RC ? ",ceil" : "",
                        p.edge_in && ring_edges_inside(Q, PX, TX, P) ? ",edges" : "");
    else snprintf(g_kernel_name, sizeof g_kernel_name, "resize_ring<%sch%d,%s,Q%d,PX%d,TX%d,P%d,K%d,%s,bw%d%s>", srcs, CH, S16 ? "u16" : "u8", Q, PX, TX, P, K, p.fma ? "fma" : "exact", bw, RC ? ",ceil" : "");
    if (name) *name = g_kernel_name;
    return hipGetLastError();
}

template <int CH, bool S16, int Q, int PX, int TX, int P, int NACC, int RC, int XHL = -1, int WS = 0, int PH = 0>
hipError_t launch_run_t(const RunParams &p, hipStream_t s, const char **name) {
    const void *k = p.fma ? reinterpret_cast<const void *>(resize_run_kernel<CH, S16, Q, PX, TX, P, NACC, true, RC, XHL, WS, PH>)
                          : reinterpret_cast<const void *>(resize_run_kernel<CH, S16, Q, PX, TX, P, NACC, false, RC, XHL, WS, PH>);
    const int bw = p.block_waves >= 1 && p.block_waves <= 4 ? p.block_waves : 4;
    using SH = RunShu<CH, S16, Q, PX, TX, P, RC>;
    if (XHL == -2 && p.stash) return hipErrorInvalidValue;   // shared unpack: the clipped columns come from the image
    if (p.edge_in && !run_edges_inside(XHL, WS, TX, P)) return hipErrorInvalidValue;
    const size_t lds = XHL == -2 ? sizeof(float) * 4 * bw * SH::NFC * SH::LANES   // the exchange slots
                     : p.stash ? sizeof(uint32_t) * bw * 2 * kWave * ((RunCfg<CH, S16, Q, PX, TX, P, RC>::NW + 3) / 4 * 4) : 0;   // the stash staging of the edge lanes
    if (p.resident_query) {
        int per_cu = 0, dev = 0, cus = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 64 * bw, lds);
        if (e == hipSuccess) e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        *p.resident_query = e == hipSuccess ? per_cu * cus : 0;
        return e;
    }
    dim3 grid((p.x_groups + 64 * bw - 1) / (64 * bw), p.n_chunks, p.n_images);
    if (p.work.items) grid = dim3(p.work.queue ? (unsigned)p.n_chunks : 8u * (unsigned)p.work.items_per_xcd, 1, 1);
    RunParams q = p;
    if (p.edge_fused) {   // two edge blocks per (row chunk, image) behind the march's; they work in the workgroup's exchange slots (kEdgeFloats floats)
        if (!run_edges_fused(CH, S16, Q, PX, TX, P, XHL) || p.work.items || !p.edge_image || lds < sizeof(float) * kEdgeFloats) return hipErrorInvalidValue;
        q.edge_bx0 = (int)grid.x;
        grid.x += 2;
    }
    if (p.fma) resize_run_kernel<CH, S16, Q, PX, TX, P, NACC, true, RC, XHL, WS, PH><<<grid, 64 * bw, lds, s>>>(q);
    else resize_run_kernel<CH, S16, Q, PX, TX, P, NACC, false, RC, XHL, WS, PH><<<grid, 64 * bw, lds, s>>>(q);
    char form[40] = "";
    if (XHL >= 0) snprintf(form, sizeof form, ",xch%d", XHL);
    if (XHL == -2) snprintf(form, sizeof form, ",shu");
    if (WS > 0) snprintf(form + strlen(form), sizeof form - strlen(form), ",ws%d", WS);
    if (PH > 0) snprintf(form + strlen(form), sizeof form - strlen(form), ",phased%d", p.phase_shift);
    if (p.edge_in) snprintf(form + strlen(form), sizeof form - strlen(form), ",edges");
    if (p.edge_fused) snprintf(form + strlen(form), sizeof form - strlen(form), ",edgeblocks");
    snprintf(g_kernel_name, sizeof g_kernel_name, "resize_run<ch%d,%s,Q%d,PX%d,TX%d,P%d,NACC%d,%s,bw%d%s%s%s>", CH, S16 ? "u16" : "u8", Q, PX, TX, P, NACC,
             p.fma ? "fma" : "exact", bw, RC ? ",ceil" : "", form, work_suffix(p.work));
    if (name) *name = g_kernel_name;
    return hipGetLastError();
}

// Instantiations per pixel format: X(Q, PX, TX, P, RC) — period in outputs, period in source pixels, taps, outputs per lane, window
// offsets inside a period rounded down / up; each
// with 4 and 6 pending output rows.  7.5x (15/2: a 4K frame to 512 columns): 21 taps with picha's default filter scale, 31
// (30 + the half pixel) at filter scale 1, for every filter but box / triangle.  Survey over formats (r03, 128 x 4K -> 512x288,
// profiles/r03_survey_run_kernel.txt): grey 31 taps -13 % against the row-staged kernel (fma -20 %), greya 31 taps -35 % (two
// outputs per lane), grey 21 taps -11 % on 256 images and a tie on 128; rgb / rgba / r16 were ties (-3 .. +4 %) and are not
// instantiated (rgba at 31 taps does not fit the edge kernel's LDS block either).  Integer scales with long windows (period 1; 128 x 4K ->
// 640x360 / 480x270, picha's defaults: 17 / 23 taps; profiles/r03_survey_run_kernel_integer_scales.txt): grey -9 % / -13 % against the
// accumulator kernel (fma -14 % / -4 %); greya, rgb, rgba -2 .. +7 %: not instantiated.  (lanczos at those scales is ring-aliased in y:
// the accumulator kernel's row list serves it, capi.cpp get_yrows.)  1.5x (period 2 / 3 source pixels, 4 taps with the offsets rounded up;
// 256 x 1080p -> 720p, profiles/r03_ab_run_kernel.txt): grey with eight outputs per lane -31 % (fma -40 %), greya -9 %; rgb +5 %, rgba -1 %:
// not instantiated.  More periodic scales with picha's defaults (128 x 4K sources, scripts/r03_call_scales.sh, same file): grey 2.5x -26 %, 3x -11 %,
// 5x -11 % (4x: a tie with the streaming kernel); greya 2.5x -11 %, 4x -14 %, 5x -11 % (3x: the symmetric-table kernel stays ahead).
#define PICHA_RUN_LIST_0(X) /* rgb */
#define PICHA_RUN_LIST_1(X) /* rgba */
#define PICHA_RUN_LIST_2(X) /* grey */ X(1, 3, 8, 8, 0) X(1, 5, 14, 4, 0) X(2, 5, 7, 8, 1) X(2, 3, 4, 8, 1) X(2, 3, 4, 4, 1) X(1, 6, 17, 4, 0) X(1, 8, 23, 4, 0) X(1, 8, 23, 8, 0) X(2, 15, 21, 2, 0) X(2, 15, 21, 4, 0) X(2, 15, 21, 8, 0) X(2, 15, 31, 2, 0) X(2, 15, 31, 4, 0) X(2, 15, 30, 4, 1)
#define PICHA_RUN_LIST_3(X) /* greya */ X(1, 4, 11, 4, 0) X(1, 5, 14, 4, 0) X(2, 5, 7, 4, 1) X(2, 3, 4, 8, 1) X(2, 3, 4, 4, 1) X(2, 15, 21, 2, 0) X(2, 15, 31, 2, 0) X(2, 15, 31, 4, 0) X(2, 15, 30, 2, 1)
#define PICHA_RUN_LIST_4(X) /* r16 */
#define PICHA_RUN_LIST_5(X) /* r16g16 */
#define PICHA_RUN_LIST_6(X) /* r16g16b16 */
#define PICHA_RUN_LIST_7(X) /* r16g16b16a16 */
// r04 forms: Z(Q, PX, TX, P, RC, XHL, WS) — XHL >= 0: neighbour exchange with XHL window pixels in front of the lane's own (dword-sized
// pixels, integer scales: cfg5 = 8K r16g16b16a16 at 4x, 15 taps, windows start 5 pixels before the lane's own eight); WS: phases of the
// period whose weights stay in SGPRs (cfg3b: one of the two phases of 7.5x).
// (r05) the shared-unpack shapes (XHL = -2) keep BOTH phases' weights in SGPRs and two row buffers: 87 VGPRs, five waves per SIMD (-4.5 % against one scalar phase
// and three buffers at 114; a 6- or 8-wave register budget spills and loses: profiles/r05_ab_shu_occupancy.txt).
// Measured and not kept (profiles/r04_ab_run_ws.txt): both phases scalar (97 VGPRs, one short of a fifth wave per SIMD: +2 % exact, +12 % fma),
// two row buffers instead of three (a tie), eight outputs per lane for the 30-tap shape (143 VGPRs once its 82-pixel window loop unrolls — see the
// Makefile's -pragma-unroll-threshold; rolled, it indexed its registers dynamically and took 17 ms: -2.5 % exact at 12 chunks, +3 % fma).
#define PICHA_RUNZ_LIST_0(Z) /* rgb */ Z(2, 15, 21, 2, 0, -2, 2) Z(1, 6, 17, 2, 0, -2, 1) Z(1, 8, 23, 2, 0, -2, 1) Z(2, 15, 30, 2, 1, -2, 2) Z(2, 15, 31, 2, 0, -2, 2) Z(1, 8, 33, 2, 0, -2, 1) Z(1, 6, 25, 2, 0, -2, 1) /* (lanczos: windows over two neighbours; at 6x / 8x the vertical table is ring-aliased: the march follows the row list) */  /* (r05) shared unpack only; the plain forms at 7.5x / 21 taps (two or four outputs per lane, scalar phase weights or not) were -2 % exact, +7 % fma against resize_acc (r04, as r03 had found): not instantiated */
#define PICHA_RUNZ_LIST_1(Z) /* rgba */ Z(1, 4, 11, 2, 0, 3, 0) Z(1, 4, 15, 2, 0, 5, 0) Z(1, 3, 8, 2, 0, 2, 0)
#define PICHA_RUNZ_LIST_2(Z) /* grey */ Z(2, 15, 30, 4, 1, -1, 1) Z(2, 15, 31, 4, 0, -1, 1) Z(2, 15, 21, 4, 0, -1, 1) Z(2, 15, 30, 4, 1, -2, 2) Z(2, 15, 31, 4, 0, -2, 2) Z(2, 15, 21, 4, 0, -2, 2) Z(1, 8, 33, 4, 0, -2, 1) Z(1, 6, 25, 4, 0, -2, 1) Z(1, 8, 23, 4, 0, -2, 1) Z(1, 6, 17, 4, 0, -2, 1)
#define PICHA_RUNZ_LIST_3(Z) /* greya */ Z(2, 15, 30, 2, 1, -1, 1) Z(2, 15, 31, 2, 0, -1, 1) Z(2, 15, 30, 2, 1, -2, 2) Z(2, 15, 31, 2, 0, -2, 2) Z(2, 15, 21, 2, 0, -2, 2) Z(1, 8, 33, 2, 0, -2, 1) Z(1, 6, 25, 2, 0, -2, 1) Z(1, 8, 23, 2, 0, -2, 1) Z(1, 6, 17, 2, 0, -2, 1)
#define PICHA_RUNZ_LIST_4(Z) /* r16 */
#define PICHA_RUNZ_LIST_5(Z) /* r16g16 */
#define PICHA_RUNZ_LIST_6(Z) /* r16g16b16 */
#define PICHA_RUNZ_LIST_7(Z) /* r16g16b16a16 */ Z(1, 4, 15, 2, 0, 5, 0) Z(1, 4, 11, 2, 0, 3, 0)

// Ring-kernel instantiations: U(Q, PX, TX, P, RC, K, TYM) — period in outputs / source pixels, x taps, outputs per lane, window
// offsets inside a period rounded down / up, ring rows, most vertical taps.  Upscales (Q = 2, 4; PX = 1; K = 4): 2x with picha's
// defaults has 3 taps with the offsets rounded up, the 4-tap filters round down.  Ring-aliased downscales: lanczos 2x (9 x 9 taps
// on a ring of 8).
#define PICHA_UP_LIST_0(U) /* rgb */ U(2, 1, 3, 4, 1, 4, 4) U(2, 1, 4, 4, 0, 4, 4) U(4, 1, 4, 8, 0, 4, 4) U(1, 2, 9, 2, 0, 8, 9) U(4, 1, 4, 4, 0, 4, 4)
#define PICHA_UP_LIST_1(U) /* rgba */ U(1, 2, 9, 2, 0, 8, 9) U(2, 1, 3, 2, 1, 4, 4) U(2, 1, 3, 4, 1, 4, 4) U(2, 1, 4, 2, 0, 4, 4) U(2, 1, 4, 4, 0, 4, 4) U(4, 1, 4, 8, 0, 4, 4) U(4, 1, 4, 4, 0, 4, 4)
#define PICHA_UP_LIST_2(U) /* grey */ U(1, 2, 9, 4, 0, 8, 9) U(1, 4, 17, 4, 0, 16, 17) U(2, 1, 3, 4, 1, 4, 4) U(2, 1, 3, 8, 1, 4, 4) U(2, 1, 4, 4, 0, 4, 4) U(2, 1, 4, 8, 0, 4, 4) U(4, 1, 4, 8, 0, 4, 4)
#define PICHA_UP_LIST_3(U) /* greya */ U(1, 2, 9, 4, 0, 8, 9) U(1, 4, 17, 2, 0, 16, 17) U(2, 1, 3, 4, 1, 4, 4) U(2, 1, 3, 8, 1, 4, 4) U(2, 1, 4, 4, 0, 4, 4) U(2, 1, 4, 8, 0, 4, 4) U(4, 1, 4, 8, 0, 4, 4)
#define PICHA_UP_LIST_4(U) /* r16 */ U(1, 2, 9, 4, 0, 8, 9) U(2, 1, 3, 4, 1, 4, 4) U(2, 1, 3, 8, 1, 4, 4) U(2, 1, 4, 4, 0, 4, 4) U(4, 1, 4, 8, 0, 4, 4)
#define PICHA_UP_LIST_5(U) /* r16g16 */ U(1, 2, 9, 2, 0, 8, 9) U(2, 1, 3, 2, 1, 4, 4) U(2, 1, 3, 4, 1, 4, 4) U(2, 1, 4, 4, 0, 4, 4)
#define PICHA_UP_LIST_6(U) /* r16g16b16 */ U(1, 2, 9, 2, 0, 8, 9) U(2, 1, 3, 2, 1, 4, 4) U(2, 1, 3, 4, 1, 4, 4) U(2, 1, 4, 2, 0, 4, 4)
#define PICHA_UP_LIST_7(U) /* r16g16b16a16 */ U(1, 2, 9, 2, 0, 8, 9) U(2, 1, 3, 2, 1, 4, 4) U(2, 1, 4, 2, 0, 4, 4) U(4, 1, 4, 4, 0, 4, 4)   /* upscales: four outputs per lane measured 13 % slower than the accumulator form (0.49 against 0.56); two (r05), with the clipped columns inside: see capi.cpp */

// Convert-in-front instantiations of the ring kernel (r04): V(SRCP, Q, PX, TX, P, RC, K, TYM) under the DESTINATION format's translation unit — the codec-edge
// pairs of LAB.md 6.3 (-> jpeg: rgb / grey; -> webp: rgb / rgba; deep png -> 8 bit) at 2x with picha's defaults (3 taps, offsets rounded up) and at 4x.
#define PICHA_UPCV_LIST_0(V) /* -> rgb */ V(1, 2, 1, 3, 4, 1, 4, 4) V(2, 2, 1, 3, 4, 1, 4, 4) V(6, 2, 1, 3, 4, 1, 4, 4) V(7, 2, 1, 3, 4, 1, 4, 4) \
    V(1, 4, 1, 4, 8, 0, 4, 4) V(2, 4, 1, 4, 8, 0, 4, 4)
#define PICHA_UPCV_LIST_1(V) /* -> rgba */ V(0, 2, 1, 3, 4, 1, 4, 4) V(3, 2, 1, 3, 4, 1, 4, 4) V(7, 2, 1, 3, 4, 1, 4, 4) \
    V(0, 4, 1, 4, 8, 0, 4, 4) V(3, 4, 1, 4, 8, 0, 4, 4)
#define PICHA_UPCV_LIST_2(V) /* -> grey */ V(0, 2, 1, 3, 8, 1, 4, 4) V(1, 2, 1, 3, 8, 1, 4, 4) V(3, 2, 1, 3, 8, 1, 4, 4) V(4, 2, 1, 3, 8, 1, 4, 4) \
    V(0, 4, 1, 4, 8, 0, 4, 4) V(1, 4, 1, 4, 8, 0, 4, 4)
#define PICHA_UPCV_LIST_3(V)
#define PICHA_UPCV_LIST_4(V)
#define PICHA_UPCV_LIST_5(V)
#define PICHA_UPCV_LIST_6(V)
#define PICHA_UPCV_LIST_7(V)

}  // namespace

#ifndef PICHA_RUN_PIXEL
#error "compile with -DPICHA_RUN_PIXEL=0..7"
#endif
#ifdef PICHA_RUN_EXPERIMENT   // developer shortcut: ONE shape (seconds instead of minutes), e.g. -DPICHA_RUN_PIXEL=2 '-DPICHA_RUN_EXPERIMENT=Z(2,15,30,8,1,-1,1)'
#define PICHA_RUN_LIST_X(X)
#define PICHA_UP_LIST_X(U)
#define PICHA_RUNZ_LIST_X(Z) PICHA_RUN_EXPERIMENT
#define PICHA_UPCV_LIST_X(V)
#undef PICHA_RUN_PIXEL_LISTS
#define PICHA_RUN_PIXEL_LISTS X
#else
#define PICHA_RUN_PIXEL_LISTS PICHA_RUN_PIXEL
#endif
#define PICHA_CAT2(a, b) a##b
#define PICHA_CAT(a, b) PICHA_CAT2(a, b)
#define PICHA_RUN_LIST PICHA_CAT(PICHA_RUN_LIST_, PICHA_RUN_PIXEL_LISTS)
#define PICHA_RUNZ_LIST PICHA_CAT(PICHA_RUNZ_LIST_, PICHA_RUN_PIXEL_LISTS)
#define PICHA_UP_LIST PICHA_CAT(PICHA_UP_LIST_, PICHA_RUN_PIXEL_LISTS)
#define PICHA_UPCV_LIST PICHA_CAT(PICHA_UPCV_LIST_, PICHA_RUN_PIXEL_LISTS)

PICHA_PROBE_HOST(PICHA_CAT(picha_probe_run_p, PICHA_RUN_PIXEL))

bool PICHA_CAT(run_supported_pixel, PICHA_RUN_PIXEL)(int q, int px, int tx, int per_lane, int nacc, int rc, int xhl, int ws) {
    if (nacc != 4 && nacc != 6) return false;
#define X(Q, PX, TX, P, RC) if (q == Q && px == PX && tx == TX && per_lane == P && rc == RC && xhl < 0 && ws == 0) return true;
    PICHA_RUN_LIST(X)
#undef X
#define Z(Q, PX, TX, P, RC, XHL, WS) if (q == Q && px == PX && tx == TX && per_lane == P && rc == RC && xhl == XHL && ws == WS) return true;
    PICHA_RUNZ_LIST(Z)
#undef Z
    return false;
}

hipError_t PICHA_CAT(launch_run_pixel, PICHA_RUN_PIXEL)(const RunParams &p, hipStream_t s, const char **name) {
    [[maybe_unused]] constexpr int CH = pixel_channels(PICHA_RUN_PIXEL);
    [[maybe_unused]] constexpr bool S16 = pixel_is16(PICHA_RUN_PIXEL);
    [[maybe_unused]] const int xhl = p.xch == 2 ? -2 : p.xch ? -p.x_base : -1;
#define X(Q, PX, TX, P, RC)                                                                               \
    if (p.period == Q && p.period_px == PX && p.x_taps == TX && p.x_per_lane == P && p.off_ceil == RC && xhl < 0 && p.wsgpr == 0) {  \
        if (p.nacc == 4) return launch_run_t<CH, S16, Q, PX, TX, P, 4, RC>(p, s, name);                   \
        if (p.nacc == 6) return launch_run_t<CH, S16, Q, PX, TX, P, 6, RC>(p, s, name);                   \
    }
    PICHA_RUN_LIST(X)
#undef X
#define Z(Q, PX, TX, P, RC, XHL, WS)                                                                      \
    if (p.period == Q && p.period_px == PX && p.x_taps == TX && p.x_per_lane == P && p.off_ceil == RC && xhl == XHL && p.wsgpr == WS) {  \
        if constexpr (XHL >= 0) {   /* phased stores: the neighbour-exchange shapes */                     \
            if (p.phase_shift > 0 && p.nacc == 4) return launch_run_t<CH, S16, Q, PX, TX, P, 4, RC, XHL, WS, 1>(p, s, name); \
            if (p.phase_shift > 0 && p.nacc == 6) return launch_run_t<CH, S16, Q, PX, TX, P, 6, RC, XHL, WS, 1>(p, s, name); \
        }                                                                                                  \
        if (p.nacc == 4) return launch_run_t<CH, S16, Q, PX, TX, P, 4, RC, XHL, PICHA_RUN_WS_OF(WS)>(p, s, name);          \
        if (p.nacc == 6) return launch_run_t<CH, S16, Q, PX, TX, P, 6, RC, XHL, PICHA_RUN_WS_OF(WS)>(p, s, name);          \
    }
    PICHA_RUNZ_LIST(Z)
#undef Z
    return hipErrorInvalidValue;
}

bool PICHA_CAT(up_supported_pixel, PICHA_RUN_PIXEL)(int q, int px, int tx, int per_lane, int rc, int ring, int ty) {
#define U(Q, PX, TX, P, RC, K, TYM) if (q == Q && px == PX && tx == TX && per_lane == P && rc == RC && ring == K && ty <= TYM) return true;
    PICHA_UP_LIST(U)
#undef U
    return false;
}

bool PICHA_CAT(up_cv_supported_pixel, PICHA_RUN_PIXEL)(int src_pixel, int q, int px, int tx, int per_lane, int rc, int ring, int ty) {
#define V(SRCP, Q, PX, TX, P, RC, K, TYM) if (src_pixel == SRCP && q == Q && px == PX && tx == TX && per_lane == P && rc == RC && ring == K && ty <= TYM) return true;
    PICHA_UPCV_LIST(V)
#undef V
    return false;
}

hipError_t PICHA_CAT(launch_up_pixel, PICHA_RUN_PIXEL)(const RunParams &p, hipStream_t s, const char **name) {
    [[maybe_unused]] constexpr int CH = pixel_channels(PICHA_RUN_PIXEL);
    [[maybe_unused]] constexpr bool S16 = pixel_is16(PICHA_RUN_PIXEL);
    if (p.cv_on) {
#define V(SRCP, Q, PX, TX, P, RC, K, TYM) if (p.cv_src_pixel == SRCP && p.period == Q && p.period_px == PX && p.x_taps == TX && p.x_per_lane == P && p.off_ceil == RC && p.ring_rows == K) return launch_ring_t<CH, S16, Q, PX, TX, P, RC, K, TYM, SRCP>(p, s, name);
        PICHA_UPCV_LIST(V)
#undef V
        return hipErrorInvalidValue;
    }
#define U(Q, PX, TX, P, RC, K, TYM) if (p.period == Q && p.period_px == PX && p.x_taps == TX && p.x_per_lane == P && p.off_ceil == RC && p.ring_rows == K) return launch_ring_t<CH, S16, Q, PX, TX, P, RC, K, TYM>(p, s, name);
    PICHA_UP_LIST(U)
#undef U
    return hipErrorInvalidValue;
}

hipError_t PICHA_CAT(launch_edges_pixel, PICHA_RUN_PIXEL)(const RunParams &p, hipStream_t s) {
    [[maybe_unused]] constexpr int CH = pixel_channels(PICHA_RUN_PIXEL);
    [[maybe_unused]] constexpr bool S16 = pixel_is16(PICHA_RUN_PIXEL);
    if (!p.stash && !p.edge_image) return hipSuccess;
    [[maybe_unused]] dim3 grid((p.dst_h + p.edge_rows - 1) / p.edge_rows, 2, p.n_images);
    [[maybe_unused]] const size_t lds = sizeof(float) * (p.edge_lds_floats > 0 && p.edge_lds_floats <= kEdgeFloats ? std::min(kEdgeFloats, (p.edge_lds_floats + 127) / 64 * 64) : kEdgeFloats);
    [[maybe_unused]] const int txe = (p.x_taps + 3) / 4 * 4;
    if (p.cv_on) {
        if (!p.edge_image) return hipErrorInvalidValue;
#define V(SRCP, Q, PX, TX, P, RC, K, TYM) if (p.cv_src_pixel == SRCP && txe == (TX + 3) / 4 * 4) { resize_edges_kernel<CH, S16, (TX + 3) / 4 * 4, SRCP><<<grid, 64, lds, s>>>(p); return hipGetLastError(); }
        PICHA_UPCV_LIST(V)
#undef V
        return hipErrorInvalidValue;
    }
#define X(Q, PX, TX, P, RC) if (txe == (TX + 3) / 4 * 4) { resize_edges_kernel<CH, S16, (TX + 3) / 4 * 4><<<grid, 64, lds, s>>>(p); return hipGetLastError(); }
    PICHA_RUN_LIST(X)
#undef X
#define Z(Q, PX, TX, P, RC, XHL, WS) if (txe == (TX + 3) / 4 * 4) { resize_edges_kernel<CH, S16, (TX + 3) / 4 * 4><<<grid, 64, lds, s>>>(p); return hipGetLastError(); }
    PICHA_RUNZ_LIST(Z)
#undef Z
#define U(Q, PX, TX, P, RC, K, TYM) if (txe == (TX + 3) / 4 * 4) { resize_edges_kernel<CH, S16, (TX + 3) / 4 * 4><<<grid, 64, lds, s>>>(p); return hipGetLastError(); }
    PICHA_UP_LIST(U)
#undef U
    return hipErrorInvalidValue;
}

}  // namespace picha_hip
