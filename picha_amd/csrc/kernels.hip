// kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels for picha's pixel hot path.
//
//   color_convert_kernel   elementwise unpack -> channel map -> pack        (src/colorconvert.cc:136-152)
//   hpass/vpass_generic    two-pass resample through an HBM float scratch   (src/resize.cc:99-132)
//   resize_fused_kernel    the hot kernel: per-wave column strip marching down the image,
//                          horizontal taps from registers, intermediate rows in a lane-private
//                          LDS ring, vertical taps from SGPRs                (src/resize.cc:99-132)
//   fill_random_kernel     synthetic pixels for the benchmark
//
// Numerics contract: IEEE binary32, every multiply and add rounded separately
// (NO fused multiply-add: this file is compiled with -ffp-contract=off and carries the pragma
// below), taps accumulated in ascending source order, unpack = float(s) * (1.0f / max),
// pack = trunc(clamp(v * max + 0.5f, 0, max)).  Tap weights are never computed on the device.
#include "kernels.h"

#include <algorithm>

#pragma clang fp contract(off)

namespace picha_hip {
namespace {

constexpr int kWave = 64;

// ---------------------------------------------------------------------------------------------
// Pixel model (src/picha.h:79-172)
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr int pixel_channels(int p) {
    return p == 0 ? 3 : p == 1 ? 4 : p == 2 ? 1 : p == 3 ? 2 : p == 4 ? 1 : p == 5 ? 2 : p == 6 ? 3 : 4;
}
__host__ __device__ constexpr bool pixel_is16(int p) { return p >= 4; }

template <bool S16> struct Sample {
    static constexpr float kMax = S16 ? 65535.0f : 255.0f;
    static constexpr float kInv = 1.0f / kMax;  // the reference multiplies by the rounded reciprocal, picha.h:104
    static constexpr int kBytes = S16 ? 2 : 1;
};

// i::linear_unpack, src/picha.h:98-105
template <bool S16> __device__ __forceinline__ float unpack_sample(uint32_t s) {
    return float(s) * Sample<S16>::kInv;
}

// i::linear_pack, src/picha.h:107-114: T(max(0, min(max, 0 + v * max + 0.5f))); inputs are finite.
template <bool S16> __device__ __forceinline__ uint32_t pack_sample(float v) {
    float t = v * Sample<S16>::kMax;
    t = t + 0.5f;
    t = __builtin_amdgcn_fmed3f(t, 0.0f, Sample<S16>::kMax);
    return uint32_t(t);
}

template <int NW> __device__ __forceinline__ void load_words(const uint8_t *p, uint32_t *w) {
    if constexpr (NW == 1) {
        w[0] = *reinterpret_cast<const uint32_t *>(p);
    } else if constexpr (NW == 2) {
        uint2 v = *reinterpret_cast<const uint2 *>(p);
        w[0] = v.x; w[1] = v.y;
    } else if constexpr (NW == 3) {
        // 12 bytes: three dwords (row bases are only 4-byte aligned in general)
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
        w[0] = q[0]; w[1] = q[1]; w[2] = q[2];
    } else {
        static_assert(NW % 2 == 0, "word count");
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
#pragma unroll
        for (int i = 0; i < NW; ++i) w[i] = q[i];
    }
}
template <int NW> __device__ __forceinline__ void store_words(uint8_t *p, const uint32_t *w) {
    uint32_t *q = reinterpret_cast<uint32_t *>(p);
#pragma unroll
    for (int i = 0; i < NW; ++i) q[i] = w[i];
}

template <bool S16> __device__ __forceinline__ uint32_t load_sample(const uint8_t *p) {
    if constexpr (S16) return uint32_t(p[0]) | (uint32_t(p[1]) << 8);  // byte loads: any alignment
    else return p[0];
}
template <bool S16> __device__ __forceinline__ void store_sample(uint8_t *p, uint32_t v) {
    p[0] = uint8_t(v);
    if constexpr (S16) p[1] = uint8_t(v >> 8);
}

// ---------------------------------------------------------------------------------------------
// Colour convert (src/colorconvert.cc:24-152)
// ---------------------------------------------------------------------------------------------

// ChannelConvertOp<SC, DC>: selected by channel count only (src/colorconvert.cc:24-134).
template <int SC, int DC>
__device__ __forceinline__ void channel_map(const float *s, float *d, float r, float g, float b) {
    if constexpr (SC == DC) {
#pragma unroll
        for (int i = 0; i < SC; ++i) d[i] = s[i];
    } else if constexpr (SC == 1) {
        d[0] = s[0];
        if constexpr (DC == 2) d[1] = 1.0f;
        if constexpr (DC >= 3) { d[1] = s[0]; d[2] = s[0]; }
        if constexpr (DC == 4) d[3] = 1.0f;
    } else if constexpr (SC == 2) {
        d[0] = s[0];
        if constexpr (DC == 3) { d[1] = s[1]; d[2] = 0.0f; }             // quirk: second channel -> G, B = 0
        if constexpr (DC == 4) { d[1] = s[0]; d[2] = s[0]; d[3] = s[1]; }  // second channel -> alpha
    } else if constexpr (DC <= 2) {  // SC 3 or 4 -> luma
        float l = s[0] * r;
        l = l + s[1] * g;  // (s0*r + s1*g) + s2*b, each product and sum rounded (colorconvert.cc:90)
        l = l + s[2] * b;
        d[0] = l;
        if constexpr (DC == 2) d[1] = (SC == 4) ? s[3] : 1.0f;
    } else {  // 3 -> 4 or 4 -> 3
        d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
        if constexpr (DC == 4) d[3] = 1.0f;
    }
}

template <int SC, bool S16S, int DC, bool S16D>
__global__ __launch_bounds__(256) void color_convert_kernel(ConvertParams p) {
    constexpr int SB = SC * Sample<S16S>::kBytes, DB = DC * Sample<S16D>::kBytes;
    constexpr int PX = 4;  // pixels per thread: PX*SB and PX*DB are multiples of 4 for every format
    constexpr int SW = PX * SB / 4, DW = PX * DB / 4;

    const int groups = (p.width + PX - 1) / PX;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    const int img = blockIdx.z;
    const uint8_t *sbase = p.src_tab ? static_cast<const uint8_t *>(p.src_tab[img]) : p.src + img * p.src_pitch;
    uint8_t *dbase = p.dst_tab ? static_cast<uint8_t *>(p.dst_tab[img]) : p.dst + img * p.dst_pitch;
    const bool aligned = (((reinterpret_cast<uintptr_t>(sbase) | uintptr_t(p.src_stride)) & 3) == 0) &&
                         (((reinterpret_cast<uintptr_t>(dbase) | uintptr_t(p.dst_stride)) & 3) == 0);
    const int x0 = g * PX;
    const int npx = min(PX, p.width - x0);

    for (int y = blockIdx.y; y < p.height; y += gridDim.y) {
        const uint8_t *srow = sbase + (long long)y * p.src_stride + (long long)x0 * SB;
        uint8_t *drow = dbase + (long long)y * p.dst_stride + (long long)x0 * DB;
        if (aligned && npx == PX) {
            uint32_t in[SW], out[DW];
            load_words<SW>(srow, in);
#pragma unroll
            for (int i = 0; i < DW; ++i) out[i] = 0;
#pragma unroll
            for (int q = 0; q < PX; ++q) {
                float s[SC], d[DC];
#pragma unroll
                for (int c = 0; c < SC; ++c) {
                    const int k = q * SC + c;  // constant after unrolling: the extraction folds into the convert
                    const uint32_t raw = S16S ? ((in[k / 2] >> (16 * (k % 2))) & 0xffffu) : ((in[k / 4] >> (8 * (k % 4))) & 0xffu);
                    s[c] = unpack_sample<S16S>(raw);
                }
                channel_map<SC, DC>(s, d, p.r, p.g, p.b);
#pragma unroll
                for (int c = 0; c < DC; ++c) {
                    const int k = q * DC + c;
                    const uint32_t v = pack_sample<S16D>(d[c]);
                    if constexpr (S16D) out[k / 2] |= v << (16 * (k % 2));
                    else out[k / 4] |= v << (8 * (k % 4));
                }
            }
            store_words<DW>(drow, out);
        } else {
            for (int q = 0; q < npx; ++q) {
                float s[SC], d[DC];
#pragma unroll
                for (int c = 0; c < SC; ++c)
                    s[c] = unpack_sample<S16S>(load_sample<S16S>(srow + q * SB + c * Sample<S16S>::kBytes));
                channel_map<SC, DC>(s, d, p.r, p.g, p.b);
#pragma unroll
                for (int c = 0; c < DC; ++c)
                    store_sample<S16D>(drow + q * DB + c * Sample<S16D>::kBytes, pack_sample<S16D>(d[c]));
            }
        }
    }
}

template <int SP, int DP>
hipError_t launch_cc(const ConvertParams &p, hipStream_t s) {
    constexpr int PX = 4;
    const int groups = (p.width + PX - 1) / PX;
    dim3 grid((groups + 255) / 256, 1, p.n_images);
    // enough row-blocks to fill the chip, each looping over rows with stride gridDim.y
    long long want = 4096 / ((long long)grid.x * grid.z) + 1;
    grid.y = (unsigned)std::min<long long>(p.height, std::max<long long>(1, std::min<long long>(want * 8, 65535)));
    if ((long long)grid.x * grid.z >= 4096) grid.y = std::min(p.height, 64);
    color_convert_kernel<pixel_channels(SP), pixel_is16(SP), pixel_channels(DP), pixel_is16(DP)>
        <<<grid, 256, 0, s>>>(p);
    return hipGetLastError();
}

template <int SP>
hipError_t launch_cc_src(int dp, const ConvertParams &p, hipStream_t s) {
    switch (dp) {
    case 0: return launch_cc<SP, 0>(p, s);
    case 1: return launch_cc<SP, 1>(p, s);
    case 2: return launch_cc<SP, 2>(p, s);
    case 3: return launch_cc<SP, 3>(p, s);
    case 4: return launch_cc<SP, 4>(p, s);
    case 5: return launch_cc<SP, 5>(p, s);
    case 6: return launch_cc<SP, 6>(p, s);
    case 7: return launch_cc<SP, 7>(p, s);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// Two-pass resize through an HBM scratch: handles every geometry / alignment.  One thread per
// (output column, source row) in the horizontal pass, one per output pixel in the vertical pass.
// ---------------------------------------------------------------------------------------------
template <int CH, bool S16>
__global__ __launch_bounds__(256) void hpass_generic_kernel(TwoPassParams p) {
    constexpr int BPS = Sample<S16>::kBytes, BPP = CH * BPS;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= p.dst_w) return;
    const int left = p.x_left[x], n = p.x_count[x];
    const float *w = p.x_weight + (size_t)x * p.x_taps;
    for (int rr = blockIdx.y; rr < p.nrows; rr += gridDim.y) {
        const uint8_t *row = p.src + (long long)(p.row0 + rr) * p.src_stride;
        float acc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = 0.0f;  // memset(tmprow, 0) resize.cc:109
        for (int j = 0; j < n; ++j) {
            const uint8_t *px = row + (long long)(left + j) * BPP;
            const float wj = w[j];
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = acc[c] + wj * unpack_sample<S16>(load_sample<S16>(px + c * BPS));
        }
        float *o = p.scratch + ((size_t)rr * p.dst_w + x) * CH;
#pragma unroll
        for (int c = 0; c < CH; ++c) o[c] = acc[c];
    }
}

template <int CH, bool S16>
__global__ __launch_bounds__(256) void vpass_generic_kernel(TwoPassParams p) {
    constexpr int BPS = Sample<S16>::kBytes, BPP = CH * BPS;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= p.dst_w) return;
    for (int y = blockIdx.y; y < p.dst_h; y += gridDim.y) {
        const int n = p.y_count[y];
        const float *w = p.y_weight + (size_t)y * p.y_taps;
        const int32_t *eff = p.y_eff + (size_t)y * p.y_taps;
        float acc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = 0.0f;  // float pixel[] = {} resize.cc:124
        for (int j = 0; j < n; ++j) {
            const float *t = p.scratch + ((size_t)(eff[j] - p.row0) * p.dst_w + x) * CH;
            const float wj = w[j];
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = acc[c] + wj * t[c];
        }
        uint8_t *o = p.dst + (long long)y * p.dst_stride + (long long)x * BPP;
#pragma unroll
        for (int c = 0; c < CH; ++c) store_sample<S16>(o + c * BPS, pack_sample<S16>(acc[c]));
    }
}

template <int CH, bool S16>
hipError_t launch_twopass_t(const TwoPassParams &p, hipStream_t s) {
    dim3 gh((p.dst_w + 255) / 256, std::min(p.nrows, 4096));
    hpass_generic_kernel<CH, S16><<<gh, 256, 0, s>>>(p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    dim3 gv((p.dst_w + 255) / 256, std::min(p.dst_h, 4096));
    vpass_generic_kernel<CH, S16><<<gv, 256, 0, s>>>(p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Fused resize kernel.
//
// Work decomposition: one WAVE owns 64 adjacent output columns of one row-chunk of one image and
// marches down the source rows that chunk needs.  A workgroup is four such waves (256 adjacent
// output columns -> the workgroup streams a contiguous span of every source row).  There is no
// cross-lane or cross-wave data flow and therefore no barrier anywhere:
//   * lane x keeps its TX horizontal weights in VGPRs for the whole march;
//   * each source row is resampled horizontally into one value per lane (all channels) and
//     written to that lane's cell of a per-wave LDS ring of `ring_slots` rows — the GPU analogue
//     of the reference's FloatBuffer ring (src/resize.cc:52-64,83);
//   * when the last row an output row needs has been resampled, the vertical taps (wave-uniform,
//     read through scalar loads) combine ring rows and the packed pixel is stored.
// Source reads are software-prefetched D rows ahead in registers.
// ---------------------------------------------------------------------------------------------
template <int CH> struct RingSlot { using type = float4; static constexpr int kFloats = 4; };
template <> struct RingSlot<1> { using type = float; static constexpr int kFloats = 1; };
template <> struct RingSlot<2> { using type = float2; static constexpr int kFloats = 2; };

// raw pixel as loaded (one tap): number of 32-bit registers
template <int CH, bool S16> struct RawPx {
    static constexpr int BPP = CH * Sample<S16>::kBytes;
    static constexpr int kRegs = (BPP == 8) ? 2 : (CH == 3 ? 3 : 1);
};

template <int CH, bool S16>
__device__ __forceinline__ void load_px(const uint8_t *p, uint32_t *r) {
    constexpr int BPP = RawPx<CH, S16>::BPP;
    if constexpr (BPP == 1) r[0] = *p;
    else if constexpr (BPP == 2) r[0] = *reinterpret_cast<const uint16_t *>(p);
    else if constexpr (BPP == 4) r[0] = *reinterpret_cast<const uint32_t *>(p);
    else if constexpr (BPP == 8) { uint2 v = *reinterpret_cast<const uint2 *>(p); r[0] = v.x; r[1] = v.y; }
    else if constexpr (BPP == 3) { r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; }
    else { const uint16_t *q = reinterpret_cast<const uint16_t *>(p); r[0] = q[0]; r[1] = q[1]; r[2] = q[2]; }  // BPP 6
}

template <int CH, bool S16>
__device__ __forceinline__ void unpack_px(const uint32_t *r, float *u) {
    constexpr int BPP = RawPx<CH, S16>::BPP;
    if constexpr (CH == 3) {
        u[0] = unpack_sample<S16>(r[0]); u[1] = unpack_sample<S16>(r[1]); u[2] = unpack_sample<S16>(r[2]);
    } else if constexpr (BPP == 1 || (BPP == 2 && S16)) {
        u[0] = unpack_sample<S16>(r[0]);
    } else if constexpr (BPP == 2) {  // greya
        u[0] = unpack_sample<false>(r[0] & 0xffu); u[1] = unpack_sample<false>(r[0] >> 8);
    } else if constexpr (BPP == 4 && !S16) {  // rgba
        u[0] = unpack_sample<false>(r[0] & 0xffu); u[1] = unpack_sample<false>((r[0] >> 8) & 0xffu);
        u[2] = unpack_sample<false>((r[0] >> 16) & 0xffu); u[3] = unpack_sample<false>(r[0] >> 24);
    } else if constexpr (BPP == 4) {  // r16g16
        u[0] = unpack_sample<true>(r[0] & 0xffffu); u[1] = unpack_sample<true>(r[0] >> 16);
    } else {  // r16g16b16a16
        u[0] = unpack_sample<true>(r[0] & 0xffffu); u[1] = unpack_sample<true>(r[0] >> 16);
        u[2] = unpack_sample<true>(r[1] & 0xffffu); u[3] = unpack_sample<true>(r[1] >> 16);
    }
}

template <int CH, bool S16>
__device__ __forceinline__ void pack_store_px(uint8_t *p, const float *v) {
    constexpr int BPP = RawPx<CH, S16>::BPP;
    uint32_t q[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) q[c] = pack_sample<S16>(v[c]);
    if constexpr (BPP == 1) *p = uint8_t(q[0]);
    else if constexpr (BPP == 2 && S16) *reinterpret_cast<uint16_t *>(p) = uint16_t(q[0]);
    else if constexpr (BPP == 2) *reinterpret_cast<uint16_t *>(p) = uint16_t(q[0] | (q[1] << 8));
    else if constexpr (BPP == 3) { p[0] = uint8_t(q[0]); p[1] = uint8_t(q[1]); p[2] = uint8_t(q[2]); }
    else if constexpr (BPP == 4 && !S16) *reinterpret_cast<uint32_t *>(p) = q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24);
    else if constexpr (BPP == 4) *reinterpret_cast<uint32_t *>(p) = q[0] | (q[1] << 16);
    else if constexpr (BPP == 6) {
        uint16_t *o = reinterpret_cast<uint16_t *>(p);
        o[0] = uint16_t(q[0]); o[1] = uint16_t(q[1]); o[2] = uint16_t(q[2]);
    } else {
        uint2 o; o.x = q[0] | (q[1] << 16); o.y = q[2] | (q[3] << 16);
        *reinterpret_cast<uint2 *>(p) = o;
    }
}

template <int CH, bool S16, int TX, int TYB, int D>
__global__ __launch_bounds__(256) void resize_fused_kernel(FusedParams p) {
    constexpr int BPP = RawPx<CH, S16>::BPP;
    constexpr int NR = RawPx<CH, S16>::kRegs;
    using Slot = typename RingSlot<CH>::type;
    constexpr int SF = RingSlot<CH>::kFloats;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int col0 = (blockIdx.x * 4 + wave) * kWave;
    if (col0 >= p.dst_w) return;  // wave-uniform: this wave has no columns (no barriers in this kernel)
    const int x = col0 + lane;
    const bool active = x < p.dst_w;
    const int xc = active ? x : p.dst_w - 1;

    const int img = blockIdx.z;
    const uint8_t *sbase = p.src_tab ? static_cast<const uint8_t *>(p.src_tab[img]) : p.src + img * p.src_pitch;
    uint8_t *dbase = p.dst_tab ? static_cast<uint8_t *>(p.dst_tab[img]) : p.dst + img * p.dst_pitch;

    const int y0 = blockIdx.y * p.rows_per_chunk;
    const int y1 = min(p.dst_h, y0 + p.rows_per_chunk);
    if (y0 >= y1) return;

    // lane-resident horizontal taps
    float wx[TX];
#pragma unroll
    for (int j = 0; j < TX; ++j) wx[j] = p.x_weight[(size_t)j * p.dst_w + xc];
    const uint32_t xoff = uint32_t(p.x_left[xc]) * BPP;

    Slot *ring = reinterpret_cast<Slot *>(lds_raw) + (size_t)wave * p.ring_slots * kWave + lane;

    const int r0 = p.y_info[y0].x;       // first source row this chunk touches
    const int r1 = p.y_info[y1 - 1].y;   // last
    int slot = r0 % p.ring_slots;
    int y = y0;
    int cur_need = p.y_info[y0].y;

    uint32_t raw[D][TX][NR];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const uint8_t *row = sbase + (long long)min(r0 + d, r1) * p.src_stride;
#pragma unroll
        for (int j = 0; j < TX; ++j) load_px<CH, S16>(row + xoff + j * BPP, raw[d][j]);
    }

    for (int r = r0; r <= r1;) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (r <= r1) {
                // ---- horizontal pass of source row r (src/resize.cc:110-117) ----
                float acc[SF];
#pragma unroll
                for (int j = 0; j < TX; ++j) {
                    float u[CH];
                    unpack_px<CH, S16>(raw[d][j], u);
#pragma unroll
                    for (int c = 0; c < CH; ++c) acc[c] = (j == 0) ? wx[0] * u[c] : acc[c] + wx[j] * u[c];
                }
                if constexpr (CH == 3) acc[3] = 0.0f;
                // refill the prefetch slot with row r + D
                {
                    const uint8_t *row = sbase + (long long)min(r + D, r1) * p.src_stride;
#pragma unroll
                    for (int j = 0; j < TX; ++j) load_px<CH, S16>(row + xoff + j * BPP, raw[d][j]);
                }
                Slot sv;
                if constexpr (SF == 1) sv = acc[0];
                else if constexpr (SF == 2) { sv.x = acc[0]; sv.y = acc[1]; }
                else { sv.x = acc[0]; sv.y = acc[1]; sv.z = acc[2]; sv.w = acc[3]; }
                ring[slot * kWave] = sv;
                slot = (slot + 1 == p.ring_slots) ? 0 : slot + 1;

                // ---- vertical pass for every output row completed by row r (src/resize.cc:122-132) ----
                while (cur_need <= r) {
                    const int cnt = p.y_info[y].z;
                    const float *wy = p.y_weight + (size_t)y * p.y_pad;
                    const int32_t *sy = p.y_slot + (size_t)y * p.y_pad;
                    float o[SF];
#pragma unroll
                    for (int j = 0; j < TYB; ++j) {
                        if (j >= cnt) break;
                        const Slot v = ring[sy[j] * kWave];
                        const float w = wy[j];
                        float vf[SF];
                        if constexpr (SF == 1) vf[0] = v;
                        else if constexpr (SF == 2) { vf[0] = v.x; vf[1] = v.y; }
                        else { vf[0] = v.x; vf[1] = v.y; vf[2] = v.z; vf[3] = v.w; }
#pragma unroll
                        for (int c = 0; c < CH; ++c) o[c] = (j == 0) ? w * vf[c] : o[c] + w * vf[c];
                    }
                    if (active) pack_store_px<CH, S16>(dbase + (long long)y * p.dst_stride + (long long)x * BPP, o);
                    ++y;
                    cur_need = (y < y1) ? p.y_info[y].y : 0x7fffffff;
                }
                ++r;
            }
        }
    }
}

template <int CH, bool S16, int TX, int TYB>
hipError_t launch_fused_t(const FusedParams &p, hipStream_t s, const char **name) {
    constexpr int NR = RawPx<CH, S16>::kRegs;
    constexpr int D = (TX * NR <= 8) ? 4 : (TX * NR <= 16 ? 2 : 1);
    auto kern = resize_fused_kernel<CH, S16, TX, TYB, D>;
    const size_t lds = (size_t)4 * p.ring_slots * kWave * sizeof(typename RingSlot<CH>::type);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    dim3 grid((p.dst_w + 255) / 256, p.n_chunks, p.n_images);
    kern<<<grid, 256, lds, s>>>(p);
    if (name) *name = "resize_fused_kernel";
    return hipGetLastError();
}

constexpr int kMaxFusedTX = 8;

template <int CH, bool S16, int TX>
hipError_t launch_fused_ty(const FusedParams &p, hipStream_t s, const char **name) {
    if (p.y_pad == 8) return launch_fused_t<CH, S16, TX, 8>(p, s, name);
    if (p.y_pad == 16) return launch_fused_t<CH, S16, TX, 16>(p, s, name);
    return launch_fused_t<CH, S16, TX, 32>(p, s, name);
}

template <int CH, bool S16>
hipError_t launch_fused_tx(const FusedParams &p, hipStream_t s, const char **name) {
    switch (p.x_taps) {
    case 1: return launch_fused_ty<CH, S16, 1>(p, s, name);
    case 2: return launch_fused_ty<CH, S16, 2>(p, s, name);
    case 3: return launch_fused_ty<CH, S16, 3>(p, s, name);
    case 4: return launch_fused_ty<CH, S16, 4>(p, s, name);
    case 5: return launch_fused_ty<CH, S16, 5>(p, s, name);
    case 6: return launch_fused_ty<CH, S16, 6>(p, s, name);
    case 7: return launch_fused_ty<CH, S16, 7>(p, s, name);
    case 8: return launch_fused_ty<CH, S16, 8>(p, s, name);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// Synthetic pixels: sample k = top bits of splitmix64(seed ^ k)
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void fill_random_kernel(uint8_t *dst, size_t bytes, uint64_t seed, int sample_bytes) {
    const size_t nwords = bytes / 4;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t wi = (size_t)blockIdx.x * 256 + threadIdx.x; wi < nwords; wi += stride) {
        uint32_t v = 0;
        if (sample_bytes == 1) {
#pragma unroll
            for (int b = 0; b < 4; ++b) v |= uint32_t(splitmix64(seed ^ (wi * 4 + b)) >> 56) << (8 * b);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) v |= uint32_t(splitmix64(seed ^ (wi * 2 + h)) >> 48) << (16 * h);
        }
        reinterpret_cast<uint32_t *>(dst)[wi] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // tail bytes
        for (size_t k = nwords * 4; k < bytes; ++k) {
            if (sample_bytes == 1) dst[k] = uint8_t(splitmix64(seed ^ k) >> 56);
            else dst[k] = uint8_t((splitmix64(seed ^ (k / 2)) >> 48) >> (8 * (k & 1)));
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Launch interface
// ---------------------------------------------------------------------------------------------
hipError_t launch_color_convert(int sp, int dp, const ConvertParams &p, hipStream_t s) {
    switch (sp) {
    case 0: return launch_cc_src<0>(dp, p, s);
    case 1: return launch_cc_src<1>(dp, p, s);
    case 2: return launch_cc_src<2>(dp, p, s);
    case 3: return launch_cc_src<3>(dp, p, s);
    case 4: return launch_cc_src<4>(dp, p, s);
    case 5: return launch_cc_src<5>(dp, p, s);
    case 6: return launch_cc_src<6>(dp, p, s);
    case 7: return launch_cc_src<7>(dp, p, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_resize_twopass(int pixel, const TwoPassParams &p, hipStream_t s) {
    switch (pixel) {
    case 0: return launch_twopass_t<3, false>(p, s);
    case 1: return launch_twopass_t<4, false>(p, s);
    case 2: return launch_twopass_t<1, false>(p, s);
    case 3: return launch_twopass_t<2, false>(p, s);
    case 4: return launch_twopass_t<1, true>(p, s);
    case 5: return launch_twopass_t<2, true>(p, s);
    case 6: return launch_twopass_t<3, true>(p, s);
    case 7: return launch_twopass_t<4, true>(p, s);
    }
    return hipErrorInvalidValue;
}

int fused_y_pad(int y_taps) { return y_taps <= 8 ? 8 : (y_taps <= 16 ? 16 : 32); }

bool fused_supported(int pixel, int x_taps, int y_taps, int ring_slots) {
    if (pixel < 0 || pixel > 7) return false;
    if (x_taps < 1 || x_taps > kMaxFusedTX) return false;
    if (y_taps < 1 || y_taps > 32) return false;
    const int ch = pixel_channels(pixel);
    const size_t slot = (ch == 1 ? 4 : ch == 2 ? 8 : 16);
    return (size_t)4 * ring_slots * kWave * slot <= 160 * 1024;
}

hipError_t launch_resize_fused(int pixel, const FusedParams &p, hipStream_t s, const char **name) {
    switch (pixel) {
    case 0: return launch_fused_tx<3, false>(p, s, name);
    case 1: return launch_fused_tx<4, false>(p, s, name);
    case 2: return launch_fused_tx<1, false>(p, s, name);
    case 3: return launch_fused_tx<2, false>(p, s, name);
    case 4: return launch_fused_tx<1, true>(p, s, name);
    case 5: return launch_fused_tx<2, true>(p, s, name);
    case 6: return launch_fused_tx<3, true>(p, s, name);
    case 7: return launch_fused_tx<4, true>(p, s, name);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_fill_random(void *dev, size_t bytes, uint64_t seed, int sample_bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    size_t blocks = std::min<size_t>((bytes / 4 + 255) / 256, 8192);
    if (blocks == 0) blocks = 1;
    fill_random_kernel<<<dim3((unsigned)blocks), 256, 0, s>>>(static_cast<uint8_t *>(dev), bytes, seed, sample_bytes);
    return hipGetLastError();
}

// host twin of the generator
void fill_random_host(void *host, size_t bytes, uint64_t seed, int sample_bytes) {
    uint8_t *d = static_cast<uint8_t *>(host);
    for (size_t k = 0; k < bytes; ++k) {
        if (sample_bytes == 1) d[k] = uint8_t(splitmix64(seed ^ k) >> 56);
        else d[k] = uint8_t((splitmix64(seed ^ (k / 2)) >> 48) >> (8 * (k & 1)));
    }
}

}  // namespace picha_hip
