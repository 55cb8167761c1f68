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
#include <cstdio>
#include <type_traits>

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
// Fused resize kernel ("span" kernel).
//
// Work decomposition: one WAVE owns 64*P adjacent output columns of one row-chunk of one image
// and marches down the source rows that chunk needs; a workgroup is four such waves (256*P
// adjacent output columns, so it streams one contiguous piece of every source row).  No data
// ever crosses lanes or waves, hence no barrier anywhere:
//   * a lane owns P consecutive output columns.  Per source row it loads ONE contiguous span of
//     S*(P-1)+TX source pixels (the union of its outputs' tap windows; output p's window starts
//     S*p pixels into the span — for P > 1 the host has verified that the tap table has this
//     uniform stride), converts every pixel of the span to float ONCE and applies its P*TX
//     lane-resident weights.  Loads are buffer loads against a descriptor of exactly the image's
//     bytes, so taps that hang over an image edge (they always carry weight 0) read as 0.
//   * the P resampled values go to the lane's cells of a per-wave LDS ring of `ring_slots` rows —
//     the GPU analogue of the reference's FloatBuffer ring (src/resize.cc:52-64,83);
//   * when the last source row an output row needs has been resampled, the vertical taps (slots
//     through scalar loads, weights broadcast into VGPRs) combine ring rows; the packed pixels
//     are stored with one buffer store per lane.
// Source rows are software-prefetched D rows ahead in registers.
//
// Numerics: FMA == false is the reference arithmetic (unfused, bit-exact with the CPU path);
// FMA == true fuses multiply-adds, folds the 1/max unpack scale into the horizontal weights and
// rounds to nearest on pack — results differ from the reference by at most one code value.
// ---------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) int32_t *const_i32_ptr;  // forces scalar (SMEM) loads
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int CH> struct RingSlot { using type = float4; static constexpr int kFloats = 4; };
template <> struct RingSlot<1> { using type = float; static constexpr int kFloats = 1; };
template <> struct RingSlot<2> { using type = float2; static constexpr int kFloats = 2; };

template <int CH, bool S16, int P, int S, int TX> struct SpanCfg {
    static constexpr int BPS = Sample<S16>::kBytes;
    static constexpr int BPP = CH * BPS;
    static constexpr int SPAN = S * (P - 1) + TX;       // source pixels a lane reads per row
    static constexpr bool WORDS = (BPP % 4 == 0);       // span loadable as whole dwords
    static constexpr int PXREGS = WORDS ? BPP / 4 : (CH == 3 ? 3 : 1);
    static constexpr int NW = SPAN * PXREGS;            // raw registers per row
    static constexpr int D = NW <= 8 ? 4 : (NW <= 16 ? 3 : (NW <= 32 ? 2 : 1));  // prefetch depth
    static_assert(WORDS || P == 1, "multi-output lanes need dword-sized pixels");
};

// CAREFUL: some lane of this wave has a span that starts before the image (negative offset; only the
// left-most lane group of multi-output lanes, and only in row 0, where there is no previous row for
// the offset to land in).  A wrapped 32-bit offset would put the WHOLE merged multi-dword load out
// of range, so such waves clamp every dword's offset separately (the clamped dwords only ever meet
// zero weights).  Waves without such a lane run the merged-load instantiation.
template <typename C, bool CAREFUL>
__device__ __forceinline__ void load_span(__amdgpu_buffer_rsrc_t rsrc, int32_t off, uint32_t *raw) {
    const uint32_t voff = uint32_t(off);
    if constexpr (C::WORDS) {
        if constexpr (CAREFUL) {
#pragma unroll
            for (int i = 0; i < C::NW; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, uint32_t(max(off + 4 * i, 0)), 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < C::NW; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff + 4 * i, 0, 0);
        }
    } else if constexpr (C::BPP == 1) {
#pragma unroll
        for (int i = 0; i < C::SPAN; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b8(rsrc, voff + i, 0, 0);
    } else if constexpr (C::BPP == 2) {
#pragma unroll
        for (int i = 0; i < C::SPAN; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b16(rsrc, voff + 2 * i, 0, 0);
    } else if constexpr (C::BPP == 3) {
#pragma unroll
        for (int i = 0; i < C::SPAN * 3; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b8(rsrc, voff + i, 0, 0);
    } else {  // 6 bytes: three u16
#pragma unroll
        for (int i = 0; i < C::SPAN * 3; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b16(rsrc, voff + 2 * i, 0, 0);
    }
}

// float value(s) of span pixel q; SCALE = multiply by 1/max (reference unpack) or leave the integer value
template <typename C, int CH, bool S16, bool SCALE>
__device__ __forceinline__ void unpack_span_px(const uint32_t *raw, int q, float *u) {
    uint32_t s[CH];
    if constexpr (CH == 3) { s[0] = raw[3 * q]; s[1] = raw[3 * q + 1]; s[2] = raw[3 * q + 2]; }
    else if constexpr (C::BPP == 1 || (C::BPP == 2 && S16)) { s[0] = raw[q]; }
    else if constexpr (C::BPP == 2) { s[0] = raw[q] & 0xffu; s[1] = raw[q] >> 8; }
    else if constexpr (C::BPP == 4 && !S16) {
        s[0] = raw[q] & 0xffu; s[1] = (raw[q] >> 8) & 0xffu; s[2] = (raw[q] >> 16) & 0xffu; s[3] = raw[q] >> 24;
    } else if constexpr (C::BPP == 4) { s[0] = raw[q] & 0xffffu; s[1] = raw[q] >> 16; }
    else { s[0] = raw[2 * q] & 0xffffu; s[1] = raw[2 * q] >> 16; s[2] = raw[2 * q + 1] & 0xffffu; s[3] = raw[2 * q + 1] >> 16; }
#pragma unroll
    for (int c = 0; c < CH; ++c) u[c] = SCALE ? unpack_sample<S16>(s[c]) : float(s[c]);
}

// pack CH floats into the little-endian bytes of one pixel, returned in up to two dwords
template <int CH, bool S16, bool FMA>
__device__ __forceinline__ void pack_px_words(const float *v, uint32_t *w) {
    if constexpr (FMA && !S16) {
        // round-to-nearest-even + saturate in one instruction per channel
        uint32_t r = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) r = __builtin_amdgcn_cvt_pk_u8_f32(v[c] * 255.0f, c, r);
        w[0] = r;
    } else if constexpr (FMA && S16) {
        // v_cvt_pknorm_u16_f32: saturating round(v * 65535) for two channels at once
#pragma unroll
        for (int c = 0; c < CH; c += 2) {
            auto pr = __builtin_amdgcn_cvt_pknorm_u16(v[c], (c + 1 < CH) ? v[c + 1] : 0.0f);
            w[c / 2] = uint32_t(pr[0]) | (uint32_t(pr[1]) << 16);
        }
    } else {
        uint32_t q[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) q[c] = pack_sample<S16>(v[c]);
        if constexpr (!S16) {
            w[0] = q[0];
            if constexpr (CH >= 2) w[0] |= q[1] << 8;
            if constexpr (CH >= 3) w[0] |= q[2] << 16;
            if constexpr (CH >= 4) w[0] |= q[3] << 24;
        } else {
            w[0] = q[0];
            if constexpr (CH >= 2) w[0] |= q[1] << 16;
            if constexpr (CH >= 3) w[1] = q[2];
            if constexpr (CH >= 4) w[1] |= q[3] << 16;
        }
    }
}

template <int CH, bool S16>
__device__ __forceinline__ void store_px_words(__amdgpu_buffer_rsrc_t rsrc, uint32_t off, const uint32_t *w) {
    constexpr int BPP = CH * Sample<S16>::kBytes;
    if constexpr (BPP == 1) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)w[0], rsrc, off, 0, 0);
    else if constexpr (BPP == 2) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)w[0], rsrc, off, 0, 0);
    else if constexpr (BPP == 3) {
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)w[0], rsrc, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(w[0] >> 8), rsrc, off + 1, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(w[0] >> 16), rsrc, off + 2, 0, 0);
    } else if constexpr (BPP == 4) __builtin_amdgcn_raw_buffer_store_b32(w[0], rsrc, off, 0, 0);
    else if constexpr (BPP == 6) {
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)w[0], rsrc, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(w[0] >> 16), rsrc, off + 2, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)w[1], rsrc, off + 4, 0, 0);
    } else {
        u32x2 o = {w[0], w[1]};
        __builtin_amdgcn_raw_buffer_store_b64(o, rsrc, off, 0, 0);
    }
}

template <int CH, bool S16, int P, int S, int TX, int TYB, bool FMA>
__global__ __launch_bounds__(256) void resize_span_kernel(FusedParams p) {
    using C = SpanCfg<CH, S16, P, S, TX>;
    using Slot = typename RingSlot<CH>::type;
    constexpr int SF = RingSlot<CH>::kFloats;
    constexpr int D = C::D;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp0 = (blockIdx.x * 4 + wave) * kWave;  // first lane-group of this wave
    if (grp0 * P >= p.dst_w) return;  // wave-uniform: no columns for this wave (kernel has no barriers)
    const int grp = min(grp0 + lane, p.x_groups - 1);  // surplus lanes duplicate the last group, never store
    const int x0 = (grp0 + lane) * P;                  // first output column of this lane

    const int img = blockIdx.z;
    const uint8_t *sbase = p.src_tab ? static_cast<const uint8_t *>(p.src_tab[img]) : p.src + img * p.src_pitch;
    uint8_t *dbase = p.dst_tab ? static_cast<uint8_t *>(p.dst_tab[img]) : p.dst + img * p.dst_pitch;
    // descriptors cover exactly one image: anything outside reads 0 / is not written
    const __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(sbase), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dst_rs = __builtin_amdgcn_make_buffer_rsrc(dbase, 0, p.dst_bytes, 0x00020000);

    const int y0 = blockIdx.y * p.rows_per_chunk;
    const int y1 = min(p.dst_h, y0 + p.rows_per_chunk);
    if (y0 >= y1) return;

    // lane-resident horizontal taps: wx[p][j] multiplies span pixel S*p + j
    float wx[P][TX];
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
        for (int j = 0; j < TX; ++j) wx[q][j] = p.x_weight[(size_t)j * p.xw_stride + grp * P + q];
    const int32_t lane_off = p.x_left[grp] * C::BPP;  // may be negative at the left edge (zero-weight taps)
    const bool edge_wave = P > 1 && __any(lane_off < 0);

    Slot *ring = reinterpret_cast<Slot *>(lds_raw) + (size_t)wave * p.ring_slots * (kWave * P) + lane;

    // All per-output-row data is wave-uniform and comes through the scalar cache (SMEM): the row's
    // {left, need, count}, its ring slots and its weights.  Nothing but the source rows uses vector
    // memory loads, so the compiler can count the prefetched rows exactly (s_waitcnt vmcnt(N)).
    const_i32_ptr yinfo = (const_i32_ptr)(p.y_info);
    const_i32_ptr yslot = (const_i32_ptr)(p.y_slot);
    const_i32_ptr ywgt = (const_i32_ptr)(p.y_weight);
    const int r0 = yinfo[4 * y0];            // first source row this chunk touches
    const int r1 = yinfo[4 * (y1 - 1) + 1];  // last
    int slot = r0 % p.ring_slots;
    int y = y0;
    int cur_need = yinfo[4 * y0 + 1];
    int cur_cnt = yinfo[4 * y0 + 2];
    // TYB == 8: the pending output row's slots and weights are fetched one output row ahead
    // (SMEM latency would otherwise sit on the critical path of every vertical pass)
    int psl[8], pwb[8];
    if constexpr (TYB == 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { psl[j] = yslot[(size_t)y0 * 8 + j]; pwb[j] = ywgt[(size_t)y0 * 8 + j]; }
    }

    // Source rows are double-buffered in two register sets of D rows each: while one set is being
    // resampled the loads of the next D rows are already in flight into the other, so every
    // s_waitcnt the compiler places leaves a full group of loads outstanding.
    auto march = [&](auto careful_c) __attribute__((always_inline)) {
    constexpr bool CAREFUL = decltype(careful_c)::value;
    uint32_t rawA[D][C::NW], rawB[D][C::NW];
#pragma unroll
    for (int d = 0; d < D; ++d) load_span<C, CAREFUL>(src_rs, min(r0 + d, r1) * p.src_stride + lane_off, rawA[d]);

    // One group = D consecutive source rows, no per-row conditionals; rows past r1 in the last group
    // are ghosts (they re-read row r1 and land in ring slots nothing reads any more).
    auto process_group = [&](uint32_t (&cur)[D][C::NW], uint32_t (&nxt)[D][C::NW], int r) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < D; ++d) load_span<C, CAREFUL>(src_rs, min(r + D + d, r1) * p.src_stride + lane_off, nxt[d]);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int row = r + d;
            // ---- horizontal pass of source row `row` (src/resize.cc:110-117) ----
            float acc[P][SF];
#pragma unroll
            for (int q = 0; q < C::SPAN; ++q) {
                float u[CH];
                unpack_span_px<C, CH, S16, !FMA>(cur[d], q, u);
#pragma unroll
                for (int o = 0; o < P; ++o) {
                    const int j = q - S * o;  // tap index of span pixel q for output o
                    if (j >= 0 && j < TX) {
#pragma unroll
                        for (int c = 0; c < CH; ++c) {
                            if (j == 0) acc[o][c] = wx[o][0] * u[c];
                            else if constexpr (FMA) acc[o][c] = __builtin_fmaf(wx[o][j], u[c], acc[o][c]);
                            else acc[o][c] = acc[o][c] + wx[o][j] * u[c];
                        }
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < P; ++o) {
                Slot sv;
                if constexpr (SF == 1) sv = acc[o][0];
                else if constexpr (SF == 2) { sv.x = acc[o][0]; sv.y = acc[o][1]; }
                else { sv.x = acc[o][0]; sv.y = acc[o][1]; sv.z = acc[o][2]; sv.w = (CH == 4) ? acc[o][3] : 0.0f; }
                ring[(slot * P + o) * kWave] = sv;
            }
            slot = (slot + 1 == p.ring_slots) ? 0 : slot + 1;

            // ---- vertical pass for every output row completed by this row (src/resize.cc:122-132) ----
            while (cur_need <= row) {
                const_i32_ptr sy = yslot + (size_t)y * p.y_pad;
                const_i32_ptr wy = ywgt + (size_t)y * p.y_pad;
                float o[P][SF];
#pragma unroll
                for (int j0 = 0; j0 < TYB; j0 += 8) {
                    if (j0 >= cur_cnt) break;
                    int sl[8], wb[8];
                    if constexpr (TYB == 8) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { sl[j] = psl[j]; wb[j] = pwb[j]; }
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { sl[j] = sy[j0 + j]; wb[j] = wy[j0 + j]; }  // two s_load_dwordx8
                    }
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int j = j0 + jj;
                        if (j >= cur_cnt) break;
                        // weight into a VGPR: an SGPR operand would halve the rate of every multiply below
                        float w;
                        asm volatile("v_mov_b32 %0, %1" : "=v"(w) : "s"(wb[jj]));
#pragma unroll
                        for (int q = 0; q < P; ++q) {
                            const Slot v = ring[(sl[jj] * P + q) * kWave];
                            float vf[SF];
                            if constexpr (SF == 1) vf[0] = v;
                            else if constexpr (SF == 2) { vf[0] = v.x; vf[1] = v.y; }
                            else { vf[0] = v.x; vf[1] = v.y; vf[2] = v.z; vf[3] = v.w; }
#pragma unroll
                            for (int c = 0; c < CH; ++c) {
                                if (j == 0) o[q][c] = w * vf[c];
                                else if constexpr (FMA) o[q][c] = __builtin_fmaf(w, vf[c], o[q][c]);
                                else o[q][c] = o[q][c] + w * vf[c];
                            }
                        }
                    }
                }
                // pack + store this lane's P pixels of output row y
                const uint32_t doff = uint32_t(y * p.dst_stride + x0 * C::BPP);
                uint32_t pw[P][2];
#pragma unroll
                for (int q = 0; q < P; ++q) pack_px_words<CH, S16, FMA>(o[q], pw[q]);
                if constexpr (P == 2 && C::BPP == 4) {
                    if (x0 + 1 < p.dst_w) { u32x2 t = {pw[0][0], pw[1][0]}; __builtin_amdgcn_raw_buffer_store_b64(t, dst_rs, doff, 0, 0); }
                    else if (x0 < p.dst_w) __builtin_amdgcn_raw_buffer_store_b32(pw[0][0], dst_rs, doff, 0, 0);
                } else if constexpr (P == 4 && C::BPP == 4) {
                    if (x0 + 3 < p.dst_w) { u32x4 t = {pw[0][0], pw[1][0], pw[2][0], pw[3][0]}; __builtin_amdgcn_raw_buffer_store_b128(t, dst_rs, doff, 0, 0); }
                    else {
#pragma unroll
                        for (int q = 0; q < P; ++q) if (x0 + q < p.dst_w) __builtin_amdgcn_raw_buffer_store_b32(pw[q][0], dst_rs, doff + 4 * q, 0, 0);
                    }
                } else if constexpr (P == 2 && C::BPP == 8) {
                    if (x0 + 1 < p.dst_w) { u32x4 t = {pw[0][0], pw[0][1], pw[1][0], pw[1][1]}; __builtin_amdgcn_raw_buffer_store_b128(t, dst_rs, doff, 0, 0); }
                    else if (x0 < p.dst_w) store_px_words<CH, S16>(dst_rs, doff, pw[0]);
                } else {
#pragma unroll
                    for (int q = 0; q < P; ++q) if (x0 + q < p.dst_w) store_px_words<CH, S16>(dst_rs, doff + q * C::BPP, pw[q]);
                }
                ++y;
                if (y < y1) {
                    cur_need = yinfo[4 * y + 1];
                    cur_cnt = yinfo[4 * y + 2];
                    if constexpr (TYB == 8) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { psl[j] = yslot[(size_t)y * 8 + j]; pwb[j] = ywgt[(size_t)y * 8 + j]; }
                    }
                } else {
                    cur_need = 0x7fffffff;
                }
            }
        }
    };

    for (int r = r0;;) {
        process_group(rawA, rawB, r);
        r += D;
        if (r > r1) break;
        process_group(rawB, rawA, r);
        r += D;
        if (r > r1) break;
    }
    };  // march
    if (edge_wave) march(std::true_type{});
    else march(std::false_type{});
}

// ---------------------------------------------------------------------------------------------
// Streaming resize kernel for uniform tap tables (integer scale factors): no LDS at all.
//
// Same horizontal scheme as the span kernel.  Vertically the tap table must be uniform too: output
// row y reads source rows SY*y + y_base + j, j < TY (host-verified; zero-weight padding at the image
// edges).  Then every resampled source row contributes to a fixed set of pending output rows, so
// instead of keeping a ring of resampled rows the kernel keeps NPEND = ceil(TY / SY) output
// accumulators in registers and folds each resampled row into them the moment it is produced.
// Rows arrive in ascending order, so every output still sums its taps in ascending source order
// exactly as the reference does (src/resize.cc:125-129) — bit-exact in the non-FMA flavour.
// "Step" m = the SY source rows SY*m + y_base + ph (ph < SY); row ph of step m is tap SY*i + ph of
// output m - i.  Output m - ICLOSE is complete after row PHCLOSE of step m.
// Rows outside the image read as zeros through the buffer descriptor (their weights are zero).
// ---------------------------------------------------------------------------------------------
constexpr int ce_gcd(int a, int b) { return b == 0 ? a : ce_gcd(b, a % b); }

template <int CH, bool S16, int P, int SX, int TX, int SY, int TY, bool FMA>
__global__ __launch_bounds__(256) void resize_stream_kernel(StreamParams p) {
    using C = SpanCfg<CH, S16, P, SX, TX>;
    constexpr int NPEND = (TY - 1) / SY + 1;
    constexpr int ICLOSE = (TY - 1) / SY, PHCLOSE = (TY - 1) % SY;
    constexpr int YW = (NPEND * SY + 3) / 4 * 4;           // floats per step in the step-weight table
    constexpr int UNR = NPEND * 2 / ce_gcd(NPEND, 2);      // lcm(NPEND, 2): accumulator roles x A/B buffers

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp0 = (blockIdx.x * 4 + wave) * kWave;
    if (grp0 * P >= p.dst_w) return;
    const int grp = min(grp0 + lane, p.x_groups - 1);
    const int x0 = (grp0 + lane) * P;

    const int img = blockIdx.z;
    const uint8_t *sbase = p.src_tab ? static_cast<const uint8_t *>(p.src_tab[img]) : p.src + img * p.src_pitch;
    uint8_t *dbase = p.dst_tab ? static_cast<uint8_t *>(p.dst_tab[img]) : p.dst + img * p.dst_pitch;
    const __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(sbase), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dst_rs = __builtin_amdgcn_make_buffer_rsrc(dbase, 0, p.dst_bytes, 0x00020000);

    const int y0 = blockIdx.y * p.rows_per_chunk;
    const int y1 = min(p.dst_h, y0 + p.rows_per_chunk);
    if (y0 >= y1) return;
    const int m_end = y1 - 1 + ICLOSE;  // last step that completes an output of this chunk

    float wx[P][TX];
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
        for (int j = 0; j < TX; ++j) wx[q][j] = p.x_weight[(size_t)j * p.xw_stride + grp * P + q];
    const int32_t lane_off = p.x_left[grp] * C::BPP;
    const bool edge_wave = P > 1 && __any(lane_off < 0);
    const_i32_ptr ystep = (const_i32_ptr)(p.y_step_weight);

    auto march = [&](auto careful_c) __attribute__((always_inline)) {
    constexpr bool CAREFUL = decltype(careful_c)::value;
    float acc[NPEND][P][CH];
    uint32_t rawA[SY][C::NW], rawB[SY][C::NW];

    auto load_step = [&](uint32_t (&t)[SY][C::NW], int m) __attribute__((always_inline)) {
#pragma unroll
        for (int ph = 0; ph < SY; ++ph) {
            const int row = SY * m + p.y_base + ph;
            // rows < 0 or >= src_h give offsets outside the image's descriptor: they read 0
            const int rr = (row >= p.src_h) ? p.src_h : row;  // keep row * stride far from int overflow
            load_span<C, CAREFUL>(src_rs, rr * p.src_stride + lane_off, t[ph]);
        }
    };

    auto step = [&](uint32_t (&cur)[SY][C::NW], uint32_t (&nxt)[SY][C::NW], int m, auto kc) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value;  // (m - y0) mod NPEND, static thanks to the unrolling
        load_step(nxt, m + 1);
        // this step's vertical weights: entry i*SY + ph = weight of output m - i for source row ph
        float wy[NPEND][SY];
        {
            int wb[YW];
#pragma unroll
            for (int t = 0; t < YW; ++t) wb[t] = ystep[(size_t)m * YW + t];
#pragma unroll
            for (int i = 0; i < NPEND; ++i)
#pragma unroll
                for (int ph = 0; ph < SY; ++ph)
                    asm volatile("v_mov_b32 %0, %1" : "=v"(wy[i][ph]) : "s"(wb[i * SY + ph]));  // VGPR operand: full VALU rate
        }
#pragma unroll
        for (int ph = 0; ph < SY; ++ph) {
            // ---- horizontal pass (src/resize.cc:110-117) ----
            float h[P][CH];
#pragma unroll
            for (int q = 0; q < C::SPAN; ++q) {
                float u[CH];
                unpack_span_px<C, CH, S16, !FMA>(cur[ph], q, u);
#pragma unroll
                for (int o = 0; o < P; ++o) {
                    const int j = q - SX * o;
                    if (j >= 0 && j < TX) {
#pragma unroll
                        for (int c = 0; c < CH; ++c) {
                            if (j == 0) h[o][c] = wx[o][0] * u[c];
                            else if constexpr (FMA) h[o][c] = __builtin_fmaf(wx[o][j], u[c], h[o][c]);
                            else h[o][c] = h[o][c] + wx[o][j] * u[c];
                        }
                    }
                }
            }
            // ---- fold into the pending outputs (src/resize.cc:125-129, one tap per output) ----
#pragma unroll
            for (int i = 0; i < NPEND; ++i) {
                const int j = SY * i + ph;
                if (j < TY) {
                    constexpr int dummy = 0; (void)dummy;
                    const int sl = (K - i + 4 * NPEND) % NPEND;
#pragma unroll
                    for (int o = 0; o < P; ++o)
#pragma unroll
                        for (int c = 0; c < CH; ++c) {
                            if (j == 0) acc[sl][o][c] = wy[i][ph] * h[o][c];
                            else if constexpr (FMA) acc[sl][o][c] = __builtin_fmaf(wy[i][ph], h[o][c], acc[sl][o][c]);
                            else acc[sl][o][c] = acc[sl][o][c] + wy[i][ph] * h[o][c];
                        }
                }
            }
            if (ph == PHCLOSE) {
                const int y = m - ICLOSE;
                if (y >= y0) {  // outputs of the previous chunk are produced there
                    const int sl = (K - ICLOSE + 4 * NPEND) % NPEND;
                    const uint32_t doff = uint32_t(y * p.dst_stride + x0 * C::BPP);
                    uint32_t pw[P][2];
#pragma unroll
                    for (int q = 0; q < P; ++q) pack_px_words<CH, S16, FMA>(acc[sl][q], pw[q]);
                    if constexpr (P == 2 && C::BPP == 4) {
                        if (x0 + 1 < p.dst_w) { u32x2 t = {pw[0][0], pw[1][0]}; __builtin_amdgcn_raw_buffer_store_b64(t, dst_rs, doff, 0, 0); }
                        else if (x0 < p.dst_w) __builtin_amdgcn_raw_buffer_store_b32(pw[0][0], dst_rs, doff, 0, 0);
                    } else if constexpr (P == 4 && C::BPP == 4) {
                        if (x0 + 3 < p.dst_w) { u32x4 t = {pw[0][0], pw[1][0], pw[2][0], pw[3][0]}; __builtin_amdgcn_raw_buffer_store_b128(t, dst_rs, doff, 0, 0); }
                        else {
#pragma unroll
                            for (int q = 0; q < P; ++q) if (x0 + q < p.dst_w) __builtin_amdgcn_raw_buffer_store_b32(pw[q][0], dst_rs, doff + 4 * q, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < P; ++q) if (x0 + q < p.dst_w) store_px_words<CH, S16>(dst_rs, doff + q * C::BPP, pw[q]);
                    }
                }
            }
        }
    };

    load_step(rawA, y0);
    for (int m = y0; m <= m_end; m += UNR) {
        // UNR steps, fully unrolled: accumulator roles and the A/B buffers are compile-time
        auto run = [&](auto self, auto kc) __attribute__((always_inline)) -> void {
            constexpr int k = decltype(kc)::value;
            if constexpr (k < UNR) {
                if (m + k <= m_end) {
                    if constexpr (k % 2 == 0) step(rawA, rawB, m + k, std::integral_constant<int, k % NPEND>{});
                    else step(rawB, rawA, m + k, std::integral_constant<int, k % NPEND>{});
                    self(self, std::integral_constant<int, k + 1>{});
                }
            }
        };
        run(run, std::integral_constant<int, 0>{});
    }
    };  // march
    if (edge_wave) march(std::true_type{});
    else march(std::false_type{});
}

// name of the instantiation the last launch used (diagnostics / bench JSON)
thread_local char g_kernel_name[96];

template <int CH, bool S16, int P, int SX, int TX, int SY, int TY>
hipError_t launch_stream_t(const StreamParams &p, hipStream_t s, const char **name) {
    dim3 grid((p.x_groups + 255) / 256, p.n_chunks, p.n_images);
    if (p.fma) resize_stream_kernel<CH, S16, P, SX, TX, SY, TY, true><<<grid, 256, 0, s>>>(p);
    else resize_stream_kernel<CH, S16, P, SX, TX, SY, TY, false><<<grid, 256, 0, s>>>(p);
    snprintf(g_kernel_name, sizeof g_kernel_name, "resize_stream<ch%d,%s,P%d,SX%d,TX%d,SY%d,TY%d,%s>", CH, S16 ? "u16" : "u8", P, SX,
             TX, SY, TY, p.fma ? "fma" : "exact");
    if (name) *name = g_kernel_name;
    return hipGetLastError();
}

// curated list of streaming instantiations: key = pixel, P, SX, TX, SY, TY
#define PICHA_STREAM_LIST(X) \
    X(1, 4, false, 1, 2, 5, 2, 5) X(1, 4, false, 2, 2, 5, 2, 5) X(1, 4, false, 4, 2, 5, 2, 5) \
    X(1, 4, false, 1, 2, 7, 2, 7) X(1, 4, false, 2, 2, 7, 2, 7)

bool stream_supported(int pixel, int per_lane, int sx, int tx, int sy, int ty) {
#define X(PIX, CH, S16, P, SX, TX, SY, TY) if (pixel == PIX && per_lane == P && sx == SX && tx == TX && sy == SY && ty == TY) return true;
    PICHA_STREAM_LIST(X)
#undef X
    return false;
}

hipError_t launch_resize_stream(int pixel, const StreamParams &p, hipStream_t s, const char **name) {
#define X(PIX, CH, S16, P, SX, TX, SY, TY) \
    if (pixel == PIX && p.x_per_lane == P && p.x_step == SX && p.x_taps == TX && p.y_step == SY && p.y_taps == TY) \
        return launch_stream_t<CH, S16, P, SX, TX, SY, TY>(p, s, name);
    PICHA_STREAM_LIST(X)
#undef X
    return hipErrorInvalidValue;
}

template <int CH, bool S16, int P, int S, int TX, int TYB, bool FMA>
hipError_t launch_span_t(const FusedParams &p, hipStream_t s, const char **name) {
    auto kern = resize_span_kernel<CH, S16, P, S, TX, TYB, FMA>;
    const size_t lds = (size_t)4 * p.ring_slots * kWave * P * sizeof(typename RingSlot<CH>::type);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    dim3 grid((p.x_groups + 255) / 256, p.n_chunks, p.n_images);
    kern<<<grid, 256, lds, s>>>(p);
    snprintf(g_kernel_name, sizeof g_kernel_name, "resize_span<ch%d,%s,P%d,S%d,TX%d,TY%d,%s>", CH, S16 ? "u16" : "u8", P, S, TX,
             TYB, FMA ? "fma" : "exact");
    if (name) *name = g_kernel_name;
    return hipGetLastError();
}

template <int CH, bool S16, int P, int S, int TX, bool FMA>
hipError_t launch_span_ty(const FusedParams &p, hipStream_t s, const char **name) {
    if (p.y_pad == 8) return launch_span_t<CH, S16, P, S, TX, 8, FMA>(p, s, name);
    if (p.y_pad == 16) return launch_span_t<CH, S16, P, S, TX, 16, FMA>(p, s, name);
    return launch_span_t<CH, S16, P, S, TX, 32, FMA>(p, s, name);
}

template <int CH, bool S16, int P, int S, int TX>
hipError_t launch_span_fma(const FusedParams &p, hipStream_t s, const char **name) {
    if (p.fma) return launch_span_ty<CH, S16, P, S, TX, true>(p, s, name);
    return launch_span_ty<CH, S16, P, S, TX, false>(p, s, name);
}

// P == 1: any tap table.  TX buckets; the host pads the weight table with zeros up to the bucket.
template <int CH, bool S16>
hipError_t launch_span_p1(const FusedParams &p, hipStream_t s, const char **name) {
    switch (p.x_taps) {
#ifndef PICHA_QUICK  // PICHA_QUICK: developer build that only instantiates the flagship tap counts
    case 1: return launch_span_fma<CH, S16, 1, 1, 1>(p, s, name);
    case 2: return launch_span_fma<CH, S16, 1, 1, 2>(p, s, name);
    case 3: return launch_span_fma<CH, S16, 1, 1, 3>(p, s, name);
    case 4: return launch_span_fma<CH, S16, 1, 1, 4>(p, s, name);
    case 6: return launch_span_fma<CH, S16, 1, 1, 6>(p, s, name);
#endif
    case 5: return launch_span_fma<CH, S16, 1, 1, 5>(p, s, name);
    case 7: return launch_span_fma<CH, S16, 1, 1, 7>(p, s, name);
#ifndef PICHA_QUICK
    case 8: return launch_span_fma<CH, S16, 1, 1, 8>(p, s, name);
    case 10: return launch_span_fma<CH, S16, 1, 1, 10>(p, s, name);
    case 12: return launch_span_fma<CH, S16, 1, 1, 12>(p, s, name);
    case 16: return launch_span_fma<CH, S16, 1, 1, 16>(p, s, name);
    case 20: return launch_span_fma<CH, S16, 1, 1, 20>(p, s, name);
    case 24: return launch_span_fma<CH, S16, 1, 1, 24>(p, s, name);
    case 32: return launch_span_fma<CH, S16, 1, 1, 32>(p, s, name);
#endif
    }
    return hipErrorInvalidValue;
}

// P > 1: uniform-stride tables only (host-verified); a curated list of (format, P, S, TX).
hipError_t launch_span_multi(int pixel, const FusedParams &p, hipStream_t s, const char **name) {
    const int key = p.x_per_lane * 10000 + p.x_step * 100 + p.x_taps;
    if (pixel == 1) {  // rgba
        switch (key) {
        case 20205: return launch_span_fma<4, false, 2, 2, 5>(p, s, name);
        case 20207: return launch_span_fma<4, false, 2, 2, 7>(p, s, name);
        case 40205: return launch_span_fma<4, false, 4, 2, 5>(p, s, name);
        case 20103: return launch_span_fma<4, false, 2, 1, 3>(p, s, name);
        case 20104: return launch_span_fma<4, false, 2, 1, 4>(p, s, name);
        }
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

bool stream_kernel_supported(int pixel, int per_lane, int sx, int tx, int sy, int ty) {
    return stream_supported(pixel, per_lane, sx, tx, sy, ty);
}
hipError_t launch_resize_stream_kernel(int pixel, const StreamParams &p, hipStream_t s, const char **name) {
    return launch_resize_stream(pixel, p, s, name);
}

namespace {
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

int fused_x_bucket(int x_taps) {
    static const int buckets[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20, 24, 32};
    for (int b : buckets)
        if (x_taps <= b) return b;
    return 0;
}

bool fused_multi_supported(int pixel, int per_lane, int step, int x_taps) {
    if (pixel != 1) return false;
    const int key = per_lane * 10000 + step * 100 + x_taps;
    return key == 20205 || key == 20207 || key == 40205 || key == 20103 || key == 20104;
}

bool fused_supported(int pixel, int x_taps, int y_taps, int ring_slots, int per_lane) {
    if (pixel < 0 || pixel > 7) return false;
    if (x_taps < 1 || fused_x_bucket(x_taps) == 0) return false;
    if (y_taps < 1 || y_taps > 32) return false;
    const int ch = pixel_channels(pixel);
    const size_t slot = (ch == 1 ? 4 : ch == 2 ? 8 : 16);
    return (size_t)4 * ring_slots * kWave * per_lane * slot <= 160 * 1024;
}

hipError_t launch_resize_fused(int pixel, const FusedParams &p, hipStream_t s, const char **name) {
    if (p.x_per_lane > 1) return launch_span_multi(pixel, p, s, name);
    switch (pixel) {
    case 1: return launch_span_p1<4, false>(p, s, name);
#ifndef PICHA_QUICK
    case 0: return launch_span_p1<3, false>(p, s, name);
    case 2: return launch_span_p1<1, false>(p, s, name);
    case 3: return launch_span_p1<2, false>(p, s, name);
    case 4: return launch_span_p1<1, true>(p, s, name);
    case 5: return launch_span_p1<2, true>(p, s, name);
    case 6: return launch_span_p1<3, true>(p, s, name);
    case 7: return launch_span_p1<4, true>(p, s, name);
#endif
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
