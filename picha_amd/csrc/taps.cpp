// taps.cpp — see taps.h.  Build with -ffp-contract=off (no fused multiply-add anywhere).
#include "taps.h"

#include <algorithm>
#include <cmath>

#include "../../include/picha_hip.h"

namespace picha_hip {
namespace {

// The six reconstruction kernels of src/resize.cc:200-260, wrapped the way
// ScaledFilter<F> (src/resize.cc:262-268) wraps them: support = s * base, value = F(v / s) / s.
class Kernel1D {
public:
    Kernel1D(int tag, float s) : tag_(tag), s_(s) {
        switch (tag) {
        case PICHA_FILTER_BOX: base_ = 0.5f; break;       // resize.cc:206
        case PICHA_FILTER_TRIANGLE: base_ = 1.0f; break;  // resize.cc:201
        default: base_ = 2.0f; break;                     // cubic / mitchel family / lanczos<2>
        }
        if (tag == PICHA_FILTER_CATMULROM) set_bc(0.0f, 0.5f);    // resize.cc:233-236
        if (tag == PICHA_FILTER_MITCHEL) set_bc(0.333f, 0.333f);  // resize.cc:238-241 (0.333f, not 1/3)
    }
    float support() const { return s_ * base_; }
    float operator()(float v) const { return base(v / s_) / s_; }

private:
    // Polynomial coefficients of the Mitchell-Netravali family, folded in float exactly as
    // the constant expressions of src/resize.cc:218-227 evaluate (left to right).
    void set_bc(float B, float C) {
        a3_ = (12 - 9 * B - 6 * C) / 6;
        a2_ = (-18 + 12 * B + 6 * C) / 6;
        a0_ = (6 - 2 * B) / 6;
        b3_ = (-B - 6 * C) / 6;
        b2_ = (6 * B + 30 * C) / 6;
        b1_ = (-12 * B - 48 * C) / 6;
        b0_ = (8 * B + 24 * C) / 6;
    }
    float base(float o) const {
        switch (tag_) {
        case PICHA_FILTER_CUBIC: {  // resize.cc:259
            float a = std::fabs(o);
            return 1.0f - a * a * (0.75f - 0.25f * a);
        }
        case PICHA_FILTER_LANCZOS: {  // resize.cc:249-252, A = 2
            float x = o * float(M_PI), x2 = x * x;
            return x2 == 0 ? 1.0f : 2 * std::sin(x) * std::sin(x / 2) / x2;
        }
        case PICHA_FILTER_CATMULROM:
        case PICHA_FILTER_MITCHEL: {  // resize.cc:213-230
            float x = std::fabs(o);
            if (x < 1) return a0_ + (x * x * (a2_ + x * a3_));
            return b0_ + (x * (b1_ + x * (b2_ + x * b3_)));
        }
        case PICHA_FILTER_BOX: return 1.0f;                    // resize.cc:207
        case PICHA_FILTER_TRIANGLE: return 1.0f - std::fabs(o);  // resize.cc:202
        }
        return 0.0f;
    }
    int tag_;
    float s_, base_;
    float a0_ = 0, a2_ = 0, a3_ = 0, b0_ = 0, b1_ = 0, b2_ = 0, b3_ = 0;
};

}  // namespace

int build_axis_taps(int filter, float filter_scale, int src_size, int dst_size, AxisTaps &t) {
    if (filter < 0 || filter >= PICHA_NUM_FILTERS) return PICHA_ERR_INVALID_FILTER;
    if (!(filter_scale > 0)) return PICHA_ERR_INVALID_FILTER_WIDTH;
    if (src_size <= 0 || dst_size <= 0) return PICHA_ERR_INVALID_DIMENSIONS;
    const Kernel1D k(filter, filter_scale);

    t = AxisTaps();
    t.filter = filter;
    t.filter_scale = filter_scale;
    t.src_size = src_size;
    t.dst_size = dst_size;
    t.left.resize(dst_size);
    t.count.resize(dst_size);
    t.need.resize(dst_size);

    // Geometry constants: src/resize.cc:22-24 (per axis) and :72-79.
    const float scale = src_size / float(dst_size);
    const float fscale = std::max(std::max(scale, 1.0f), 1.0f / k.support());
    const float fsupport = k.support() * fscale;
    const float iscale = 1.0f / fscale;
    t.ref_ring = int(std::ceil(2 * fsupport));

    // Pass 1: ranges and raw weights in the reference's order; the centre is ACCUMULATED
    // (center += scale), never recomputed (src/resize.cc:26-27).
    std::vector<float> packed;  // concatenated, like the reference's shared PixelContribs
    std::vector<size_t> start(dst_size);
    packed.reserve(size_t(dst_size) * size_t(t.ref_ring + 1));
    float center = float(0.5) * scale;
    t.first_row = int(std::max(float(0), std::ceil(center - fsupport)));  // resize.cc:100
    for (int i = 0; i < dst_size; ++i, center += scale) {
        int lo = int(std::max(float(0), std::ceil(center - fsupport)));
        int hi = int(std::min(float(src_size - 1), std::floor(center + fsupport)));
        while (lo < hi && k((center - lo) * iscale) == 0) ++lo;  // trim exact zeros, resize.cc:31-34
        while (hi > lo && k((center - hi) * iscale) == 0) --hi;
        start[i] = packed.size();
        float total = 0;
        for (int j = lo; j <= hi; ++j) {
            float w = k((center - float(j)) * iscale);
            packed.push_back(w);
            total += w;
        }
        if (!(total > 0)) return PICHA_ERR_DEGENERATE_TAPS;  // the reference asserts, resize.cc:45
        const float norm = float(1) / total;
        for (size_t q = start[i]; q < packed.size(); ++q) packed[q] *= norm;
        t.left[i] = lo;
        t.count[i] = hi - lo + 1;
        t.taps = std::max(t.taps, hi - lo + 1);
        t.need[i] = std::min(src_size - 1, int(center + fsupport));  // resize.cc:104
    }

    // Pass 2: padded row-major weights + the rows the vertical pass really reads.
    t.weight.assign(size_t(dst_size) * t.taps, 0.0f);
    t.eff.assign(size_t(dst_size) * t.taps, 0);
    for (int i = 0; i < dst_size; ++i) {
        const int n = t.count[i];
        for (int j = 0; j < n; ++j) {
            t.weight[size_t(i) * t.taps + j] = packed[start[i] + j];
            // The reference keeps resampled rows in ref_ring slots (slot = row % ref_ring) and has
            // resampled every row up to need[i] when it produces output i; slot c % ref_ring then
            // holds the LAST such row congruent to c.  (Only differs from c when n > ref_ring.)
            const int c = t.left[i] + j;
            t.eff[size_t(i) * t.taps + j] = c + ((t.need[i] - c) / t.ref_ring) * t.ref_ring;
        }
        for (int j = n; j < t.taps; ++j) t.eff[size_t(i) * t.taps + j] = t.eff[size_t(i) * t.taps];
        t.span = std::max(t.span, t.need[i] - t.left[i] + 1);
    }
    return PICHA_OK;
}

}  // namespace picha_hip
