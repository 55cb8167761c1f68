// taps.h — host-side filter tap tables for the MI355X resize kernels.
//
// Product code (part of libpicha_hip.so).  Builds, per axis, the table the reference builds in
// makeContribs (src/resize.cc:19-50) for ScaledFilter<F>(filterScale) (src/resize.cc:262-268),
// plus what the device kernels need on top of it: the vertical pass's "needed row" per output
// (src/resize.cc:104) and the ring-slot aliasing of the reference's FloatBuffer
// (src/resize.cc:83,108,126) resolved into explicit source-row indices.
//
// All arithmetic is IEEE binary32 in the reference's evaluation order; this file must be built
// with -ffp-contract=off.  Tap weights are ONLY computed here, on the host (lanczos needs libm
// sinf, src/resize.cc:249-252); device code only multiplies and adds.
#pragma once

#include <cstdint>
#include <vector>

namespace picha_hip {

struct AxisTaps {
    int filter = 0;
    float filter_scale = 0;
    int src_size = 0, dst_size = 0;
    int taps = 0;                // largest tap count over all outputs
    std::vector<int32_t> left;   // [dst] first contributing source index
    std::vector<int32_t> count;  // [dst] number of taps (right - left + 1)
    std::vector<float> weight;   // [dst][taps] row-major, zero padded, normalised

    // Vertical-role data (meaningful when the axis is used as the row axis).
    int ref_ring = 0;            // maxycontrib = int(ceil(2 * fsupport))        resize.cc:79
    int first_row = 0;           // first source row the reference resamples     resize.cc:100
    std::vector<int32_t> need;   // [dst] last source row resampled before output i is produced, resize.cc:104
    std::vector<int32_t> eff;    // [dst][taps] source row actually read for tap j once the
                                 // reference's ring aliasing (slot = c % ref_ring) is applied;
                                 // == left+j whenever count <= ref_ring.  Padded taps repeat tap 0.
    int span = 0;                // max over outputs of need[i] - left[i] + 1 (device ring size)
};

// Returns 0, or a picha_status (<0): PICHA_ERR_INVALID_FILTER, PICHA_ERR_DEGENERATE_TAPS.
int build_axis_taps(int filter, float filter_scale, int src_size, int dst_size, AxisTaps &out);

}  // namespace picha_hip
