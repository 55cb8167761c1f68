// kernels.h — launch interface between the C ABI (capi.cpp) and the gfx950 kernels (kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace picha_hip {

// ---- colour convert -------------------------------------------------------------------------
struct ConvertParams {
    const uint8_t *src;  // image 0, row 0
    uint8_t *dst;
    long long src_pitch;  // bytes between consecutive images of the batch (0 for n == 1)
    long long dst_pitch;
    const void *const *src_tab;  // optional per-image pointers (device array); overrides pitch
    void *const *dst_tab;
    int src_stride, dst_stride;
    int width, height;
    int n_images;
    float r, g, b;
    // wide kernel (16 / 8 pixels per thread, 16-byte accesses): every image base and both strides are 4-byte aligned and
    // one image spans less than 4 GB; src_bytes / dst_bytes = bytes of ONE image (buffer descriptor range)
    int wide_ok;
    unsigned src_bytes, dst_bytes;
    int cc_mode;  // option "cc_mode", snapshotted by the caller: 0 auto, 1 the 4-pixel kernel, 2 wide + nt stores, 3 store-shaped
};
// src_pixel != dst_pixel.  Returns hipSuccess or the launch error; *kernel_name = the kernel layout that was launched.
hipError_t launch_color_convert(int src_pixel, int dst_pixel, const ConvertParams &p, hipStream_t s, const char **kernel_name);

// ---- work items (r03): an explicit list of (column-strip block, output rows [y0, y1), image) per workgroup ---------
// Without it a launch is a 3-D grid of equal row chunks.  With it the host orders the work: the list is 8 sub-lists of
// `items_per_xcd` entries, one per XCD (workgroups are dealt to XCDs round-robin by linear id), each a contiguous piece of
// the batch in (image, rows, strip) order — neighbouring strips of the same rows run on one XCD at about the same time —
// whose LAST entries are short row chunks, so that the waves still running when the list runs dry are short ones.
//   queue == nullptr  one workgroup per item: item of workgroup id L = sub-list L % 8, entry L / 8
//   queue != nullptr  persistent launch: the grid is only as many workgroups as the chip holds at once; each takes items
//                     from its XCD's sub-list with an atomic counter (queue[xcd], zeroed before the launch) and steals
//                     from the other XCDs' sub-lists once its own is exhausted
struct WorkList {
    const int4 *items = nullptr;   // {strip block, y0, y1, image}; y0 == y1: padding
    int items_per_xcd = 0;
    unsigned *queue = nullptr;     // [8]
};

// ---- resize: two-pass kernels through an HBM scratch (any geometry) ---------------------------
struct TwoPassParams {
    const uint8_t *src;
    uint8_t *dst;
    int src_stride, dst_stride;
    int src_w, src_h, dst_w, dst_h;
    // x axis
    const int32_t *x_left, *x_count;
    const float *x_weight;  // [dst_w][x_taps]
    int x_taps;
    // y axis
    const int32_t *y_count, *y_eff;  // eff: [dst_h][y_taps]
    const float *y_weight;           // [dst_h][y_taps]
    int y_taps;
    int row0, nrows;  // source rows [row0, row0 + nrows) are resampled into scratch
    float *scratch;   // [nrows][dst_w][channels]
};
hipError_t launch_resize_twopass(int pixel, const TwoPassParams &p, hipStream_t s);

// ---- resize: fused single kernel (horizontal pass -> per-wave LDS ring -> vertical pass) -------
struct FusedParams {
    const uint8_t *src;
    uint8_t *dst;
    long long src_pitch, dst_pitch;
    const void *const *src_tab;
    void *const *dst_tab;
    int src_stride, dst_stride;
    int src_w, src_h, dst_w, dst_h;
    unsigned src_bytes, dst_bytes;  // bytes of ONE image (buffer descriptor range)
    int n_images;
    // x axis: a lane owns x_per_lane consecutive outputs ("group"); output q of a group uses the
    // span pixels [x_step*q, x_step*q + x_taps) of the span that starts at x_left[group]
    const int32_t *x_left;   // [x_groups] first source pixel of the group's span (may be < 0)
    const float *x_weight;   // tap-major [x_taps][xw_stride], zero padded
    int x_taps;              // kernel tap count (bucket >= table taps)
    int x_per_lane, x_step;
    int x_groups, xw_stride;
    int fma;                 // 0: reference arithmetic; 1: fused multiply-add flavour (weights pre-scaled)
    // y axis
    const int4 *y_info;      // [dst_h] {left, need, count, 0}
    const float *y_weight;   // [dst_h][y_pad] zero padded
    const int32_t *y_slot;   // [dst_h][y_pad] ring slot (eff % ring_slots); padded taps repeat tap 0
    int y_taps, y_pad;
    int ring_slots;
    int rows_per_chunk, n_chunks;
    int xcd_remap;  // 1: XCD-contiguous workgroup order (see kernels_impl.inc, xcd_contiguous_block)
    // accumulator kernel (resize_acc_kernel): per SOURCE row a record of 2*nacc dwords — the weight this row has in each
    // of the nacc pending outputs (output y lives in accumulator y % nacc) and, per accumulator, the output row that
    // COMPLETES with this source row (or -1); per output row {first source row needed by it or any later row, last}
    const int32_t *yrow_rec;    // [src_h + 8][2 * nacc]
    const int32_t *yrow_range;  // [dst_h][2]
    // ring aliasing of the one-row kind (optional, accumulator kernel only): the records and ranges then refer to positions in a
    // LIST of source rows — every row in order, plus, ahead of each aliased window, the window's last row once more, whose record
    // carries the aliased first tap's weight — and yrow_src[position] is the source row to resample (nullptr: position == row)
    const int32_t *yrow_src;    // [list length + 8] or nullptr
    int long_block;             // resize_acc_long_kernel: threads per workgroup (64 / 128 / 192 / 256)
    int nacc;
    int xch;  // accumulator kernel, neighbour-exchange form: a lane loads only its own x_step * x_per_lane pixels (uniform-base x table)
    WorkList work;  // accumulator kernel only: explicit work items / persistent launch
    int *resident_query;  // host pointer; set: launch nothing, report how many workgroups of this kernel the device holds at once
    // accumulator kernel, colour-convert epilogue (resize -> colorConvert in one kernel, r03): epi_on != 0: a finished pixel is
    // packed to the resize's own format, unpacked, channel-mapped and packed to epi_pixel at store time (dst_* describe THAT image)
    int epi_on, epi_pixel;
    float epi_r, epi_g, epi_b;
};
// Accumulator kernel: instantiations exist for x tap buckets >= 4 and nacc in {4, 6}.  Windows of 33..256 taps (kAccLongMaxTaps) go to the
// long-window variant, which walks the window in segments of acc_long_segment() pixels (64..96 bytes: measured best —
// 8-pixel segments cost +35 %, 32-pixel segments of rgba +19 %, of grey -27 %) plus an optional half segment at the end,
// so its tap count is a multiple of half a segment.
constexpr int acc_long_segment(int bytes_per_pixel) { return bytes_per_pixel <= 3 ? 32 : (bytes_per_pixel <= 6 ? 16 : 8); }
// ... with this many segment buffers per wave (r03, profiles/r03_ab_acc_long.txt: four in flight are 0-10 % faster than two for
// pixels of up to 4 bytes, 0-15 % slower for 8-byte pixels)
// Longest window of the long-window kernel: 256 taps = 64 KB of LDS weights per wave, two single-wave workgroups per CU (r03: was 128;
// beyond it the two-pass kernels remain)
constexpr int kAccLongMaxTaps = 256;
constexpr int acc_long_bufs(int bytes_per_pixel) { return bytes_per_pixel <= 4 ? 4 : 2; }
bool acc_supported(int pixel, int x_taps_bucket, int nacc);
bool acc_p2_supported(int pixel, int step, int taps, int nacc);  // two outputs per lane: window of the second `step` pixels after the first
bool acc_xch_supported(int pixel, int per_lane, int step, int taps, int nacc);  // neighbour-exchange form (dword-sized pixels)
bool acc_epi_supported(int pixel, int x_taps_bucket, int nacc);  // colour-convert epilogue: one output per lane, tap buckets 4 / 5 / 7 / 8 / 10 / 12 / 16
bool acc_up_supported(int pixel, int per_lane, int taps, int nacc);  // integer upscales: per_lane outputs share one window of `taps` pixels (step 0)
hipError_t launch_resize_acc(int pixel, const FusedParams &p, hipStream_t s, const char **kernel_name);
// True if a fused instantiation exists for this format / tap counts and its LDS ring fits.
bool fused_supported(int pixel, int x_taps, int y_taps, int ring_slots, int per_lane);
bool fused_multi_supported(int pixel, int per_lane, int step, int x_taps);
// Kernel tap count for a table with x_taps taps (0: none).  Padded row length of the y tables.
int fused_x_bucket(int x_taps, int pixel, bool acc = false);
// True if the kernel for that bucket fetches a sub-dword-pixel span as whole dwords: the host must then
// keep every lane's window inside its row (shifted, zero-padded in front).
bool fused_x_bytespan(int x_taps_bucket, int pixel);
// Sub-dword pixel formats (bytes per pixel 1, 2, 3, 6) in the single-output span kernels, two ways to fetch a window
// of `x_taps_bucket` pixels as whole dwords:
//   rotated   aligned dword loads from the dword holding the window's first byte, then v_alignbyte_b32 by the lane's
//             byte phase: any window start, costs one extra dword and ceil(bytes/4) rotate instructions per row;
//   aligned   the host moves each window's start down to a 4-byte boundary of the row (up to 3 extra tap slots for
//             1- and 3-byte pixels, 1 for 2- and 6-byte pixels) and pads the tap count so the window is whole dwords.
// Measured r01 (scripts/survey_formats.py): rotation wins for 1- and 3-byte pixels with short windows (rgb 3x down
// -16 %, rgb 2x up -49 %) and loses by ~10 % on long ones and on 2-/6-byte pixels, hence this rule.
constexpr bool span_rotate(int bytes_per_pixel, int x_taps_bucket) {
    return bytes_per_pixel % 4 == 0 ? false
         : bytes_per_pixel % 2 == 1 ? x_taps_bucket <= 20
                                    : (x_taps_bucket <= 3 || (x_taps_bucket * bytes_per_pixel) % 4 != 0);
}

// Which formats the span/stream kernels unpack through the denormal form in the reference flavour (then the x weights
// must be scaled by 2^22).  Measured r01 (profiles/r01_ab_denorm_unpack.txt): 16-bit samples −5.5 % kernel time (one
// v_mul_f32_sdwa instead of and/shift + cvt + mul), grey −3 %; 8-bit multi-channel formats are 4–10 % SLOWER that
// way (their v_cvt_f32_ubyteN overlaps with the multiply/add stream, an SDWA multiply does not), so they keep cvt + mul.
// (r04: 8-bit grey left this set.  An SDWA instruction costs the issue of three or four plain ones (profiles/r05_ubench_valu_issue.txt: any v_mul_f32_sdwa
//  in a multiply-add stream puts it at 4.0 cycles per instruction instead of 2.05) while v_cvt_f32_ubyteN overlaps with the multiply-add pipe: cvt + mul measured
//  -2.5 .. -7 % on the grey downscales (profiles/r04_ab_unpack_forms.txt).  16-bit samples keep the denormal form: and / shift + cvt + mul is 9 % slower on cfg5.)
constexpr bool exact_unpack_scaled(int channels, bool is16) { (void)channels; return is16; }
int fused_y_pad(int y_taps);
hipError_t launch_resize_fused(int pixel, const FusedParams &p, hipStream_t s, const char **kernel_name);

// ---- resize: streaming kernel for uniform tap tables (no LDS) --------------------------------------
struct StreamParams {
    const uint8_t *src;
    uint8_t *dst;
    long long src_pitch, dst_pitch;
    const void *const *src_tab;
    void *const *dst_tab;
    int src_stride, dst_stride;
    int src_w, src_h, dst_w, dst_h;
    unsigned src_bytes, dst_bytes;
    int n_images;
    const int32_t *x_left;
    const float *x_weight;
    int x_taps, x_per_lane, x_step, x_groups, xw_stride;
    int fma;
    // output row y reads source rows y_step*y + y_base + j, j < y_taps
    const float *y_step_weight;  // [dst_h + (y_taps-1)/y_step][YW]; YW = roundup4(ceil(y_taps/y_step)*y_step + 1);
                                 // last dword of a record: int flags, bit 0 = weights identical to the previous record, bit 1 = canonical step
    int y_step, y_taps, y_base;
    int rot;  // 1: row-granular variant with rotating accumulators; 0: step-granular, fully unrolled roles
    int xcd_remap;  // 1: permute workgroup ids so each XCD gets a contiguous range of (strip, chunk, image)
    int dpp;  // 1: cross-lane pixel-sharing variant (x tables: per_lane 1, uniform base step*x + L0, L0 == -1)
    int rows_per_chunk, n_chunks;
    // resize_sym_kernel (kernels_sym.hip): uniform AND symmetric tap tables.  Outputs [cx0, cx1) have the canonical weights
    // (x_canon[min(j, T-1-j)], float bits) and the window start x_step*x + x_base; the step records of y_step_weight carry
    // bit 1 in their flag dword when every output row of the step is canonical (weights y_canon[...]).
    int32_t x_canon[8], y_canon[8];
    int cx0, cx1, x_base;
    int block_waves;  // waves per workgroup (1..4)
    int sync_steps;   // 1: one s_barrier per step keeps a workgroup's waves in lockstep (shared edge lines hit in L1)
    int xch;          // 1: neighbour-exchange variant (a lane loads only its own x_step * x_per_lane pixels; uniform-base x table)
    WorkList work;    // resize_sym_kernel only: explicit work items / persistent launch
    // resize_sym_kernel, colour-convert epilogue (r04; as FusedParams::epi_*): dst_* describe the converted image
    int epi_on, epi_pixel;
    float epi_r, epi_g, epi_b;
    int *resident_query;   // (r05, resize_stream_kernel) host pointer; set: launch nothing, report how many workgroups of this kernel the device holds at once
};
bool stream_kernel_supported(int pixel, int per_lane, int sx, int tx, int sy, int ty, int rot);
hipError_t launch_resize_stream_kernel(int pixel, const StreamParams &p, hipStream_t s, const char **kernel_name);
int stream_kernel_resident_blocks(int pixel, const StreamParams &p);   // workgroups of the chosen instantiation the device holds at once (cached per shape)
bool stream_dpp_supported(int pixel, int sx, int tx, int sy, int ty);
bool sym_kernel_supported(int pixel, int per_lane, int sx, int tx, int sy, int ty, int xch = 0);
bool sym_epi_supported(int pixel, int per_lane, int sx, int tx, int sy, int ty);   // colour-convert epilogue instantiation
hipError_t launch_resize_sym_kernel(int pixel, const StreamParams &p, hipStream_t s, const char **kernel_name);
// workgroups of the persistent form the device holds at once for the configuration in p (0: no such instantiation)
int sym_kernel_resident_blocks(int pixel, const StreamParams &p);
int acc_kernel_resident_blocks(int pixel, const FusedParams &p);  // 0: unknown / unsupported

// ---- resize: row-staged kernel (kernels_rows.hip): pixels converted once per row into LDS, optional fused colour convert ----
struct RowsParams {
    const uint8_t *src;
    uint8_t *dst;
    long long src_pitch, dst_pitch;
    const void *const *src_tab;
    void *const *dst_tab;
    int src_stride, dst_stride;
    int src_w, src_h, dst_w, dst_h;
    unsigned src_bytes, dst_bytes;
    int n_images;
    // x axis: thread t of strip b owns outputs (b*256 + t)*x_per_lane ..; its window starts at source pixel x_left[group]
    // (aligned for the LDS reads), the second output's x_step pixels later; weights tap-major [x_taps][xw_stride]
    const int32_t *x_left;
    const float *x_weight;
    const int32_t *strip_left;   // [n_strips] first source pixel a strip stages (a multiple of the stage piece)
    int x_taps, x_per_lane, x_step, x_groups, xw_stride;
    int n_strips, stage_pieces;  // stage pieces per thread and row: 1, 2, 4 or 8 (x 256 threads x pixels per piece)
    int fma;
    float r, g, b;               // luma weights of the fused colour convert
    // y axis: the accumulator kernel's per-source-row records
    const int32_t *yrow_rec, *yrow_range;
    int nacc;
    int rows_per_chunk, n_chunks;
};
// variant: 0 grey, 1 rgba, 2 rgb -> grey, 3 rgba -> grey, 4 greya, 5 r16g16b16a16, 6 rgb -> greya, 7 rgba -> greya, 8 r16g16b16 -> r16,
// 9 r16g16b16a16 -> r16, 10 r16g16b16 -> grey, 11 r16g16b16a16 -> grey (rows_variant(): -1 if none)
int rows_variant(int src_pixel, int dst_pixel);
int rows_stage_pixels(int variant);   // source pixels per stage piece
int rows_window_align(int variant);   // window starts are multiples of this many pixels (8-byte LDS reads)
bool rows_kernel_supported(int variant, int per_lane, int step, int taps);
hipError_t launch_resize_rows_kernel(int variant, const RowsParams &p, hipStream_t s, const char **kernel_name);

// ---- resize: run kernel (kernels_run.hip, r03): periodic tap tables, a lane resamples a run of consecutive outputs ----------
struct RunParams {
    const uint8_t *src;
    uint8_t *dst;
    long long src_pitch, dst_pitch;
    const void *const *src_tab;
    void *const *dst_tab;
    int src_stride, dst_stride;
    int src_w, src_h, dst_w, dst_h;
    unsigned src_bytes, dst_bytes;
    int n_images;
    // x axis: the table repeats every `period` outputs / `period_px` source pixels; a lane owns x_per_lane (a multiple of the
    // period) consecutive outputs; output x's window of x_taps pixels starts at x_base + (x / period) * period_px +
    // ((x % period) * period_px + off_ceil * (period - 1)) / period; x_canon = the period's weights [period][x_taps] in the kernel's flavour (plain,
    // 1/max folded for fma, or scaled by 2^22 where the format unpacks through the denormal form)
    int period, period_px, x_taps, x_per_lane, x_base, x_groups, off_ceil;
    const float *x_canon;
    int fma;
    const int32_t *yrow_rec, *yrow_range;  // the accumulator kernel's per-source-row records
    const int32_t *yrow_src;               // (r05, shared-unpack form) ring-aliased tables: records / ranges are positions in this LIST of source rows (FusedParams::yrow_src); nullptr: position == row
    int nacc;
    int rows_per_chunk, n_chunks;
    int block_waves;  // waves per workgroup (1..4): adjacent column strips of the same rows
    WorkList work;
    int *resident_query;
    // columns outside [cx0, cx1) have windows that an image edge clips (other weights): resize_edges_kernel recomputes them
    // from the generic tables below after the march
    int cx0, cx1;
    // ... reading the source pixels from a stash the march writes: per (image, side, source row) the stash_words raw dwords of the
    // lane that holds the side's columns; that lane's window starts stash_org[side] bytes into the row (a multiple of 4)
    uint32_t *stash;       // [n_images][2][src_h][stash_words]; nullptr: no edge columns
    int stash_words, stash_img0;
    int stash_org[2];
    int edge_rows;  // output rows per wave of that kernel (<= 64; its tables and horizontal sums must fit 8 KB of LDS: kernels_run.hip)
    const int32_t *x_left, *x_count;
    const float *x_weight;  // [dst_w][x_wtaps], reference weights
    int x_wtaps;
    const int32_t *y_count, *y_eff;
    const float *y_weight;  // [dst_h][y_taps]
    int y_taps;
    const int32_t *y_left;  // [dst_h] (resize_ring_kernel)
    const int4 *y_info;     // [dst_h] {., need, ., .}: the last source row the reference has resampled when it produces the row
    int ring_rows;          // resize_ring_kernel: rows of its register ring (the reference's ring where the table is aliased)
    // r04 forms of the run kernel (kernels_run.hip)
    int xch;         // 1: neighbour exchange — a lane loads only its own (x_per_lane / period) * period_px pixels (dword-sized pixels; x_base <= 0 is minus the window pixels in front of them)
    int wsgpr;       // phases of the period whose weights stay in SGPRs (0 .. period)
    int edge_image;  // 1: resize_edges_kernel reads the edge columns' windows from the image, not from a stash
    int edge_in;     // 1 (r05, resize_ring_kernel shapes with up_edges_inside()): the march takes the clipped columns' own weights and produces them itself — no edge kernel
    // (r05, third session) the edge blocks INSIDE the march's grid: per (row chunk, image) two more workgroups at blockIdx.x >= edge_bx0 (one wave each, one per side)
    // run the edge pass over the chunk's rows reading the image, next in dispatch order to the march's workgroups of the same rows; the march leaves the clipped
    // columns alone (its side lanes store pixel by pixel).  Plain grids only; shapes: run_edges_fused() in kernels_run.hip
    int edge_fused, edge_bx0;
    int edge_lds_floats;   // LDS floats a block of the edge kernel needs (host-computed for edge_rows; 0: kEdgeFloats) — smaller blocks, more of them resident
    int phase_shift; // > 0 (neighbour-exchange shapes): time-phased stores — finished rows wait in registers until s_memrealtime >> phase_shift changes
    // colour convert in front of an integer upscale (r04, resize_ring_kernel / resize_edges_kernel): the source images are in cv_src_pixel, every window
    // pixel goes through src/colorconvert.cc:136-152 (unpack, channel map, pack = the intermediate image's quantised samples) in registers and is then
    // unpacked as the kernel's own format; no stash (edge_image = 1)
    int cv_on, cv_src_pixel;
    float cv_r, cv_g, cv_b;
};
// the in-grid edge blocks are a measurement build (-DPICHA_RUN_EFUSE): measured slower than the edge kernel in the reference flavour (profiles/r05_ab_edge_blocks.txt)
#ifdef PICHA_RUN_EFUSE
constexpr bool kRunEfuseBuilt = true;
#else
constexpr bool kRunEfuseBuilt = false;
#endif
bool run_kernel_supported(int pixel, int period, int period_px, int taps, int per_lane, int nacc, int off_ceil, int xhl = -1, int wsgpr = 0);
hipError_t launch_resize_run_kernel(int pixel, const RunParams &p, hipStream_t s, const char **kernel_name);
// ring-in-registers kernel (integer upscales: period = the factor, period_px = 1, ring 4; ring-aliased downscales: ring = the
// reference's): yrow_rec unused; yrow_range: per output row {first source row needed by it or any later row, last row so far}
bool up_kernel_supported(int pixel, int period, int period_px, int taps, int per_lane, int off_ceil, int ring_rows, int y_taps);
constexpr bool run_xch_edges_inside(int taps, int per_lane) { return per_lane * taps <= 32; }   // == run_edges_inside for the neighbour-exchange forms (kernels_run.hip)
constexpr bool up_edges_inside(int period, int period_px, int taps, int per_lane) { return period > period_px && per_lane * taps <= 32; }   // == ring_edges_inside (kernels_run.hip)
hipError_t launch_resize_up_kernel(int pixel, const RunParams &p, hipStream_t s, const char **kernel_name);
bool up_cv_kernel_supported(int pixel, int src_pixel, int period, int period_px, int taps, int per_lane, int off_ceil, int ring_rows, int y_taps);  // p.cv_on forms
hipError_t launch_resize_edges_kernel(int pixel, const RunParams &p, hipStream_t s);  // after the run kernel, same stream: the columns outside [cx0, cx1)
int run_kernel_resident_blocks(int pixel, const RunParams &p);

// ---- Image helpers on device images (lib/image.js:57-73) -------------------------------------------
struct CompareParams {
    const void *const *a_tab;  // per-image row-0 pointers (device arrays)
    const void *const *b_tab;
    int a_stride, b_stride;
    int row_bytes, height;     // payload bytes per row (width * bytes per pixel)
    int rows_per_block;
    int n_images;
    unsigned long long *out;   // [n_images][2]: sum |a - b| over payload bytes, number of differing bytes (pre-zeroed)
};
hipError_t launch_image_compare(const CompareParams &p, hipStream_t s);

// ---- synthetic pixels ---------------------------------------------------------------------------
hipError_t launch_fill_random(void *dev, size_t bytes, uint64_t seed, int sample_bytes, hipStream_t s);
void fill_random_host(void *host, size_t bytes, uint64_t seed, int sample_bytes);

}  // namespace picha_hip
