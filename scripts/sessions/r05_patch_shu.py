# one-off editing aid of round 5 (kept as a record): inserts the shared-unpack form (XHL == -2) into kernels_run.hip
import sys
p = 'picha_amd/csrc/kernels_run.hip'
s = open(p).read()

def rep(old, new, cnt=1):
    global s
    if old not in s:
        sys.exit('missing anchor: ' + old[:80])
    s = s.replace(old, new, cnt)

# --- the struct ---
rep("""// PH (r04): TIME-PHASED STORES.""", """// XHL == -2 (r05): SHARED UNPACK for 8-bit sub-dword pixels (grey, greya).  In the plain form a lane converts its whole window — at 7.5x / 30 taps 53
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
    static constexpr int HR = SPAN - OWN;            // window pixels held by lane + 1
    static constexpr int NF = HR * CH;               // floats taken from the neighbour = bytes of the wave's halo
    static constexpr int NFC = (NF + 3) / 4;         // ... in 16-byte chunks
    static constexpr int LANES = kWave + 1;          // slots per chunk: 64 lanes + the virtual lane behind the wave
    static constexpr bool OK = !S16 && BPP < 4 && HR > 0 && HR <= OWN && NF <= kWave;
};
// PH (r04): TIME-PHASED STORES.""")

# --- kernel prologue ---
rep("""    constexpr bool XCH = XHL >= 0;
    static_assert(P % Q == 0, "a lane owns whole periods");""", """    constexpr bool XCH = XHL >= 0;
    constexpr bool SHU = XHL == -2;
    using S = RunShu<CH, S16, Q, PX, TX, P, RC>;
    using CO = EdgeCfg<CH, S16, S::OWN>;   // the lane's own pixels as a byte span (shared-unpack form)
    static_assert(!SHU || (S::OK && !PH), "shared unpack: 8-bit sub-dword pixels, the neighbour's share within its own pixels");
    static_assert(P % Q == 0, "a lane owns whole periods");""")
rep("""    constexpr int NRAW = XCH ? X::NOWN + 1 : C::NW;""", """    constexpr int NRAW = XCH ? X::NOWN + 1 : SHU ? CO::NW + 1 : C::NW;""")

# halo offset + loads
rep("""    __shared__ uint32_t halo_lds[XCH ? 4 * kWave : 1];
""", """    __shared__ uint32_t halo_lds[XCH ? 4 * kWave : 1];
    // shared unpack: lane k fetches sample k of the NF behind the wave's run (the other lanes repeat the last one)
    [[maybe_unused]] const int32_t shu_halo_off = (p.x_base + (grp0 + kWave) * S::OWN) * C::BPP + sadj + min(lane, S::NF - 1);
""")
rep("""            t[X::NOWN] = __builtin_amdgcn_raw_buffer_load_b32(src_rs, uint32_t(row * p.src_stride + halo_off), 0, 0);
        } else {""", """            t[X::NOWN] = __builtin_amdgcn_raw_buffer_load_b32(src_rs, uint32_t(row * p.src_stride + halo_off), 0, 0);
        } else if constexpr (SHU) {
            load_span<CO, false>(src_rs, row * p.src_stride + lane_off, t);
            t[CO::NW] = __builtin_amdgcn_raw_buffer_load_b8(src_rs, uint32_t(row * p.src_stride + shu_halo_off), 0, 0);
        } else {""")
rep("""        if constexpr (XCH) { load_row(raw[k], row); }
        else {
            const int32_t off = row * p.src_stride + lane_off;
            if (careful && row <= 0) load_span<C, true>(src_rs, off, raw[k]);
            else load_span<C, false>(src_rs, off, raw[k]);
        }""", """        if constexpr (XCH) { load_row(raw[k], row); }
        else if constexpr (SHU) {
            if (careful && row <= 0) {
                load_span<CO, true>(src_rs, row * p.src_stride + lane_off, raw[k]);
                raw[k][CO::NW] = __builtin_amdgcn_raw_buffer_load_b8(src_rs, uint32_t(row * p.src_stride + shu_halo_off), 0, 0);
            } else load_row(raw[k], row);
        } else {
            const int32_t off = row * p.src_stride + lane_off;
            if (careful && row <= 0) load_span<C, true>(src_rs, off, raw[k]);
            else load_span<C, false>(src_rs, off, raw[k]);
        }""")

# stash: never with SHU
rep("""        if constexpr (!XCH) {
            if (st_wave && row <= r1) stash_row(cur, row);  // wave-uniform
        }""", """        if constexpr (!XCH && !SHU) {
            if (st_wave && row <= r1) stash_row(cur, row);  // wave-uniform
        }""")

# horizontal pass
rep("""        uint32_t win[XCH ? C::NW : (C::BROT ? C::NWIN : 1)];
        if constexpr (C::BROT) rotate_span<C>(cur, uint32_t(min(row, r1) * p.src_stride + lane_off) & 3u, win);""", """        uint32_t win[SHU ? CO::NWIN : XCH ? C::NW : (C::BROT ? C::NWIN : 1)];
        [[maybe_unused]] float uown[SHU ? S::OWN * CH : 1], unext[SHU ? S::NFC * 4 : 1];
        if constexpr (SHU) {
            rotate_span<CO>(cur, uint32_t(min(row, r1) * p.src_stride + lane_off) & 3u, win);
#pragma unroll
            for (int q = 0; q < S::OWN; ++q) unpack_span_px<CO, CH, S16, FMA>(win, q, &uown[q * CH]);
            const float uh = FMA ? float(cur[CO::NW]) : unpack_sample<S16>(cur[CO::NW]);
            f32x4 *xl = reinterpret_cast<f32x4 *>(stash_lds) + wave * (S::NFC * S::LANES);
#pragma unroll
            for (int c = 0; c < S::NFC; ++c) {
                f32x4 v = {uown[4 * c], uown[4 * c + 1], uown[4 * c + 2], uown[4 * c + 3]};   // (NFC * 4 <= OWN * CH + 3: checked below)
                xl[c * S::LANES + lane] = v;
            }
            const int hk = min(lane, S::NF - 1);
            reinterpret_cast<float *>(xl)[((hk >> 2) * S::LANES + kWave) * 4 + (hk & 3)] = uh;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int c = 0; c < S::NFC; ++c) {
                const f32x4 v = xl[c * S::LANES + lane + 1];
                unext[4 * c] = v[0]; unext[4 * c + 1] = v[1]; unext[4 * c + 2] = v[2]; unext[4 * c + 3] = v[3];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else
        if constexpr (C::BROT) rotate_span<C>(cur, uint32_t(min(row, r1) * p.src_stride + lane_off) & 3u, win);""")
rep("""#else
            if constexpr (C::BROT || XCH) unpack_span_px<C, CH, S16, FMA>(win, q, u);
            else unpack_span_px<C, CH, S16, FMA>(cur, q, u);
#endif""", """#else
            if constexpr (SHU) {
#pragma unroll
                for (int c = 0; c < CH; ++c) u[c] = q < S::OWN ? uown[q * CH + c] : unext[(q - S::OWN) * CH + c];
            }
            else if constexpr (C::BROT || XCH) unpack_span_px<C, CH, S16, FMA>(win, q, u);
            else unpack_span_px<C, CH, S16, FMA>(cur, q, u);
#endif""")
rep("""    static_assert(!PH || C::WORDS, "phased stores: dword-sized pixels");""", """    static_assert(!PH || C::WORDS, "phased stores: dword-sized pixels");
    static_assert(!SHU || S::NFC * 4 <= S::OWN * CH, "shared unpack: whole chunks of own floats");""")

# launch: LDS + name
rep("""    const size_t lds = p.stash ? sizeof(uint32_t) * bw * 2 * kWave * ((RunCfg<CH, S16, Q, PX, TX, P, RC>::NW + 3) / 4 * 4) : 0;   // the stash staging of the edge lanes
    if (p.resident_query) {""", """    using SH = RunShu<CH, S16, Q, PX, TX, P, RC>;
    if (XHL == -2 && p.stash) return hipErrorInvalidValue;   // shared unpack: the clipped columns come from the image
    const size_t lds = XHL == -2 ? sizeof(float) * 4 * bw * SH::NFC * SH::LANES   // the exchange slots
                     : p.stash ? sizeof(uint32_t) * bw * 2 * kWave * ((RunCfg<CH, S16, Q, PX, TX, P, RC>::NW + 3) / 4 * 4) : 0;   // the stash staging of the edge lanes
    if (p.resident_query) {""")
rep("""    if (XHL >= 0) snprintf(form, sizeof form, ",xch%d", XHL);""", """    if (XHL >= 0) snprintf(form, sizeof form, ",xch%d", XHL);
    if (XHL == -2) snprintf(form, sizeof form, ",shu");""")
rep("""    [[maybe_unused]] const int xhl = p.xch ? -p.x_base : -1;""", """    [[maybe_unused]] const int xhl = p.xch == 2 ? -2 : p.xch ? -p.x_base : -1;""")
# instantiations
rep("""#define PICHA_RUNZ_LIST_2(Z) /* grey */ Z(2, 15, 30, 4, 1, -1, 1) Z(2, 15, 31, 4, 0, -1, 1) Z(2, 15, 21, 4, 0, -1, 1)""",
    """#define PICHA_RUNZ_LIST_2(Z) /* grey */ Z(2, 15, 30, 4, 1, -1, 1) Z(2, 15, 31, 4, 0, -1, 1) Z(2, 15, 21, 4, 0, -1, 1) Z(2, 15, 30, 4, 1, -2, 1) Z(2, 15, 31, 4, 0, -2, 1) Z(2, 15, 21, 4, 0, -2, 1)""")
rep("""#define PICHA_RUNZ_LIST_3(Z) /* greya */ Z(2, 15, 30, 2, 1, -1, 1) Z(2, 15, 31, 2, 0, -1, 1)""",
    """#define PICHA_RUNZ_LIST_3(Z) /* greya */ Z(2, 15, 30, 2, 1, -1, 1) Z(2, 15, 31, 2, 0, -1, 1) Z(2, 15, 30, 2, 1, -2, 1) Z(2, 15, 31, 2, 0, -2, 1)""")
open(p, 'w').write(s)
print('ok')
