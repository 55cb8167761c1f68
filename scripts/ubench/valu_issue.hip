// valu_issue.hip (r05) — cycles per wave64 VALU instruction PER SIMD, by the number of waves that are co-resident ON THAT SIMD.
//
// Replaces valu_cycles.hip's reading (r04: "2.45 cycles per instruction"), whose residency was a chip-wide average over a launch that ran in
// two ragged rounds: idle SIMDs diluted the rate.  Here
//   * ONE resident round: the grid is 256 x k workgroups of 256 threads, k <= 8 (at most the 32 waves a CU holds);
//   * a START GATE: every wave adds itself to a global counter and spins (s_sleep, bounded by 50 ms of s_memrealtime) until all waves of the
//     grid have arrived, so the instruction stream of every wave runs while all the others run theirs;
//   * every wave records HW_REG_HW_ID and HW_REG_XCC_ID: co-residency is COUNTED per (XCC, SE, SH, CU, SIMD), not averaged;
//   * the loop body is 256 independent-enough instructions (8 chains), so the scalar loop overhead (s_add / s_cmp / s_cbranch) is < 2 %;
//   * rate of a SIMD = (last end - first start of its waves, s_memtime = shader cycles) / (its waves x instructions per wave).
// Output: per instruction class / mix, the median (p10 .. p90) over the SIMDs that held exactly n waves, for every n that occurred.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

struct Rec { unsigned long long t0, t1, rt0, rt1; unsigned hwid, xcc, gate_ok, pad; };

#define REP4(X) X X X X
#define REP32(X) REP4(X) REP4(X) REP4(X) REP4(X) REP4(X) REP4(X) REP4(X) REP4(X)

// body: 8 instructions on a0..a7 (floats), r0..r3 (unsigned), operands s (vgpr float), ss (sgpr float)
#define DEFK(NAME, ASM)                                                                                                             \
    __global__ __launch_bounds__(256) void NAME(Rec *rec, unsigned *gate, unsigned nwaves, int iters, float s, unsigned u, float *out) { \
        float a0 = threadIdx.x + s, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;         \
        unsigned r0 = threadIdx.x * 2654435761u + u, r1 = r0 ^ 0x1234567u, r2 = r0 * 3u, r3 = r1 * 5u;                                \
        double d0 = a0, d1 = a1, d2 = a2, d3 = a3;                                                                                    \
        float ss = __builtin_amdgcn_readfirstlane(__float_as_uint(s)) ? s : 0.f;                                                       \
        ss = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ss)));                                                     \
        unsigned ok = 1;                                                                                                             \
        {   /* start gate */                                                                                                         \
            if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(gate, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                \
            const unsigned long long lim = __builtin_amdgcn_s_memrealtime() + 5000000ull;                                            \
            while (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwaves) {                                    \
                __builtin_amdgcn_s_sleep(8);                                                                                         \
                if (__builtin_amdgcn_s_memrealtime() > lim) { ok = 0; break; }                                                        \
            }                                                                                                                        \
        }                                                                                                                            \
        unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();                                                                   \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                                        \
        for (int i = 0; i < iters; ++i) {                                                                                            \
            asm volatile(REP32(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(r0), "+v"(r1), "+v"(r2), \
                         "+v"(r3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(s), "s"(ss), "v"(u));                                     \
        }                                                                                                                            \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                                        \
        unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();                                                                   \
        if ((threadIdx.x & 63) == 0) {                                                                                               \
            Rec r; r.t0 = t0; r.t1 = t1; r.rt0 = rt0; r.rt1 = rt1; r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);                \
            r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20); r.gate_ok = ok; r.pad = 0;                                             \
            rec[blockIdx.x * 4 + (threadIdx.x >> 6)] = r;                                                                            \
        }                                                                                                                            \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + r0 + r1 + r2 + r3 + (float)(d0 + d1 + d2 + d3);   \
    }
// operand numbers: %0-%7 a0-a7, %8-%11 r0-r3, %12-%15 d0-d3 (register pairs), %16 s (vgpr), %17 ss (sgpr), %18 u (vgpr)
DEFK(k_fma, "v_fma_f32 %0, %0, %16, %1\n v_fma_f32 %1, %1, %16, %2\n v_fma_f32 %2, %2, %16, %3\n v_fma_f32 %3, %3, %16, %4\n v_fma_f32 %4, %4, %16, %5\n v_fma_f32 %5, %5, %16, %6\n v_fma_f32 %6, %6, %16, %7\n v_fma_f32 %7, %7, %16, %0\n")
DEFK(k_mul, "v_mul_f32 %0, %0, %16\n v_mul_f32 %1, %1, %16\n v_mul_f32 %2, %2, %16\n v_mul_f32 %3, %3, %16\n v_mul_f32 %4, %4, %16\n v_mul_f32 %5, %5, %16\n v_mul_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n")
DEFK(k_add, "v_add_f32 %0, %0, %16\n v_add_f32 %1, %1, %16\n v_add_f32 %2, %2, %16\n v_add_f32 %3, %3, %16\n v_add_f32 %4, %4, %16\n v_add_f32 %5, %5, %16\n v_add_f32 %6, %6, %16\n v_add_f32 %7, %7, %16\n")
DEFK(k_muladd, "v_mul_f32 %0, %0, %16\n v_add_f32 %1, %1, %16\n v_mul_f32 %2, %2, %16\n v_add_f32 %3, %3, %16\n v_mul_f32 %4, %4, %16\n v_add_f32 %5, %5, %16\n v_mul_f32 %6, %6, %16\n v_add_f32 %7, %7, %16\n")
// the dependent form of the kernels' sums: a product and the add that consumes it, 4 chains
DEFK(k_muladd_dep, "v_mul_f32 %0, %4, %16\n v_mul_f32 %1, %5, %16\n v_mul_f32 %2, %6, %16\n v_mul_f32 %3, %7, %16\n v_add_f32 %4, %4, %0\n v_add_f32 %5, %5, %1\n v_add_f32 %6, %6, %2\n v_add_f32 %7, %7, %3\n")
DEFK(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n")
DEFK(k_cvtub, "v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %8\n v_cvt_f32_ubyte2 %2, %8\n v_cvt_f32_ubyte3 %3, %8\n v_cvt_f32_ubyte0 %4, %9\n v_cvt_f32_ubyte1 %5, %9\n v_cvt_f32_ubyte2 %6, %9\n v_cvt_f32_ubyte3 %7, %9\n")
DEFK(k_med3, "v_med3_f32 %0, %0, %16, %1\n v_med3_f32 %1, %1, %16, %2\n v_med3_f32 %2, %2, %16, %3\n v_med3_f32 %3, %3, %16, %4\n v_med3_f32 %4, %4, %16, %5\n v_med3_f32 %5, %5, %16, %6\n v_med3_f32 %6, %6, %16, %7\n v_med3_f32 %7, %7, %16, %0\n")
DEFK(k_mul_sgpr, "v_mul_f32 %0, %17, %0\n v_mul_f32 %1, %17, %1\n v_mul_f32 %2, %17, %2\n v_mul_f32 %3, %17, %3\n v_mul_f32 %4, %17, %4\n v_mul_f32 %5, %17, %5\n v_mul_f32 %6, %17, %6\n v_mul_f32 %7, %17, %7\n")
DEFK(k_mul_sdwa, "v_mul_f32_sdwa %0, %8, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_mul_f32_sdwa %1, %8, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_mul_f32_sdwa %2, %8, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n v_mul_f32_sdwa %3, %8, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_mul_f32_sdwa %4, %9, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_mul_f32_sdwa %5, %9, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_mul_f32_sdwa %6, %9, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n v_mul_f32_sdwa %7, %9, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n")
DEFK(k_mov_dpp, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
DEFK(k_pk_mul, "v_pk_mul_f32 %12, %12, %14\n v_pk_mul_f32 %13, %13, %14\n v_pk_mul_f32 %12, %12, %15\n v_pk_mul_f32 %13, %13, %15\n v_pk_mul_f32 %12, %12, %14\n v_pk_mul_f32 %13, %13, %14\n v_pk_mul_f32 %12, %12, %15\n v_pk_mul_f32 %13, %13, %15\n")
DEFK(k_cvt_pk_u8, "v_cvt_pk_u8_f32 %8, %0, 0, %8\n v_cvt_pk_u8_f32 %9, %1, 1, %9\n v_cvt_pk_u8_f32 %10, %2, 2, %10\n v_cvt_pk_u8_f32 %11, %3, 3, %11\n v_cvt_pk_u8_f32 %8, %4, 0, %8\n v_cvt_pk_u8_f32 %9, %5, 1, %9\n v_cvt_pk_u8_f32 %10, %6, 2, %10\n v_cvt_pk_u8_f32 %11, %7, 3, %11\n")
// mixes
DEFK(k_mix_cvt26, "v_cvt_f32_ubyte0 %0, %8\n v_mul_f32 %1, %1, %16\n v_add_f32 %2, %2, %16\n v_mul_f32 %3, %3, %16\n v_cvt_f32_ubyte1 %4, %9\n v_mul_f32 %5, %5, %16\n v_add_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n")
DEFK(k_mix_cvt44, "v_cvt_f32_ubyte0 %0, %8\n v_mul_f32 %1, %1, %16\n v_cvt_f32_ubyte1 %2, %8\n v_add_f32 %3, %3, %16\n v_cvt_f32_ubyte2 %4, %9\n v_mul_f32 %5, %5, %16\n v_cvt_f32_ubyte3 %6, %9\n v_add_f32 %7, %7, %16\n")
DEFK(k_mix_sdwa17, "v_mul_f32_sdwa %0, %8, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_mul_f32 %1, %1, %16\n v_add_f32 %2, %2, %16\n v_mul_f32 %3, %3, %16\n v_add_f32 %4, %4, %16\n v_mul_f32 %5, %5, %16\n v_add_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n")
DEFK(k_mix_sdwa26, "v_mul_f32_sdwa %0, %8, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_mul_f32 %1, %1, %16\n v_add_f32 %2, %2, %16\n v_mul_f32 %3, %3, %16\n v_mul_f32_sdwa %4, %9, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_mul_f32 %5, %5, %16\n v_add_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n")
DEFK(k_mix_dpp17, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32 %1, %1, %16\n v_add_f32 %2, %2, %16\n v_mul_f32 %3, %3, %16\n v_add_f32 %4, %4, %16\n v_mul_f32 %5, %5, %16\n v_add_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n")
DEFK(k_mix_dpp26, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32 %1, %1, %16\n v_add_f32 %2, %2, %16\n v_mul_f32 %3, %3, %16\n v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32 %5, %5, %16\n v_add_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n")
DEFK(k_mix_pk26, "v_pk_mul_f32 %12, %12, %14\n v_mul_f32 %1, %1, %16\n v_add_f32 %2, %2, %16\n v_mul_f32 %3, %3, %16\n v_pk_mul_f32 %13, %13, %14\n v_mul_f32 %5, %5, %16\n v_add_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n")
DEFK(k_mix_sgpr26, "v_mul_f32 %0, %17, %0\n v_mul_f32 %1, %1, %16\n v_add_f32 %2, %2, %16\n v_mul_f32 %3, %3, %16\n v_mul_f32 %4, %17, %4\n v_mul_f32 %5, %5, %16\n v_add_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n")
DEFK(k_mix_med26, "v_med3_f32 %0, %0, %16, %1\n v_mul_f32 %1, %1, %16\n v_add_f32 %2, %2, %16\n v_mul_f32 %3, %3, %16\n v_med3_f32 %4, %4, %16, %5\n v_mul_f32 %5, %5, %16\n v_add_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n")
DEFK(k_mix_cvtpk26, "v_cvt_pk_u8_f32 %8, %0, 0, %8\n v_mul_f32 %1, %1, %16\n v_add_f32 %2, %2, %16\n v_mul_f32 %3, %3, %16\n v_cvt_pk_u8_f32 %9, %4, 1, %9\n v_mul_f32 %5, %5, %16\n v_add_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n")

typedef void (*kfn)(Rec *, unsigned *, unsigned, int, float, unsigned, float *);
struct Case { const char *name; kfn fn; bool all_levels; };

static double pct(std::vector<double> &v, double p) {
    std::sort(v.begin(), v.end());
    return v[std::min(v.size() - 1, size_t(p * (v.size() - 1) + 0.5))];
}

int main(int argc, char **argv) {
    Case cases[] = {{"v_fma_f32", k_fma, true}, {"v_mul_f32", k_mul, true}, {"v_add_f32", k_add, false}, {"mul add alternating (independent)", k_muladd, true},
                    {"mul then dependent add, 4 chains", k_muladd_dep, true}, {"v_mov_b32", k_mov, false}, {"v_cvt_f32_ubyteN", k_cvtub, true}, {"v_med3_f32", k_med3, false},
                    {"v_mul_f32 sgpr operand", k_mul_sgpr, false}, {"v_mul_f32_sdwa", k_mul_sdwa, true}, {"v_mov_b32_dpp wave_shr", k_mov_dpp, false},
                    {"v_pk_mul_f32", k_pk_mul, false}, {"v_cvt_pk_u8_f32", k_cvt_pk_u8, false},
                    {"mix 2 cvt_ubyte : 6 mul/add", k_mix_cvt26, true}, {"mix 4 cvt_ubyte : 4 mul/add", k_mix_cvt44, false},
                    {"mix 1 mul_sdwa : 7 mul/add", k_mix_sdwa17, false}, {"mix 2 mul_sdwa : 6 mul/add", k_mix_sdwa26, true},
                    {"mix 1 mov_dpp : 7 mul/add", k_mix_dpp17, false}, {"mix 2 mov_dpp : 6 mul/add", k_mix_dpp26, false},
                    {"mix 2 v_pk_mul_f32 : 6 mul/add", k_mix_pk26, false}, {"mix 2 sgpr-operand mul : 6 mul/add", k_mix_sgpr26, false},
                    {"mix 2 med3 : 6 mul/add", k_mix_med26, false}, {"mix 2 cvt_pk_u8 : 6 mul/add", k_mix_cvtpk26, false}};
    const int iters = argc > 1 ? atoi(argv[1]) : 1500;  // x 256 instructions per wave
    const double ninstr = double(iters) * 256.0;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("# device %s, %d CUs; %d x 256 VALU instructions per wave; rate per SIMD = (last end - first start of the SIMD's waves, shader cycles) / (waves x instructions)\n", prop.name,
           prop.multiProcessorCount, iters);
    const int max_blocks = 256 * 8;
    Rec *rec; unsigned *gate; float *out;
    hipMalloc(&rec, sizeof(Rec) * max_blocks * 4); hipMalloc(&gate, 64); hipMalloc(&out, sizeof(float) * max_blocks * 256);
    std::vector<Rec> h(max_blocks * 4);
    for (int wps : {1, 2, 3, 4, 5, 6, 8}) {
        const int blocks = 256 * wps;
        const unsigned nwaves = blocks * 4;
        printf("---- grid of %d workgroups x 4 waves = %d waves per SIMD if spread evenly ----\n", blocks, wps);
        for (auto &c : cases) {
            if (!c.all_levels && wps != 2 && wps != 4 && wps != 8) continue;
            for (int pass = 0; pass < 2; ++pass) {   // pass 0 warms the clocks / code
                hipMemset(gate, 0, 64);
                c.fn<<<blocks, 256>>>(rec, gate, nwaves, pass ? iters : 64, 1.0001f, 0x03020100u, out);
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), rec, sizeof(Rec) * nwaves, hipMemcpyDeviceToHost);
            struct S { unsigned long long first = ~0ull, last = 0, rfirst = ~0ull, rlast = 0; int n = 0; };
            std::map<unsigned, S> simd;
            int timeouts = 0;
            for (unsigned i = 0; i < nwaves; ++i) {
                const Rec &r = h[i];
                timeouts += !r.gate_ok;
                // HW_ID (gfx9 layout): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]; XCC_ID[3:0]
                const unsigned key = ((r.xcc & 0xf) << 16) | (r.hwid & 0xff30);
                S &s = simd[key];
                s.first = std::min(s.first, r.t0); s.last = std::max(s.last, r.t1);
                s.rfirst = std::min(s.rfirst, r.rt0); s.rlast = std::max(s.rlast, r.rt1);
                ++s.n;
            }
            std::map<int, std::vector<double>> by_n, clk_by_n;
            for (auto &kv : simd) {
                const S &s = kv.second;
                by_n[s.n].push_back(double(s.last - s.first) / (s.n * ninstr));
                clk_by_n[s.n].push_back(double(s.last - s.first) / (double(s.rlast - s.rfirst) * 10.0));
            }
            printf("%-36s SIMDs seen %4zu%s |", c.name, simd.size(), timeouts ? " GATE TIMEOUT (not one round)" : "");
            for (auto &kv : by_n) {
                auto &v = kv.second;
                printf("  n=%d on %zu SIMDs: %.2f (%.2f-%.2f) cyc/instr @ %.2f GHz;", kv.first, v.size(), pct(v, 0.5), pct(v, 0.1), pct(v, 0.9), pct(clk_by_n[kv.first], 0.5));
            }
            printf("\n");
        }
    }
    return 0;
}
