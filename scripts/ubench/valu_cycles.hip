// Microbenchmark 2: cycles per wave-instruction per SIMD for the VALU forms the resize kernel
// can use, measured with s_memtime inside the kernel (independent of DVFS), 8 resident waves/SIMD.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

#define DEFK(NAME, ASM, ...)                                                                         \
    __global__ __launch_bounds__(256) void NAME(unsigned long long *cyc, float *out, int iters, float s, unsigned u) { \
        float a0 = threadIdx.x + s, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;                            \
        double d0 = a0, d1 = a1, d2 = a2, d3 = a3;                                                   \
        unsigned r0 = threadIdx.x * 2654435761u + u, r1 = r0 ^ 0x1234567u;                           \
        unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();                                   \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                        \
        for (int i = 0; i < iters; ++i) { asm volatile(REP8(ASM) : __VA_ARGS__); }                   \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                        \
        unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();                                   \
        if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0; cyc[gridDim.x * 4 + blockIdx.x * 4 + (threadIdx.x >> 6)] = rt1 - rt0; \
            cyc[gridDim.x * 8 + blockIdx.x * 4 + (threadIdx.x >> 6)] = rt0; cyc[gridDim.x * 12 + blockIdx.x * 4 + (threadIdx.x >> 6)] = rt1; } \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(d0 + d1 + d2 + d3) + r0 + r1; \
    }

// each ASM body = 4 independent instructions; REP8 -> 32 per loop iteration
DEFK(k_mul, "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_mul_s, "v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s))
DEFK(k_add, "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_fma, "v_fma_f32 %0, %0, %4, %1\n v_fma_f32 %1, %1, %4, %2\n v_fma_f32 %2, %2, %4, %3\n v_fma_f32 %3, %3, %4, %0\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_fmac, "v_fmac_f32 %0, %4, %1\n v_fmac_f32 %1, %4, %2\n v_fmac_f32 %2, %4, %3\n v_fmac_f32 %3, %4, %0\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_pkmul, "v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %3\n v_pk_mul_f32 %1, %1, %3\n", "+v"(d0), "+v"(d1) : "v"(d2), "v"(d3))
DEFK(k_pkfma, "v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %3, %2\n v_pk_fma_f32 %1, %1, %3, %2\n", "+v"(d0), "+v"(d1) : "v"(d2), "v"(d3))
DEFK(k_cvtub, "v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(r0))
DEFK(k_cvtu32, "v_cvt_f32_u32 %0, %4\n v_cvt_f32_u32 %1, %5\n v_cvt_f32_u32 %2, %4\n v_cvt_f32_u32 %3, %5\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(r0), "v"(r1))
DEFK(k_cvtsdwa, "v_cvt_f32_u32_sdwa %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %2, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %3, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(r0), "v"(r1))
DEFK(k_med3, "v_med3_f32 %0, %0, %4, %1\n v_med3_f32 %1, %1, %4, %2\n v_med3_f32 %2, %2, %4, %3\n v_med3_f32 %3, %3, %4, %0\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_max, "v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_cvtu, "v_cvt_u32_f32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_cvt_u32_f32 %2, %2\n v_cvt_u32_f32 %3, %3\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_cvtpku8, "v_cvt_pk_u8_f32 %0, %4, 0, %0\n v_cvt_pk_u8_f32 %1, %4, 1, %1\n v_cvt_pk_u8_f32 %2, %4, 2, %2\n v_cvt_pk_u8_f32 %3, %4, 3, %3\n", "+v"(r0), "+v"(r1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_perm, "v_perm_b32 %0, %0, %1, %4\n v_perm_b32 %1, %1, %0, %4\n v_perm_b32 %0, %0, %1, %4\n v_perm_b32 %1, %1, %0, %4\n", "+v"(r0), "+v"(r1), "+v"(a2), "+v"(a3) : "v"(u))
DEFK(k_lshlor, "v_lshl_or_b32 %0, %0, 8, %1\n v_lshl_or_b32 %1, %1, 8, %0\n v_lshl_or_b32 %0, %0, 8, %1\n v_lshl_or_b32 %1, %1, 8, %0\n", "+v"(r0), "+v"(r1), "+v"(a2), "+v"(a3) : "v"(u))
DEFK(k_or, "v_or_b32 %0, %0, %1\n v_or_b32 %1, %1, %0\n v_or_b32 %0, %0, %1\n v_or_b32 %1, %1, %0\n", "+v"(r0), "+v"(r1), "+v"(a2), "+v"(a3) : "v"(u))
DEFK(k_movdpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_movwshr, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_cvtdpp, "v_cvt_f32_ubyte0_dpp %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_cvt_f32_ubyte1_dpp %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_cvt_f32_ubyte2_dpp %2, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_cvt_f32_ubyte3_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(r0))
DEFK(k_muldpp, "v_mul_f32_dpp %0, %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %2, %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %3, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_mulsdwa, "v_mul_f32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_mul_f32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_mul_f32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_mul_f32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))

DEFK(k_mul_lit, "v_mul_f32 %0, 0x3b808081, %0\n v_mul_f32 %1, 0x3b808081, %1\n v_mul_f32 %2, 0x3b808081, %2\n v_mul_f32 %3, 0x3b808081, %3\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_add_inl, "v_add_f32 %0, 0.5, %0\n v_add_f32 %1, 0.5, %1\n v_add_f32 %2, 0.5, %2\n v_add_f32 %3, 0.5, %3\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_addu, "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0\n v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0\n", "+v"(r0), "+v"(r1), "+v"(a2), "+v"(a3) : "v"(u))
DEFK(k_addu_s, "v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n", "+v"(r0), "+v"(r1), "+v"(a2), "+v"(a3) : "s"(u))
DEFK(k_fma_s, "v_fma_f32 %0, %4, %1, %0\n v_fma_f32 %1, %4, %2, %1\n v_fma_f32 %2, %4, %3, %2\n v_fma_f32 %3, %4, %0, %3\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s))
DEFK(k_fmac_dpp, "v_fmac_f32_dpp %0, %4, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %4, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %2, %4, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %4, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))

// r04 mixes: does a half-rate class overlap with the multiply-add pipe, or do the costs add up?  (4 instructions per body; sum model: the per-class cycles averaged)
DEFK(k_mix_cvt13, "v_cvt_f32_ubyte0 %0, %4\n v_mul_f32 %1, %1, %5\n v_add_f32 %2, %2, %5\n v_mul_f32 %3, %3, %5\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(r0), "v"(s))
DEFK(k_mix_cvt22, "v_cvt_f32_ubyte0 %0, %4\n v_mul_f32 %1, %1, %5\n v_cvt_f32_ubyte1 %2, %4\n v_add_f32 %3, %3, %5\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(r0), "v"(s))
DEFK(k_mix_sdwa13, "v_mul_f32_sdwa %0, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_mul_f32 %1, %1, %5\n v_add_f32 %2, %2, %5\n v_mul_f32 %3, %3, %5\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(r0), "v"(s))
DEFK(k_mix_muladd, "v_mul_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_mix_med13, "v_med3_f32 %0, %0, %4, %1\n v_mul_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_mix_sgpr13, "v_mul_f32 %0, %5, %0\n v_mul_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s), "s"(s))
DEFK(k_mix_sgpr22, "v_mul_f32 %0, %5, %0\n v_add_f32 %1, %1, %4\n v_mul_f32 %2, %5, %2\n v_add_f32 %3, %3, %4\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s), "s"(s))
DEFK(k_mix_dpp13, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_mix_align13, "v_alignbyte_b32 %0, %0, %1, 1\n v_mul_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_mix_cvtu32_13, "v_cvt_f32_u32 %0, %5\n v_mul_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s), "v"(r0))
DEFK(k_mix_cvtpk13, "v_cvt_pk_u8_f32 %0, %4, 0, %0\n v_mul_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n", "+v"(r0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s))
DEFK(k_mix_pk13, "v_pk_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %5\n v_add_f32 %2, %2, %5\n v_mul_f32 %3, %3, %5\n", "+v"(d0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(d2), "v"(s))
DEFK(k_mix_pk22, "v_pk_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n", "+v"(d0), "+v"(a1), "+v"(d1), "+v"(a3) : "v"(d2), "v"(s))
DEFK(k_mix_pkfma13, "v_pk_fma_f32 %0, %0, %4, %4\n v_mul_f32 %1, %1, %5\n v_add_f32 %2, %2, %5\n v_mul_f32 %3, %3, %5\n", "+v"(d0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(d2), "v"(s))
typedef void (*kfn)(unsigned long long *, float *, int, float, unsigned);
struct Case { const char *name; kfn fn; };

int main() {
    Case cases[] = {{"v_mul_f32", k_mul}, {"v_mul_f32 (sgpr src)", k_mul_s}, {"v_add_f32", k_add}, {"v_fma_f32", k_fma}, {"v_fmac_f32", k_fmac},
                    {"v_pk_mul_f32", k_pkmul}, {"v_pk_fma_f32", k_pkfma}, {"v_cvt_f32_ubyteN", k_cvtub}, {"v_cvt_f32_u32", k_cvtu32},
                    {"v_cvt_f32_u32_sdwa", k_cvtsdwa}, {"v_med3_f32", k_med3}, {"v_max_f32", k_max}, {"v_cvt_u32_f32", k_cvtu},
                    {"v_cvt_pk_u8_f32", k_cvtpku8}, {"v_perm_b32", k_perm}, {"v_lshl_or_b32", k_lshlor}, {"v_or_b32", k_or},
                    {"v_mov_b32", k_mov}, {"v_mov_b32_dpp row_shr", k_movdpp}, {"v_mov_b32_dpp wave_shr", k_movwshr},
                    {"v_cvt_f32_ubyteN_dpp", k_cvtdpp}, {"v_mul_f32_dpp", k_muldpp}, {"v_mul_f32_sdwa", k_mulsdwa}, {"mix 1 cvt_ubyte : 3 mul/add", k_mix_cvt13}, {"mix 2 cvt_ubyte : 2 mul/add", k_mix_cvt22}, {"mix 1 mul_sdwa : 3 mul/add", k_mix_sdwa13}, {"mix mul add mul add", k_mix_muladd}, {"mix 1 med3 : 3 mul/add", k_mix_med13}, {"mix 1 v_pk_mul_f32 : 3 mul/add", k_mix_pk13}, {"mix 2 v_pk_mul_f32 : 2 mul/add", k_mix_pk22}, {"mix 1 v_pk_fma_f32 : 3 mul/add", k_mix_pkfma13}, {"mix 1 sgpr-operand mul : 3 mul/add", k_mix_sgpr13}, {"mix 2 sgpr-operand mul : 2 add", k_mix_sgpr22}, {"mix 1 v_mov_dpp wave_shr : 3 mul/add", k_mix_dpp13}, {"mix 1 v_alignbyte : 3 mul/add", k_mix_align13}, {"mix 1 v_cvt_f32_u32 : 3 mul/add", k_mix_cvtu32_13}, {"mix 1 v_cvt_pk_u8_f32 : 3 mul/add", k_mix_cvtpk13}, {"v_mul_f32 literal", k_mul_lit}, {"v_add_f32 inline 0.5", k_add_inl}, {"v_add_u32", k_addu}, {"v_add_u32 sgpr", k_addu_s}, {"v_fma_f32 sgpr", k_fma_s}, {"v_fmac_f32_dpp", k_fmac_dpp}};
    const int iters = 20000;
    // r04: waves per SIMD 1 .. 8, and how many waves were really resident at once: the sum of the waves' own durations over the span from the first start to
    // the last end (s_memrealtime, chip-wide) per SIMD.  (The r01 table divided a wave's ticks by 8 "resident" waves; see profiles/r04_ubench_valu_rates.txt.)
    const bool brief = true;
    for (int wps : {1, 2, 4, 8}) {          // waves per SIMD asked for
        const int blocks = 256 * wps;    // 256-thread blocks = 1 wave per SIMD each
        unsigned long long *cyc; float *out;
        hipMalloc(&cyc, blocks * 4 * 8 * 4); hipMalloc(&out, blocks * 256 * 4);
        printf("---- %d wave(s) per SIMD asked for ----\n", wps);
        int idx = 0;
        for (auto &c : cases) {
            ++idx;
            if (brief && wps != 8 && idx > 5 && strcmp(c.name, "v_cvt_f32_ubyteN") && strcmp(c.name, "v_med3_f32") && strcmp(c.name, "v_mul_f32_sdwa")) continue;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            c.fn<<<blocks, 256>>>(cyc, out, 10, 1.0001f, 0x03020100u);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            c.fn<<<blocks, 256>>>(cyc, out, iters, 1.0001f, 0x03020100u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(blocks * 4 * 4);
            hipMemcpy(h.data(), cyc, blocks * 4 * 8 * 4, hipMemcpyDeviceToHost);
            double sum = 0, rsum = 0; unsigned long long first = ~0ull, last = 0;
            for (int i = 0; i < blocks * 4; ++i) { sum += h[i]; rsum += h[blocks * 4 + i]; first = std::min(first, h[blocks * 8 + i]); last = std::max(last, h[blocks * 12 + i]); }
            const double per_wave = sum / (blocks * 4), rt = rsum / (blocks * 4);
            const double span = double(last - first);                 // 10 ns ticks
            const double resident = rsum / span / 1024.0;             // average waves in flight per SIMD
            const double clock = per_wave / (rt * 10.0);              // GHz
            const double ns_per_instr_simd = span * 10.0 / (iters * 32.0) / wps;   // by the chip-wide clock: all SIMDs run wps waves' worth of instructions within the span
            printf("%-26s in-kernel %.3f ms/wave, span %.3f ms, event %.3f ms | %.2f waves resident per SIMD | clock %.2f GHz | %.2f ns = %.2f cycles per wave-instruction per SIMD\n", c.name,
                   rt * 1e-5, span * 1e-5, ms, resident, clock, ns_per_instr_simd, ns_per_instr_simd * clock);
        }
        hipFree(cyc); hipFree(out);
    }
    return 0;
}
