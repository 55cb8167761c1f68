#!/bin/bash
# r05 call 16: cfg5's kernel issues an SDWA / DPP stream (4.0 cycles per instruction, 85 % busy at that rate: profiles/r05_issue_util.txt).  The same kernel without SDWA
# (extraction + multiply as two plain instructions, compiler's SDWA peephole off), without DPP (window through LDS), and without both; phased and unphased (4 waves per SIMD)
export PYTHONPATH=. TMPDIR=/tmp
out=gpurun_out/r05; mkdir -p $out
bash scripts/unpack_variants.sh
V=picha_amd/variants
for num in exact fma; do
timeout -k 10 600 python scripts/ab_shared.py --case cfg5 --numerics $num --rounds 5 --reps 10 base=. nosdwa=$V/libpicha_hip_nosdwa.so xlds=$V/libpicha_hip_xlds.so both=$V/libpicha_hip_both.so base_unph=.:run_phase=-1 both_unph=$V/libpicha_hip_both.so:run_phase=-1 both_c8=$V/libpicha_hip_both.so:row_chunks=8 nosdwa_unph=$V/libpicha_hip_nosdwa.so:run_phase=-1 > $out/call16_cfg5_nosdwa_$num.txt 2>&1
cut -c1-250 $out/call16_cfg5_nosdwa_$num.txt
done
timeout -k 10 300 python scripts/ab_shared.py --case 64:r16g16b16a16:3840:2160:960:540 --rounds 3 --reps 10 base=. both=$V/libpicha_hip_both.so nosdwa=$V/libpicha_hip_nosdwa.so > $out/call16_rgba16_4x_default.txt 2>&1; cut -c1-250 $out/call16_rgba16_4x_default.txt
