#!/bin/bash
# r05 call 9: the memory pattern alone (loads + stores, no arithmetic), held at the shipped kernel's occupancy by unused LDS (bw4 workgroups: 36 KB = 4 waves per SIMD,
# 48 KB = 3, 72 KB = 2), over outputs per lane 3 / 5 / 6 / 10 / 15 (strip widths 1.5 / 2.5 / 3 / 5 / 7.5 KB per wave and row) and chunk lengths
export PYTHONPATH=. TMPDIR=/tmp
out=gpurun_out/r05; mkdir -p $out
bash scripts/unpack_variants.sh
A=picha_amd/variants/libpicha_hip_abl3w.so
args="full=."
for lds in 0 36864 49152 73728; do args="$args p3_lds$lds=$A:PICHA_SYM_LDS=$lds"; done
for pl in 5 6 10 15; do for lds in 36864 49152 73728; do args="$args p${pl}_lds$lds=$A:per_lane=$pl,PICHA_SYM_LDS=$lds"; done; done
timeout -k 10 600 python scripts/ab_shared.py --case cfg2 --no-oracle --rounds 3 --reps 10 $args > $out/call09_cfg2_noarith_widths.txt 2>&1
cut -c1-210 $out/call09_cfg2_noarith_widths.txt
args="full=."
for c in 9 18 27 54; do for pl in 3 6; do args="$args p${pl}_c${c}=$A:per_lane=$pl,row_chunks=$c,PICHA_SYM_LDS=49152"; done; done
timeout -k 10 600 python scripts/ab_shared.py --case cfg2 --no-oracle --rounds 3 --reps 10 $args > $out/call09_cfg2_noarith_chunks.txt 2>&1
cut -c1-210 $out/call09_cfg2_noarith_chunks.txt
