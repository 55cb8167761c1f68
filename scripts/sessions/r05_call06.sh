#!/bin/bash
# r05 call 6: few waves, deep prefetch?  NB = 4 / 6 row buffers (3 / 5 rows in flight; 130 / 148 VGPRs = 3 waves per SIMD) in the list form and as persistent waves at 3 / 2.5 / 2 waves per SIMD
export PYTHONPATH=. TMPDIR=/tmp
out=gpurun_out/r05; mkdir -p $out
bash scripts/unpack_variants.sh
V=picha_amd/variants
for num in exact fma; do
timeout -k 10 400 python scripts/ab_shared.py --case cfg2 --numerics $num --rounds 5 --reps 10 base=. nb4=$V/libpicha_hip_nb4.so nb6=$V/libpicha_hip_nb6.so nb6_bw1=$V/libpicha_hip_nb6.so:block_waves=1 nb6_pw=$V/libpicha_hip_nb6.so:work=4 nb6_pw833=$V/libpicha_hip_nb6.so:work=4,pw_grid=833 nb6_pw667=$V/libpicha_hip_nb6.so:work=4,pw_grid=667 nb6_pw667_bw2=$V/libpicha_hip_nb6.so:work=4,pw_grid=667,block_waves=2 nb6_pw667_c9=$V/libpicha_hip_nb6.so:work=4,pw_grid=667,row_chunks=9 nb4_pw=$V/libpicha_hip_nb4.so:work=4 nb4_pw667=$V/libpicha_hip_nb4.so:work=4,pw_grid=667 > $out/call06_cfg2_nbuf_$num.txt 2>&1
cut -c1-215 $out/call06_cfg2_nbuf_$num.txt
done
