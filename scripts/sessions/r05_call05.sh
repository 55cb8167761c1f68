#!/bin/bash
# r05 call 5: (a) do line-aligned windows pay? (measurement builds: every window shifted by 8 / 72 bytes, 13 lines per wave-row instead of 14; results shifted)
# (b) do persistent waves march in lockstep? (pseudo-random start delay per wave)
export PYTHONPATH=. TMPDIR=/tmp
out=gpurun_out/r05; mkdir -p $out
bash scripts/unpack_variants.sh
V=picha_amd/variants
for num in exact fma; do
timeout -k 10 400 python scripts/ab_shared.py --case cfg2 --numerics $num --no-oracle --rounds 6 --reps 10 base=. al8=$V/libpicha_hip_al8.so al72=$V/libpicha_hip_al72.so pw=.:work=4 pw_stag=$V/libpicha_hip_stag.so:work=4 pw500=.:work=4,pw_grid=500 pw500_stag=$V/libpicha_hip_stag.so:work=4,pw_grid=500 pw500bw2_stag=$V/libpicha_hip_stag.so:work=4,pw_grid=500,block_waves=2 > $out/call05_cfg2_align_stag_$num.txt 2>&1
cut -c1-215 $out/call05_cfg2_align_stag_$num.txt
done
