#!/bin/bash
# r05 call 11: cfg3b / cfg5 launch as 2 - 2.4 rounds of resident waves and spend their last third ramping down (profiles/r05_timeline.txt): chunk counts that make ONE round
# (cfg3b: 8 chunks = 4096 waves) or whole numbers of rounds, and less halo with it
export PYTHONPATH=. TMPDIR=/tmp
out=gpurun_out/r05; mkdir -p $out
for num in exact fma; do
timeout -k 10 600 python scripts/ab_shared.py --case cfg3b --numerics $num --rounds 4 --reps 10 base=. c6=.:row_chunks=6 c8=.:row_chunks=8 c10=.:row_chunks=10 c12=.:row_chunks=12 c16=.:row_chunks=16 c24=.:row_chunks=24 c8_bw1=.:row_chunks=8,block_waves=1 c16_bw1=.:row_chunks=16,block_waves=1 > $out/call11_cfg3b_chunks_$num.txt 2>&1
cut -c1-220 $out/call11_cfg3b_chunks_$num.txt
timeout -k 10 600 python scripts/ab_shared.py --case cfg5 --numerics $num --rounds 4 --reps 10 base=. c6=.:row_chunks=6 c8=.:row_chunks=8 c9=.:row_chunks=9 c10=.:row_chunks=10 c12=.:row_chunks=12 c16=.:row_chunks=16 c18=.:row_chunks=18 > $out/call11_cfg5_chunks_$num.txt 2>&1
cut -c1-220 $out/call11_cfg5_chunks_$num.txt
done
