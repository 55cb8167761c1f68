#!/bin/bash
# r05 call 17: cfg3 (rgb -> grey -> 512x512 lanczos as ONE kernel, resize_rows): one round of 1024 four-wave workgroups today (4 chunks); chunk counts, and cfg3b / lanczos 2x / thumbnails over chunk counts
export PYTHONPATH=. TMPDIR=/tmp
out=gpurun_out/r05; mkdir -p $out
for num in exact fma; do
timeout -k 10 600 python scripts/ab_shared.py --case cfg3 --numerics $num --rounds 4 --reps 10 base=. c2=.:row_chunks=2 c3=.:row_chunks=3 c4=.:row_chunks=4 c5=.:row_chunks=5 c6=.:row_chunks=6 c8=.:row_chunks=8 c12=.:row_chunks=12 c16=.:row_chunks=16 > $out/call17_cfg3_chunks_$num.txt 2>&1
cut -c1-230 $out/call17_cfg3_chunks_$num.txt
done
for c in lanczos2x lanczos8x jpegthumb cfg2_7tap; do
timeout -k 10 600 python scripts/ab_shared.py --case $c --rounds 3 --reps 10 base=. c4=.:row_chunks=4 c6=.:row_chunks=6 c8=.:row_chunks=8 c12=.:row_chunks=12 c16=.:row_chunks=16 c24=.:row_chunks=24 c32=.:row_chunks=32 bw1=.:block_waves=1 bw2=.:block_waves=2 bw4=.:block_waves=4 > $out/call17_${c}_chunks.txt 2>&1
cut -c1-230 $out/call17_${c}_chunks.txt
done
