#!/bin/bash
# r05 call 10: which part of the batch does each XCD work on?  contiguous eighths (shipped) / (image, row chunk) units round-robin / images round-robin; both flavours, a few chunk lengths
export PYTHONPATH=. TMPDIR=/tmp
out=gpurun_out/r05; mkdir -p $out
for num in exact fma; do
timeout -k 10 600 python scripts/ab_shared.py --case cfg2 --numerics $num --rounds 5 --reps 10 base=. o1=.:work_order=1 o2=.:work_order=2 o1_c27=.:work_order=1,row_chunks=27 o2_c27=.:work_order=2,row_chunks=27 o0_c27=.:row_chunks=27 o1_c54=.:work_order=1,row_chunks=54 o2_c54=.:work_order=2,row_chunks=54 o1_bw2=.:work_order=1,block_waves=2 o2_bw2=.:work_order=2,block_waves=2 o2_t0=.:work_order=2,work_tail=0 o1_t0=.:work_order=1,work_tail=0 > $out/call10_cfg2_work_order_$num.txt 2>&1
cut -c1-200 $out/call10_cfg2_work_order_$num.txt
done
