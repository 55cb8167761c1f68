#!/bin/bash
# r05 call 7: is the persistent form's loss an address-stride effect?  All persistent waves march in step, so the rows read at one moment are chunk-length x row-pitch apart
# (60 rows x 15360 B = 2^12 x 225).  Chunk lengths with fewer / more trailing zero bits, in both launch forms
export PYTHONPATH=. TMPDIR=/tmp
out=gpurun_out/r05; mkdir -p $out
args=""
for c in 15 16 17 18 19 20 21 23 27; do args="$args pw_c$c=.:work=4,block_waves=2,row_chunks=$c"; done
for c in 15 17 19 23; do args="$args list_c$c=.:row_chunks=$c"; done
for c in 17 19 23; do args="$args pw500_c$c=.:work=4,pw_grid=500,block_waves=2,row_chunks=$c"; done
timeout -k 10 600 python scripts/ab_shared.py --case cfg2 --rounds 4 --reps 10 base=. $args > $out/call07_cfg2_chunk_stride.txt 2>&1
cut -c1-200 $out/call07_cfg2_chunk_stride.txt
