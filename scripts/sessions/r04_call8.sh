#!/bin/bash
# r04 call 8: how the cost of the stores scales with how often a wave stores (same 16 bytes per lane each time)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/call8_march_ydiv.txt; : > $out
M=scripts/ubench/march_pattern
for yd in 1 2 4 8 16 32; do
  timeout -k 10 60 $M 32 61440 4320 $yd 11 0:S,D 16,1,4,240,0,0,0,1,0 16,1,4,240,0,0,0,0,0 >> $out 2>&1
done
grep "^rnd 1\|^march" $out | cut -c1-200
