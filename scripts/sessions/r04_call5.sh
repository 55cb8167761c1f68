#!/bin/bash
# r04 call 5: read/write mixing in cfg5's march — cache policies of loads and stores, placement of the destination
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/call5_march.txt; : > $out
M=scripts/ubench/march_pattern
V=""
for sp in 1 3 16 17 18 19; do V="$V 16,1,4,240,0,0,0,3,0,0,$sp,0,0"; done
for lp in 2 16 1 18; do V="$V 16,1,4,240,0,0,0,1,0,$lp,0,0,0 16,1,4,240,0,0,0,0,0,$lp,0,0,0"; done
V="$V 16,1,4,240,0,0,0,3,0,2,3,0,0 16,1,4,240,0,0,0,3,0,2,16,0,0 16,1,4,240,0,0,0,3,0,2,18,0,0 16,1,4,240,0,0,0,3,0,16,16,0,0 16,1,4,240,0,0,0,3,0,16,18,0,0 16,1,4,240,0,0,0,3,0,18,18,0,0"
for pl in 1 2; do for gap in 0 1 4 61 1024 32768; do V="$V 16,1,4,240,0,0,0,1,0,0,0,$pl,$gap"; done; done
timeout -k 10 500 $M 32 61440 4320 4 11 0 16,1,4,240,0,0,0,1,0 16,1,4,240,0,0,0,0,0 16,1,4,240,0,0,0,2,0 $V >> $out 2>&1
grep "^rnd 1\|^march" $out | cut -c1-200
