#!/bin/bash
# r04 call 6: placement of the destination relative to the source at GB scale (one allocation, gap sweep), both orders
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/call6_march_placement.txt; : > $out
M=scripts/ubench/march_pattern
V="16,1,4,240,0,0,0,1,0 16,1,4,240,0,0,0,0,0"
for gap in 0 262144 524288 1048576 2097152 3145728 4194304 6291456 8388608 12582912 16777216; do V="$V 16,1,4,240,0,0,0,1,0,0,0,1,$gap"; done
for gap in 0 524288 1048576 2097152 4194304 8388608 16777216; do V="$V 16,1,4,240,0,0,0,1,0,0,0,3,$gap"; done
timeout -k 10 500 $M 32 61440 4320 4 11 0 $V >> $out 2>&1
grep "^rnd\|^march" $out | cut -c1-200
