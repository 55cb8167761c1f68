#!/bin/bash
# r04 call 7: does the physical distance between source and destination matter?  fresh process per allocation plan
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/call7_march_plans.txt; : > $out
M=scripts/ubench/march_pattern
V="16,1,4,240,0,0,0,1,0 16,1,4,240,0,0,0,0,0"
for plan in S,D D,S S,P1,D S,P2,D S,P4,D S,P8,D S,P16,D S,P32,D S,P64,D S,P128,D P16,S,D P16,D,S D,P16,S P64,S,P64,D; do
  timeout -k 10 60 $M 32 61440 4320 4 11 0:$plan $V >> $out 2>&1
done
grep "^rnd 1\|^plan" $out | cut -c1-60,150-230
