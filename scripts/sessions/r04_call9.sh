#!/bin/bash
# r04 call 9: time-phased stores (every wave holds its finished outputs and stores them when a chip-wide 100 MHz clock crosses a phase boundary)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/call9_march_phased.txt; : > $out
M=scripts/ubench/march_pattern
V="16,1,4,240,0,0,0,1,0 16,1,4,240,0,0,0,0,0"
for us in 5 10 20 40 60 80 100 150; do V="$V 16,1,4,240,0,0,0,1,0,0,0,0,0,$us"; done
for plan in S,D D,S; do timeout -k 10 100 $M 32 61440 4320 4 11 0:$plan $V >> $out 2>&1; done
grep "^rnd 1\|^march\|^plan" $out | cut -c1-60,128-240
