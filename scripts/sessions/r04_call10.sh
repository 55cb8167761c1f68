#!/bin/bash
# r04 call 10: time-phased stores on cfg2's geometry (256 x 4K rgba rows of 15360 bytes, 4:1 read:write) and cfg3b-like (grey rows of 3840 bytes)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/call10_march_phased_cfg2.txt; : > $out
M=scripts/ubench/march_pattern
V="4,2,4,120,0,0,0,1,0 4,2,4,120,0,0,0,0,0 4,4,4,120,0,0,0,1,0"
for us in 2 5 10 15 20 30 40; do V="$V 4,2,4,120,0,0,0,1,0,0,0,0,0,$us"; done
timeout -k 10 200 $M 256 15360 2160 4 3 0:S,D $V >> $out 2>&1
echo "## 8:1" >> $out
V="4,2,4,120,0,0,0,1,0 4,2,4,120,0,0,0,0,0"
for us in 5 10 20 40; do V="$V 4,2,4,120,0,0,0,1,0,0,0,0,0,$us"; done
timeout -k 10 200 $M 256 15360 2160 8 3 0:S,D $V >> $out 2>&1
grep "^rnd 1\|^march\|^plan\|^##" $out | cut -c1-60,128-240
