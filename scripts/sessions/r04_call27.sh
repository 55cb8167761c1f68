#!/bin/bash
# r04 call 27: cfg3b-like geometry (see r04_call26.sh), time-phased stores 0 / 5 / 10 / 20 / 40 / 80 us, two store densities (16 B per lane every 16 / 8 rows)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
M=scripts/ubench/march_pattern
for yd in 16 8; do
timeout -k 10 200 $M 256 4096 2160 $yd 14 0:S,D 8,1,2,135,0,0,0,1,0 8,1,2,135,0,0,0,0,0 8,1,2,135,0,0,0,1,0,0,0,0,0,5 8,1,2,135,0,0,0,1,0,0,0,0,0,10 8,1,2,135,0,0,0,1,0,0,0,0,0,20 8,1,2,135,0,0,0,1,0,0,0,0,0,40 8,1,2,135,0,0,0,1,0,0,0,0,0,80 8,1,2,135,0,0,0,1,0,0,0,0,0,160 >> gpurun_out/r04/call27_march_cfg3b_phase.txt 2>&1 || exit 1
done
grep "^rnd 1\|^march" gpurun_out/r04/call27_march_cfg3b_phase.txt | cut -c1-60,128-250
