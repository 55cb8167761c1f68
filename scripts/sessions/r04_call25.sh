#!/bin/bash
# r04 call 25: the memory pattern of the thumbnail kernels (192 x 4K rgb-like rows of 12 KB, one 16-byte store per lane every 8 rows, 14 halo rows per 120-row chunk)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
M=scripts/ubench/march_pattern
timeout -k 10 200 $M 192 12288 2160 8 14 0:S,D 4,1,4,120,0,0,0,1,0 4,1,4,120,0,0,0,0,0 4,2,4,120,0,0,0,1,0 4,4,4,120,0,0,0,1,0 8,1,4,120,0,0,0,1,0 8,2,4,120,0,0,0,1,0 16,1,3,120,0,0,0,1,0 16,2,3,120,0,0,0,1,0 4,2,4,240,0,0,0,1,0 4,2,4,60,0,0,0,1,0 8,2,4,120,0,0,0,1,0,0,0,0,0,20 8,2,4,120,0,0,0,1,0,0,0,0,0,40 > gpurun_out/r04/call25_march_thumb.txt 2>&1
grep "^rnd 1\|^march" gpurun_out/r04/call25_march_thumb.txt | cut -c1-60,128-250
