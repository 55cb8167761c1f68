#!/bin/bash
# r04 call 26: the memory pattern of cfg3b with COALESCED loads (256 x 4K grey rows, here 4096 B so that two 2 KB wave strips tile a row; 14 halo rows per 135-row chunk,
# 16 bytes stored per lane every 16 rows ~ the 512-wide output) — against the shipped kernel's loads + stores alone (0.51 ms, windows at a 30-byte lane stride)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
M=scripts/ubench/march_pattern
timeout -k 10 200 $M 256 4096 2160 16 14 0:S,D 8,1,2,135,0,0,0,1,0 8,1,2,135,0,0,0,0,0 8,2,2,135,0,0,0,1,0 8,4,2,135,0,0,0,1,0 8,2,1,135,0,0,0,1,0 8,2,4,135,0,0,0,1,0 8,2,2,270,0,0,0,1,0 8,2,2,68,0,0,0,1,0 4,2,4,135,0,0,0,1,0 16,1,1,135,0,0,0,1,0 > gpurun_out/r04/call26_march_cfg3b.txt 2>&1
grep "^rnd 1\|^march" gpurun_out/r04/call26_march_cfg3b.txt | cut -c1-60,128-250
