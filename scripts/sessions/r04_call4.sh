#!/bin/bash
# r04 call 4: what bounds cfg5's read pattern?  stores, workgroup width, row pitch, footprint, a linear stream for comparison
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/call4_march.txt; : > $out
M=scripts/ubench/march_pattern
B="16,1,4,240,0,0,0,1,0"
echo "## base, stores off / plain, full-row workgroups (15 / 16 waves), with a barrier per row, XCD-contiguous order" >> $out
timeout -k 10 120 $M 32 61440 4320 4 11 0 $B 16,1,4,240,0,0,0,0,0 16,1,4,240,0,0,0,2,0 16,1,15,240,0,0,0,1,0 16,1,16,240,0,0,0,1,0 16,1,15,240,0,0,0,1,1 16,1,4,240,0,0,0,1,1 16,1,15,240,1,0,0,1,0 16,2,15,240,0,0,0,1,0 16,1,15,240,0,0,0,0,0 >> $out 2>&1
for pad in 64 128 256 1024 4096; do
  echo "## row pitch + $pad" >> $out
  timeout -k 10 60 $M 32 61440 4320 4 11 $pad $B 16,1,15,240,0,0,0,1,0 >> $out 2>&1
done
echo "## footprint: 1, 4, 8 images (chunks shortened to keep >= 2304 workgroups)" >> $out
timeout -k 10 60 $M 1 61440 4320 4 11 0 16,1,4,8,0,0,0,1,0 16,1,4,16,0,0,0,1,0 >> $out 2>&1
timeout -k 10 60 $M 4 61440 4320 4 11 0 16,1,4,32,0,0,0,1,0 16,1,4,60,0,0,0,1,0 >> $out 2>&1
timeout -k 10 60 $M 8 61440 4320 4 11 0 16,1,4,60,0,0,0,1,0 >> $out 2>&1
echo "## the same bytes as ONE linear stream per workgroup: rows of 16 KB (a workgroup's four waves), 16200 rows, chunk 900, no halo" >> $out
timeout -k 10 60 $M 32 16384 16200 4 0 0 16,1,4,900,0,0,0,1,0 16,1,4,900,0,0,0,0,0 16,2,4,900,0,0,0,1,0 16,1,4,225,0,0,0,1,0 16,1,4,60,0,0,0,1,0 16,1,4,16,0,0,0,1,0 >> $out 2>&1
echo "## rows of 4 KB: one wave per row (fully sequential per wave)" >> $out
timeout -k 10 60 $M 32 4096 64800 4 0 0 16,1,1,3600,0,0,0,1,0 16,1,1,900,0,0,0,1,0 16,1,1,64,0,0,0,1,0 16,1,1,16,0,0,0,1,0 >> $out 2>&1
cat $out
