#!/bin/bash
# r05 call 23: cfg3 (fused rows kernel): one output per thread (two strips of 256 outputs per row: 16 KB of LDS per workgroup, 8 workgroups per CU) against the default two
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
{
for num in exact fma; do
python scripts/ab_shared.py --case cfg3 --numerics $num default=. p1=.:per_lane=1 p1_c2=.:per_lane=1,row_chunks=2 p1_c3=.:per_lane=1,row_chunks=3 p1_c4=.:per_lane=1,row_chunks=4 p1_c8=.:per_lane=1,row_chunks=8 c5=.:row_chunks=5 c6=.:row_chunks=6
done
python scripts/ab_shared.py --case 256:grey:3840:2160:512:512:lanczos rows=.:rows=1,run=2 rows_p1=.:rows=1,run=2,per_lane=1 rows_p1_c2=.:rows=1,run=2,per_lane=1,row_chunks=2 default=.
} > $o/ab_rows_p1.txt 2>&1
grep -v amdgpu.ids $o/ab_rows_p1.txt | cut -c1-250
