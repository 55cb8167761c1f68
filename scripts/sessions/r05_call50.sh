#!/bin/bash
# r05 call 50: cfg3b's edge pass as two more workgroups per (row chunk, image) INSIDE the march's grid (run_efuse = 1; no resize_edges launch, the march's side lanes
# store only their unclipped columns) against the march + edge kernel; parity of the run-kernel tests first
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05c; mkdir -p $o
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "edge_blocks or shared_unpack or edge_columns or fuzz" > $o/gputest_efuse.log 2>&1; rc=$?; tail -4 $o/gputest_efuse.log | cut -c1-300
[ $rc = 0 ] || exit $rc
{
for num in exact fma; do
python scripts/ab_shared.py --case cfg3b --numerics $num edgekernel=.:run_efuse=2 edgeblocks=.:run_efuse=1 edgeblocks_c8=.:run_efuse=1,row_chunks=8 edgeblocks_c12=.:run_efuse=1,row_chunks=12 noedge=.:run=3
done
python scripts/ab_shared.py --case 64:grey:3840:2160:512:512:lanczos edgekernel=.:run_efuse=2 edgeblocks=.:run_efuse=1
python scripts/ab_shared.py --case 256:grey:3840:2160:512:288:lanczos edgekernel=.:run_efuse=2 edgeblocks=.:run_efuse=1
} > $o/ab_edge_blocks.txt 2>&1
grep -v amdgpu.ids $o/ab_edge_blocks.txt | grep -v "^# device" | cut -c1-220
