#!/bin/bash
# r05 call 30: colorConvert -> 7.5x downscale on the shared-unpack run kernel (convert in front) against the row-staged kernel: cfg3, rgba -> grey, default-filter thumbnails; GPU suite
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
python -m pytest tests -m gpu -q > $o/gputest_cv.log 2>&1; tail -25 $o/gputest_cv.log | cut -c1-300
{
for num in exact fma; do
python scripts/ab_shared.py --case cfg3 --numerics $num rows=.:run=2 runcv=. runcv_c8=.:row_chunks=8 runcv_c10=.:row_chunks=10 runcv_c12=.:row_chunks=12 runcv_bw1=.:block_waves=1
done
} > $o/ab_run_cv.txt 2>&1
grep -v amdgpu.ids $o/ab_run_cv.txt | cut -c1-230
