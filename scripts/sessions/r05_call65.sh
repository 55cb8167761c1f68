#!/bin/bash
# r05 call 65: the new chunk cap on SMALL batches of narrow pixels (16 / 8 / 4 images), against the counts the 80 % cap gave
export PYTHONPATH=. TMPDIR=/tmp PICHA_HIP_TRACE_PLAN=1
o=gpurun_out/r05c; mkdir -p $o
{
python scripts/ab_shared.py --case 16:grey:3840:2160:384:216:cubic new=. old_c27=.:row_chunks=27 c36=.:row_chunks=36
python scripts/ab_shared.py --case 8:grey:3840:2160:384:216:cubic new=. c54=.:row_chunks=54 c43=.:row_chunks=43
python scripts/ab_shared.py --case 16:greya:3840:2160:320:180:lanczos new=. c12=.:row_chunks=12 c10=.:row_chunks=10 c9=.:row_chunks=9
python scripts/ab_shared.py --case 32:grey:3840:2160:320:180:lanczos new=. c12=.:row_chunks=12 c13=.:row_chunks=13 c14=.:row_chunks=14
python scripts/ab_shared.py --case 4:grey:3840:2160:400:225:lanczos new=. c90=.:row_chunks=90 c113=.:row_chunks=113
} > $o/ab_acc_long_small.txt 2>&1
grep -v amdgpu.ids $o/ab_acc_long_small.txt | grep "^# [0-9]\|long-window plan\|median" | awk '!seen[$0]++' | sed 's/libpicha_hip.so *//; s/rounds:.*//' | cut -c1-160
