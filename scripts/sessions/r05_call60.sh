#!/bin/bash
# r05 call 60: the long-window accumulator kernel's chunk rule for one- and two-byte pixels (one free slot per CU instead of 80 % of the residency; one round instead of
# several ragged ones) against the old choices (forced row_chunks), the whole GPU suite first
export PYTHONPATH=. TMPDIR=/tmp PICHA_HIP_TRACE_PLAN=1
o=gpurun_out/r05c; mkdir -p $o
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $o/gputest_long_chunks.log 2>&1; rc=$?; tail -3 $o/gputest_long_chunks.log | cut -c1-300
[ $rc = 0 ] || exit $rc
{
python scripts/ab_shared.py --case 64:grey:3840:2160:384:216:cubic new=. old_c6=.:row_chunks=6
python scripts/ab_shared.py --case 64:grey:3840:2160:384:216:cubic --numerics fma new=. old_c6=.:row_chunks=6
python scripts/ab_shared.py --case 64:grey:3840:2160:320:180:lanczos new=. old_c6=.:row_chunks=6
python scripts/ab_shared.py --case 64:greya:3840:2160:333:187:mitchel new=. old_c6=.:row_chunks=6
python scripts/ab_shared.py --case 64:grey:3840:2160:240:135:cubic new=. old_c8=.:row_chunks=8
python scripts/ab_shared.py --case 256:grey:3840:2160:384:216:cubic new=. old_c6=.:row_chunks=6 c1=.:row_chunks=1
python scripts/ab_shared.py --case 128:grey:3840:2160:384:216:cubic new=. old_c6=.:row_chunks=6
python scripts/ab_shared.py --case 256:greya:3840:2160:333:187:mitchel new=. old_c6=.:row_chunks=6
python scripts/ab_shared.py --case 128:grey:3840:2160:320:180:lanczos new=. old_c6=.:row_chunks=6 c4=.:row_chunks=4
python scripts/ab_shared.py --case 32:grey:3840:2160:384:216:cubic new=. c12=.:row_chunks=12 c14=.:row_chunks=14 c16=.:row_chunks=16
python scripts/ab_shared.py --case 64:grey:7680:4320:512:288:lanczos new=. c4=.:row_chunks=4 c6=.:row_chunks=6 c8=.:row_chunks=8
} > $o/ab_acc_long_rule.txt 2>&1
grep -v amdgpu.ids $o/ab_acc_long_rule.txt | grep "^# [0-9]\|long-window plan\|median" | awk '!seen[$0]++' | sed 's/libpicha_hip.so *//; s/rounds:.*//' | cut -c1-160
