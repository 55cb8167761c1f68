#!/bin/bash
# r05 call 62: the one-free-slot cap of the long-window accumulator kernel for rgb / rgba too (not the one-round cut) against the old choices; the whole GPU suite first
export PYTHONPATH=. TMPDIR=/tmp PICHA_HIP_TRACE_PLAN=1
o=gpurun_out/r05c; mkdir -p $o
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $o/gputest_long_chunks2.log 2>&1; rc=$?; tail -3 $o/gputest_long_chunks2.log | cut -c1-300
[ $rc = 0 ] || exit $rc
{
python scripts/ab_shared.py --case 64:rgb:3840:2160:384:216:cubic new=. old_c6=.:row_chunks=6
python scripts/ab_shared.py --case 64:rgb:3840:2160:384:216:cubic --numerics fma new=. old_c6=.:row_chunks=6
python scripts/ab_shared.py --case 64:rgb:3840:2160:320:180:lanczos new=. old_c6=.:row_chunks=6 old_c7=.:row_chunks=7
python scripts/ab_shared.py --case 64:rgba:3840:2160:384:216:cubic new=. old_c8=.:row_chunks=8
python scripts/ab_shared.py --case 64:rgba:3840:2160:320:180:lanczos new=. old_c7=.:row_chunks=7
python scripts/ab_shared.py --case 256:rgba:3840:2160:384:216:cubic new=. old_c6=.:row_chunks=6
python scripts/ab_shared.py --case 256:rgb:3840:2160:384:216:cubic new=. old_c6=.:row_chunks=6
python scripts/ab_shared.py --case 64:rgb:3840:2160:240:135:lanczos new=. c6=.:row_chunks=6 c8=.:row_chunks=8
} > $o/ab_acc_long_rule_wide.txt 2>&1
grep -v amdgpu.ids $o/ab_acc_long_rule_wide.txt | grep "^# [0-9]\|long-window plan\|median" | awk '!seen[$0]++' | sed 's/libpicha_hip.so *//; s/rounds:.*//' | cut -c1-160
