#!/bin/bash
# r05 call 61: does the one-free-slot / one-round chunk rule of the long-window accumulator kernel carry over to rgb / rgba / 16-bit pixels?  chunk sweeps with the plan traced
export PYTHONPATH=. TMPDIR=/tmp PICHA_HIP_TRACE_PLAN=1
o=gpurun_out/r05c; mkdir -p $o
{
V="default=."
for c in 2 3 4 5 6 7 8 9 10 12 14 16; do V="$V c$c=.:row_chunks=$c"; done
python scripts/ab_shared.py --rounds 3 --case 64:rgb:3840:2160:384:216:cubic $V
python scripts/ab_shared.py --rounds 3 --case 64:rgba:3840:2160:384:216:cubic $V
python scripts/ab_shared.py --rounds 3 --case 256:rgba:3840:2160:384:216:cubic $V
python scripts/ab_shared.py --rounds 3 --case 64:rgba:3840:2160:320:180:lanczos $V
python scripts/ab_shared.py --rounds 3 --case 64:r16:3840:2160:384:216:cubic $V
python scripts/ab_shared.py --rounds 3 --case 32:r16g16b16a16:3840:2160:384:216:cubic $V
} > $o/ab_acc_long_wide.txt 2>&1
grep -v amdgpu.ids $o/ab_acc_long_wide.txt | grep "^# [0-9]\|long-window plan\|median" | awk '!seen[$0]++' | sed 's/libpicha_hip.so *//; s/rounds:.*//' | cut -c1-150
