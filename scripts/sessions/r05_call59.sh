#!/bin/bash
# r05 call 59: chunk-count sweep of the long-window accumulator kernel on the cells of call 58 (one-wave workgroups), with the planner's own choice traced
export PYTHONPATH=. TMPDIR=/tmp PICHA_HIP_TRACE_PLAN=1
o=gpurun_out/r05c; mkdir -p $o
{
V="default=."
for c in 2 3 4 5 6 7 8 9 10 11 12 14; do V="$V c$c=.:row_chunks=$c"; done
python scripts/ab_shared.py --rounds 3 --case 64:grey:3840:2160:384:216:cubic $V
python scripts/ab_shared.py --rounds 3 --case 64:grey:3840:2160:320:180:lanczos $V
python scripts/ab_shared.py --rounds 3 --case 64:greya:3840:2160:333:187:mitchel $V
python scripts/ab_shared.py --rounds 3 --case 64:grey:3840:2160:240:135:cubic $V
python scripts/ab_shared.py --rounds 3 --case 16:grey:3840:2160:384:216:cubic $V
python scripts/ab_shared.py --rounds 3 --case 256:grey:3840:2160:384:216:cubic $V
} > $o/ab_acc_long_chunks.txt 2>&1
grep -v amdgpu.ids $o/ab_acc_long_chunks.txt | grep -v "^# device" | sort -u -k1,1 -s | head -0
grep -v amdgpu.ids $o/ab_acc_long_chunks.txt | grep "^#\|long-window plan\|median" | awk '!seen[$0]++' | cut -c1-20,90-135,1-0 | head -120
