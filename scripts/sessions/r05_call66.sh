#!/bin/bash
# r05 call 66: survey cells where a periodic-table instantiation exists but the planner took the accumulator kernel (grey / greya at 1.5x and 3x, 64 x 4K): run = 1 (wherever
# instantiated) against the default, with the decline trace
export PYTHONPATH=. TMPDIR=/tmp PICHA_HIP_TRACE_PLAN=1
o=gpurun_out/r05c; mkdir -p $o
{
for c in 64:grey:3840:2160:2560:1440 64:grey:3840:2160:1280:720 64:greya:3840:2160:2560:1440 64:grey:3840:2160:2560:1440:cubic 64:grey:3840:2160:1280:720:cubic 256:grey:1920:1080:1280:720 64:grey:3840:2160:1536:864; do
python scripts/ab_shared.py --rounds 4 --case $c default=. run1=.:run=1 sym1=.:sym=1
done
} > $o/ab_planner_declines.txt 2>&1
grep -v amdgpu.ids $o/ab_planner_declines.txt | grep "^# [0-9]\|\[picha_hip\]\|median" | awk '!seen[$0]++' | sed 's/libpicha_hip.so *//; s/rounds: [0-9. ]*//' | cut -c1-200
