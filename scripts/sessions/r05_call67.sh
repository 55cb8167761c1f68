#!/bin/bash
# r05 call 67: the run kernel's short-window shapes for one- and two-channel pixels by WORK (>= 400 000 wave-rows) as well as by image count: the whole GPU suite, then the
# new default against the accumulator kernel (run = 2) on 4K batches of 32 .. 96 images
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05c; mkdir -p $o
timeout -k 10 600 python -m pytest tests -m gpu -q > $o/gputest_wave_rows.log 2>&1; rc=$?; tail -3 $o/gputest_wave_rows.log | cut -c1-300
[ $rc = 0 ] || exit $rc
{
for c in 64:grey:3840:2160:2560:1440 64:grey:3840:2160:1536:864 64:grey:3840:2160:1280:720 64:greya:3840:2160:2560:1440 64:greya:3840:2160:1536:864 48:grey:3840:2160:2560:1440 80:grey:3840:2160:768:432 64:grey:3840:2160:960:540 32:grey:7680:4320:3072:1728 64:grey:3840:2160:2560:1440:triangle; do
python scripts/ab_shared.py --rounds 4 --case $c default=. acc=.:run=2 run1=.:run=1
done
python scripts/ab_shared.py --rounds 4 --case 64:grey:3840:2160:2560:1440 --numerics fma default=. acc=.:run=2
python scripts/ab_shared.py --rounds 4 --case 64:grey:3840:2160:1536:864 --numerics fma default=. acc=.:run=2
} > $o/ab_planner_wave_rows.txt 2>&1
grep -v amdgpu.ids $o/ab_planner_wave_rows.txt | grep "^# [0-9]\|median" | sed 's/libpicha_hip.so *//; s/rounds: [0-9. ]*//; s/; ONE source.*numerics/;/' | cut -c1-190
