#!/bin/bash
# r05 call 36: survey — upscales whose longest run does not fit the per-output weight sets (4x: runs of eight; 2x with 4-tap filters: runs of eight): runs of four with the
# clipped columns inside against runs of eight + edge kernel
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
{
for px in rgb grey greya r16; do
python scripts/ab_shared.py --case 128:$px:480:270:1920:1080 p8=.:per_lane=8 p4_inside=.:per_lane=4
python scripts/ab_shared.py --case 128:$px:480:270:1920:1080 --numerics fma p8=.:per_lane=8 p4_inside=.:per_lane=4
done
for px in grey greya r16; do
python scripts/ab_shared.py --case 128:$px:960:540:1920:1080:mitchel p8=.:per_lane=8 p4_inside=.:per_lane=4
done
} > $o/ab_up_survey.txt 2>&1
grep -v amdgpu.ids $o/ab_up_survey.txt | grep -v "^# device" | cut -c1-200
