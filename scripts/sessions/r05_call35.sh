#!/bin/bash
# r05 call 35: 4x upscales: runs of four with the clipped columns inside against runs of eight + edge kernel (rgba), against the accumulator form (r16g16b16a16)
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
{
for num in exact fma; do
python scripts/ab_shared.py --case 256:rgba:480:270:1920:1080 --numerics $num p8=. p4_inside=.:per_lane=4
python scripts/ab_shared.py --case 64:r16g16b16a16:480:270:1920:1080 --numerics $num acc=.:run=2 p4_inside=.
done
python scripts/ab_shared.py --case 64:rgba:480:270:1920:1080:mitchel p8=. p4_inside=.:per_lane=4
} > $o/ab_up4.txt 2>&1
grep -v amdgpu.ids $o/ab_up4.txt | cut -c1-200
