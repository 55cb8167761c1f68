#!/bin/bash
# r05 call 37: GPU suite; runs of eight with 4-tap windows (32 weight registers per lane) with the clipped columns inside against the edge kernel
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
python -m pytest tests -m gpu -q > $o/gputest_ein32.log 2>&1; tail -6 $o/gputest_ein32.log | cut -c1-250
{
for px in grey greya r16; do
python scripts/ab_shared.py --case 128:$px:480:270:1920:1080 edge_kernel=.:run_edges=2 inside=.
python scripts/ab_shared.py --case 128:$px:960:540:1920:1080:mitchel edge_kernel=.:run_edges=2 inside=.
done
python scripts/ab_shared.py --case 128:grey:480:270:1920:1080 --numerics fma edge_kernel=.:run_edges=2 inside=.
python scripts/ab_shared.py --case 128:rgb:480:270:1920:1080 p8=.:per_lane=8 default=.
} > $o/ab_edges_inside32.txt 2>&1
grep -v amdgpu.ids $o/ab_edges_inside32.txt | grep -v "^# device" | cut -c1-200
