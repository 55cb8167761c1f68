#!/bin/bash
# r05 call 34: r16g16b16a16 2x upscales on the ring kernel with two outputs per lane (clipped columns inside) against the accumulator form
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
{
for num in exact fma; do
python scripts/ab_shared.py --case 64:r16g16b16a16:960:540:1920:1080 --numerics $num acc=.:run=2 ring_p2=.
python scripts/ab_shared.py --case 64:r16g16b16a16:960:540:1920:1080:mitchel --numerics $num acc=.:run=2 ring_p2=.
done
python scripts/ab_shared.py --case 16:r16g16b16a16:1920:1080:3840:2160 acc=.:run=2 ring_p2=.
} > $o/ab_up_rgba16.txt 2>&1
grep -v amdgpu.ids $o/ab_up_rgba16.txt | cut -c1-200
python -m pytest tests -m gpu -x -q -k "upscale or edge_columns or fuzz or planner or selection" > $o/gputest_up16.log 2>&1; tail -3 $o/gputest_up16.log
