#!/bin/bash
# r05 call 44: GPU suite; the neighbour-exchange run kernel (cfg5 and the rgba / r16g16b16a16 4x shapes) with the clipped columns inside against the edge kernel
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
python -m pytest tests -m gpu -q > $o/gputest_xch_ein.log 2>&1; tail -6 $o/gputest_xch_ein.log | cut -c1-250
{
for num in exact fma; do
python scripts/ab_shared.py --case cfg5 --numerics $num --rounds 8 edge_kernel=.:run_edges=2 inside=.
done
python scripts/ab_shared.py --case 64:r16g16b16a16:3840:2160:960:540 edge_kernel=.:run_edges=2 inside=.
python scripts/ab_shared.py --case 128:rgba:3840:2160:960:540 --numerics fma edge_kernel=.:run_edges=2 inside=.
} > $o/ab_xch_edges_inside.txt 2>&1
grep -v amdgpu.ids $o/ab_xch_edges_inside.txt | grep -v "^# device" | cut -c1-200
