#!/bin/bash
# r05 call 33: GPU suite; upscales with the clipped columns produced inside the march (per-output weight sets) against the edge kernel
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
python -m pytest tests -m gpu -q > $o/gputest_ein.log 2>&1; tail -30 $o/gputest_ein.log | cut -c1-250
{
for num in exact fma; do
python scripts/ab_shared.py --case 64:rgba:960:540:1920:1080 --numerics $num edge_kernel=.:run_edges=2 inside=.
python scripts/ab_shared.py --case 64:r16:960:540:1920:1080 --numerics $num edge_kernel=.:run_edges=2 inside=.
done
python scripts/ab_shared.py --case 64:rgb:960:540:1920:1080 edge_kernel=.:run_edges=2 inside=.
python scripts/ab_shared.py --case 64:grey:960:540:1920:1080 edge_kernel=.:run_edges=1 inside=.
python scripts/ab_shared.py --case 64:greya:960:540:1920:1080:mitchel edge_kernel=.:run_edges=2 inside=.
python scripts/ab_shared.py --case 256:rgba:480:270:1920:1080 edge_kernel=.:run_edges=2 inside=.
python scripts/ab_shared.py --case 256:grey:480:270:1920:1080 edge_kernel=.:run_edges=1 inside=.
python scripts/ab_shared.py --case 8:rgba:960:540:1920:1080 edge_kernel=.:run_edges=2 inside=.
} > $o/ab_edges_inside.txt 2>&1
grep -v amdgpu.ids $o/ab_edges_inside.txt | cut -c1-200
