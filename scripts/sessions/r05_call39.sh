#!/bin/bash
# r05 call 39: placement against allocation history — ten destination candidates in each of eight successive processes
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
{
for i in 1 2 3 4 5 6 7 8; do python scripts/r05_placement_history.py --cands 10 2>&1 | grep -v amdgpu; done
} > $o/placement_history10.txt 2>&1
cat $o/placement_history10.txt | cut -c1-200
