#!/bin/bash
# r05 call 38: placement against allocation history — successive processes, with and without a ballast allocated first
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
{
for b in 0 0 0 0 32 32 0 96 96 0 0 ; do python scripts/r05_placement_history.py --ballast-gb $b --cands 5 2>&1 | grep -v amdgpu; done
for b in 16 64; do python scripts/r05_placement_history.py --ballast-after-src-gb $b --cands 4 2>&1 | grep -v amdgpu; done
python scripts/r05_placement_history.py --src-cands 3 --cands 3 2>&1 | grep -v amdgpu
python scripts/r05_placement_history.py --ballast-gb 64 --src-cands 3 --cands 3 2>&1 | grep -v amdgpu
} > $o/placement_history.txt 2>&1
cat $o/placement_history.txt | cut -c1-250
