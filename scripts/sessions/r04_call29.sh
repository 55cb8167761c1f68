#!/bin/bash
# r04 call 29: the 4x cases of scripts/r04_ab_convert_up.py again, two more placements of the buffers (spacer allocations of 0.5 / 3 GB in front)
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 400 python scripts/r04_ab_convert_up.py 4x 512 > gpurun_out/r04/call29_ab_convert_up_4x_a.txt 2>&1 && timeout -k 10 400 python scripts/r04_ab_convert_up.py 4x 3072 > gpurun_out/r04/call29_ab_convert_up_4x_b.txt 2>&1
cut -c1-175 gpurun_out/r04/call29_ab_convert_up_4x_a.txt gpurun_out/r04/call29_ab_convert_up_4x_b.txt | grep "fuse 1"
