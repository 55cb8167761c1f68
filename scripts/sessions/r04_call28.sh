#!/bin/bash
# r04 call 28: colorConvert -> integer upscale as one kernel against the two-kernel chain (scripts/r04_ab_convert_up.py)
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 700 python scripts/r04_ab_convert_up.py > gpurun_out/r04/call28_ab_convert_up.txt 2>&1
cut -c1-230 gpurun_out/r04/call28_ab_convert_up.txt | grep -v "fuse 0"
