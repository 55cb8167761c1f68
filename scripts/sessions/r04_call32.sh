#!/bin/bash
# r04 call 32: VALU issue rates again (scripts/ubench/valu_cycles.hip), with the number of waves really resident per SIMD measured
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 300 scripts/ubench/valu_cycles > gpurun_out/r04/call32_valu_cycles.txt 2>&1
cut -c1-230 gpurun_out/r04/call32_valu_cycles.txt
