#!/bin/bash
# r04 call 3: memory-pattern microbenchmark for cfg5's march + in-kernel clocks of the shipped kernels
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 400 scripts/ubench/march_pattern 32 61440 4320 4 11 > gpurun_out/r04/call3_march_cfg5.txt 2>&1; echo "march rc=$?"
cat gpurun_out/r04/call3_march_cfg5.txt
PICHA_HIP_LIB=picha_amd/variants/libpicha_hip_probe.so timeout -k 10 400 python scripts/r04_clock_probe.py > gpurun_out/r04/call3_clock.txt 2>&1; echo "clock rc=$?"
cat gpurun_out/r04/call3_clock.txt
