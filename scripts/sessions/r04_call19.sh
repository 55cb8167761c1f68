#!/bin/bash
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r04; mkdir -p $o/sec
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $o/gputest_b.txt 2>&1; echo "gpu tests rc=$?"; tail -5 $o/gputest_b.txt
PICHA_HIP_LIB=picha_amd/variants/libpicha_hip_probe.so timeout -k 10 500 python scripts/r04_clock_probe.py > $o/sec/clock2.txt 2>&1; echo "clock rc=$?"
grep -v amdgpu.ids $o/sec/clock2.txt | cut -c1-250
