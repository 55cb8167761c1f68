#!/bin/bash
# r05 call 52: a much longer fuzz sweep of the final library (call 49 showed a seed takes 0.1 s): 1500 resize seeds x 70 cases, 400 device-batch seeds, on another unused stream
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05c; mkdir -p $o
PICHA_FUZZ_SEEDS=1500 PICHA_FUZZ_DEVICE_SEEDS=400 PICHA_FUZZ_BASE=2460000 timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x > $o/fuzz_very_long.log 2>&1; rc=$?
tail -6 $o/fuzz_very_long.log | cut -c1-400
exit $rc
