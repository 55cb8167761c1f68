#!/bin/bash
# r05 call 49: long fuzz sweep of the HEAD library over a random stream no earlier sweep used (PICHA_FUZZ_BASE=830000: 120 resize seeds x 70 cases, 40 device-batch
# seeds), bounded; then the bench line twice more in fresh processes (process-to-process spread on this box)
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05c; mkdir -p $o
PICHA_FUZZ_SEEDS=120 PICHA_FUZZ_DEVICE_SEEDS=40 PICHA_FUZZ_BASE=830000 timeout -k 10 840 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x > $o/fuzz_long.log 2>&1; rc=$?
tail -4 $o/fuzz_long.log | cut -c1-300
[ $rc = 0 ] || exit $rc
python bench.py --no-cpu-baseline > $o/bench_line.run_b.json 2> $o/bench_line.run_b.err && python bench.py --no-cpu-baseline > $o/bench_line.run_c.json 2> $o/bench_line.run_c.err
cut -c1-300 $o/bench_line.run_b.json $o/bench_line.run_c.json
