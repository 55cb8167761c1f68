#!/bin/bash
# r04 call 16: in-kernel clocks with the low-overhead probe; destination placement sensitivity of the real kernels; bench + gpu tests of the new bench.py
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
PICHA_HIP_LIB=picha_amd/variants/libpicha_hip_probe.so timeout -k 10 400 python scripts/r04_clock_probe.py > gpurun_out/r04/call16_clock.txt 2>&1; echo "clock rc=$?"
grep -v amdgpu.ids gpurun_out/r04/call16_clock.txt
for c in cfg5 cfg2 cfg3b; do timeout -k 10 300 python scripts/r04_placement.py $c 6 0 >> gpurun_out/r04/call16_placement.txt 2>&1; done
timeout -k 10 300 python scripts/r04_placement.py cfg5 6 1024 >> gpurun_out/r04/call16_placement.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/call16_placement.txt
timeout -k 10 600 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_parity.py -x -q -m gpu -k "bench or per_call or eviction" > gpurun_out/r04/call16_tests.txt 2>&1; tail -4 gpurun_out/r04/call16_tests.txt
timeout -k 10 300 python bench.py > gpurun_out/r04/call16_bench.json 2> gpurun_out/r04/call16_bench.err; cat gpurun_out/r04/call16_bench.json | cut -c1-1800
