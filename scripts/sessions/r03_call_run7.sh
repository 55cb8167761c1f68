#!/bin/bash
# r03: run kernel, window offsets rounded down (31 taps) against up (30 taps) on cfg3b and the greya twin
set -e
mkdir -p gpurun_out
python scripts/ab_opts.py '[{"run":2},{"run_round":1},{"run_round":2},{"run_round":1,"numerics":1},{"run_round":2,"numerics":1}]' '{"filter":"lanczos"}' 256:grey:3840:2160:512:512 4 10 > gpurun_out/run7.txt 2>&1
python scripts/ab_opts.py '[{"run":2},{"run_round":1},{"run_round":2}]' '{"filter":"lanczos"}' 128:greya:3840:2160:512:288 3 8 >> gpurun_out/run7.txt 2>&1
grep -v amdgpu gpurun_out/run7.txt
timeout -k 10 300 python -m pytest tests/test_gpu_chain_and_helpers.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -q -k "run_kernel or half_integer or kernel_selection" 2>&1 | tail -5
