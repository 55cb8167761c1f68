#!/bin/bash
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 400 python scripts/ab_opts.py '[{"run_xch":2},{"run_phase":-1},{"run_phase":12},{"run_phase":12,"row_chunks":12},{"run_phase":12,"row_chunks":9},{"numerics":1,"run_xch":2},{"numerics":1,"run_phase":-1},{"numerics":1,"run_phase":12},{"numerics":1,"run_phase":12,"row_chunks":12}]' '{"filter":"mitchel"}' 32:r16g16b16a16:7680:4320:1920:1080 4 10 > gpurun_out/r04/call20_ab_cfg5.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/call20_ab_cfg5.txt
timeout -k 10 400 python scripts/ab_opts.py '[{"run_xch":2},{"run_phase":-1},{"run_phase":12},{"run_phase":12,"row_chunks":12}]' '{"filter":"mitchel"}' 128:r16g16b16a16:7680:4320:1920:1080 3 5 > gpurun_out/r04/call20_ab_cfg5_128.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/call20_ab_cfg5_128.txt
timeout -k 10 600 python -m pytest tests/test_gpu_chain_and_helpers.py tests/test_gpu_fullsize.py -x -q -m gpu -k "neighbour_exchange or cfg5" > gpurun_out/r04/call20_tests.txt 2>&1; tail -3 gpurun_out/r04/call20_tests.txt
