#!/bin/bash
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_chain_and_helpers.py -x -q -m gpu -k "run_kernel" > gpurun_out/r04/call22_tests.txt 2>&1; tail -3 gpurun_out/r04/call22_tests.txt
timeout -k 10 400 python scripts/ab_opts.py '[{},{"per_lane":8},{"per_lane":8,"row_chunks":12},{"per_lane":8,"row_chunks":16},{"per_lane":8,"row_chunks":24},{"per_lane":8,"block_waves":1,"row_chunks":16},{"numerics":1},{"numerics":1,"per_lane":8},{"numerics":1,"per_lane":8,"row_chunks":12},{"numerics":1,"per_lane":8,"row_chunks":16},{"numerics":1,"per_lane":8,"row_chunks":24}]' '{"filter":"lanczos"}' 256:grey:3840:2160:512:512 4 10 > gpurun_out/r04/call22_ab_cfg3b_p8.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/call22_ab_cfg3b_p8.txt
