#!/bin/bash
# r03: run kernel + edge kernel (stash) on cfg3b and 7.5x thumbnails: launch forms, row chunks, numerics; kernel-trace; parity test
set -e
mkdir -p gpurun_out
python scripts/ab_opts.py '[{"run":2},{"run":0,"work":3},{"run":3,"work":3},{"run":0,"work":1},{"run":0,"work":3,"row_chunks":8},{"run":0,"work":3,"row_chunks":11},{"run":0,"work":3,"block_waves":1},{"run":3,"work":3,"numerics":1},{"run":0,"work":3,"numerics":1},{"run":2,"numerics":1}]' '{"filter":"lanczos"}' 256:grey:3840:2160:512:512 4 10 > gpurun_out/run3_cfg3b.txt 2>&1
cat gpurun_out/run3_cfg3b.txt
python scripts/ab_opts.py '[{"run":2},{"run":0},{"run":0,"work":3},{"run":0,"per_lane":2,"work":3},{"run":0,"per_lane":8,"work":3},{"run":0,"work":3,"numerics":1},{"run":2,"numerics":1}]' '{}' 256:grey:3840:2160:512:288 3 10 > gpurun_out/run3_thumb.txt 2>&1
cat gpurun_out/run3_thumb.txt
PROF_TRACE_ONLY=1 bash scripts/prof.sh run_cfg3b --workload cfg3b_4k_grey_to_512_lanczos | head -8
timeout -k 10 300 python -m pytest tests/test_gpu_chain_and_helpers.py -x -q -k "run_kernel" 2>&1 | tail -15
