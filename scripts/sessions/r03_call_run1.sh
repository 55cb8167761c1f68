#!/bin/bash
# r03: measurements of the run kernel (kernels_run.hip) on cfg3b and a 7.5x thumbnail against the kernels that serve them
set -e
mkdir -p gpurun_out
python scripts/ab_opts.py '[{"run":2},{"run":0},{"run":0,"work":3},{"run":0,"row_chunks":8},{"run":0,"row_chunks":8,"work":3},{"run":0,"row_chunks":11},{"run":0,"row_chunks":24},{"run":0,"per_lane":2},{"run":0,"numerics":1},{"run":2,"numerics":1}]' '{"filter":"lanczos"}' 256:grey:3840:2160:512:512 3 10 > gpurun_out/run1_cfg3b.txt 2>&1
cat gpurun_out/run1_cfg3b.txt
python scripts/ab_opts.py '[{"run":2},{"run":0},{"run":0,"per_lane":2},{"run":0,"per_lane":8},{"run":0,"work":3},{"run":0,"numerics":1},{"run":2,"numerics":1}]' '{}' 256:grey:3840:2160:512:288 3 10 > gpurun_out/run1_thumb.txt 2>&1
cat gpurun_out/run1_thumb.txt
PROF_TRACE_ONLY=1 bash scripts/prof.sh run_cfg3b --workload cfg3b_4k_grey_to_512_lanczos | head -8
