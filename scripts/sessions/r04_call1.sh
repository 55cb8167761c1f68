#!/bin/bash
# r04 call 1: parity of the new run-kernel forms, then interleaved A/Bs on cfg5 and cfg3b
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_chain_and_helpers.py -x -q -m gpu -k "neighbour_exchange or scalar_phase" > gpurun_out/r04/call1_parity.txt 2>&1; echo "parity rc=$?" 
tail -5 gpurun_out/r04/call1_parity.txt
timeout -k 10 300 python scripts/ab_opts.py '[{"run_xch":2},{},{"work":1},{"work":2},{"block_waves":2},{"block_waves":1},{"row_chunks":12},{"row_chunks":27},{"numerics":1,"run_xch":2},{"numerics":1}]' '{"filter":"mitchel"}' 32:r16g16b16a16:7680:4320:1920:1080 3 10 > gpurun_out/r04/call1_ab_cfg5.txt 2>&1; echo "cfg5 rc=$?"
cat gpurun_out/r04/call1_ab_cfg5.txt
timeout -k 10 300 python scripts/ab_opts.py '[{"run_ws":2},{},{"run_ws":2,"numerics":1},{"numerics":1},{"row_chunks":8},{"row_chunks":16},{"row_chunks":32},{"block_waves":1},{"work":1}]' '{"filter":"lanczos"}' 256:grey:3840:2160:512:512 3 10 > gpurun_out/r04/call1_ab_cfg3b.txt 2>&1; echo "cfg3b rc=$?"
cat gpurun_out/r04/call1_ab_cfg3b.txt
