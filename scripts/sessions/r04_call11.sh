#!/bin/bash
# r04 call 11: time-phased stores in the real cfg5 kernel: parity, then phase length sweep against the unphased form and the r03 kernel
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_chain_and_helpers.py -x -q -m gpu -k "neighbour_exchange or scalar_phase" > gpurun_out/r04/call11_parity.txt 2>&1; echo "parity rc=$?"
tail -3 gpurun_out/r04/call11_parity.txt
timeout -k 10 400 python scripts/ab_opts.py '[{"run_xch":2},{"run_phase":-1},{"run_phase":10},{"run_phase":11},{"run_phase":12},{"run_phase":13},{"run_phase":14},{"run_phase":12,"block_waves":2},{"run_phase":12,"row_chunks":12},{"numerics":1,"run_xch":2},{"numerics":1,"run_phase":-1},{"numerics":1,"run_phase":12}]' '{"filter":"mitchel"}' 32:r16g16b16a16:7680:4320:1920:1080 4 10 > gpurun_out/r04/call11_ab_cfg5.txt 2>&1; echo "cfg5 rc=$?"
cat gpurun_out/r04/call11_ab_cfg5.txt
