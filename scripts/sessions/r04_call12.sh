#!/bin/bash
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_chain_and_helpers.py -x -q -m gpu -k "neighbour_exchange or scalar_phase" > gpurun_out/r04/call12_parity.txt 2>&1; echo "parity rc=$?"
tail -3 gpurun_out/r04/call12_parity.txt
timeout -k 10 300 python scripts/r04_dbg_phase3.py > gpurun_out/r04/call12_dbg3.txt 2>&1; grep -c OK gpurun_out/r04/call12_dbg3.txt; grep -c BAD gpurun_out/r04/call12_dbg3.txt
timeout -k 10 400 python scripts/ab_opts.py '[{"run_xch":2},{"run_phase":-1},{"run_phase":11},{"run_phase":12},{"run_phase":13},{"numerics":1,"run_xch":2},{"numerics":1,"run_phase":-1},{"numerics":1,"run_phase":12}]' '{"filter":"mitchel"}' 32:r16g16b16a16:7680:4320:1920:1080 4 10 > gpurun_out/r04/call12_ab_cfg5.txt 2>&1; echo "cfg5 rc=$?"
cat gpurun_out/r04/call12_ab_cfg5.txt
