#!/bin/bash
# r04 call 18: the symmetric-table kernel's colour-convert epilogue — parity, then fused vs chained on 4K batches
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_chain_and_helpers.py -x -q -m gpu -k "epilogue" > gpurun_out/r04/call18_parity.txt 2>&1; echo "parity rc=$?"; tail -6 gpurun_out/r04/call18_parity.txt
timeout -k 10 600 python scripts/r04_ab_epilogue.py > gpurun_out/r04/call18_ab_epilogue.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04/call18_ab_epilogue.txt
