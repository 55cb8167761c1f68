#!/bin/bash
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 400 python scripts/ab_opts.py '[{"run_xch":2},{"run_phase":12,"row_chunks":18},{},{"numerics":1,"run_xch":2},{"numerics":1,"row_chunks":18},{"numerics":1}]' '{"filter":"mitchel"}' 32:r16g16b16a16:7680:4320:1920:1080 4 10 > gpurun_out/r04/call21_ab_cfg5.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/call21_ab_cfg5.txt
timeout -k 10 400 python scripts/ab_opts.py '[{"run_xch":2},{"row_chunks":16},{"row_chunks":12},{"row_chunks":8},{}]' '{"filter":"mitchel"}' 128:r16g16b16a16:7680:4320:1920:1080 3 5 > gpurun_out/r04/call21_ab_cfg5_128.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/call21_ab_cfg5_128.txt
timeout -k 10 300 python scripts/ab_opts.py '[{"run_xch":2},{"row_chunks":18},{}]' '{}' 64:r16g16b16a16:3840:2160:960:540 3 10 > gpurun_out/r04/call21_ab_r16_4x.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/call21_ab_r16_4x.txt
