#!/bin/bash
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 400 python scripts/ab_opts.py '[{"run":2},{"run":1},{"run":1,"run_ws":2},{"run":1,"per_lane":4},{"run":2,"numerics":1},{"run":1,"numerics":1},{"run":1,"run_ws":2,"numerics":1}]' '{}' 192:rgb:3840:2160:512:288 3 10 > gpurun_out/r04/call24_rgb_thumb.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/call24_rgb_thumb.txt
