#!/bin/bash
# r03: resize_up_kernel (integer upscales, register ring) against the kernels that served 2x upscales: 960x540 -> 1080p, 512 and 64 images
mkdir -p gpurun_out
: > gpurun_out/up3.txt
for px in grey greya rgb rgba r16 r16g16 r16g16b16 r16g16b16a16; do
  echo "## $px" >> gpurun_out/up3.txt
  n=512; case $px in r16g16b16|r16g16b16a16) n=256;; esac
  timeout -k 10 200 python scripts/ab_opts.py '[{"run":2},{"run":2,"numerics":1},{"run":1},{"run":1,"numerics":1},{"run":1,"per_lane":4},{"run":1,"per_lane":2}]' '{}' $n:$px:960:540:1920:1080 3 8 >> gpurun_out/up3.txt 2>&1 || echo FAILED >> gpurun_out/up3.txt
  timeout -k 10 200 python scripts/ab_opts.py '[{"run":2},{"run":1}]' '{}' 64:$px:960:540:1920:1080 3 8 >> gpurun_out/up3.txt 2>&1 || echo FAILED >> gpurun_out/up3.txt
done
grep -v amdgpu gpurun_out/up3.txt
timeout -k 10 300 python -m pytest tests/test_gpu_chain_and_helpers.py -q -k "integer_upscale" 2>&1 | tail -12
