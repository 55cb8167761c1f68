#!/bin/bash
# r03: ring-in-registers kernel on ring-aliased lanczos downscales (4K -> 1080p: 2x; -> 960x540: 4x) against the kernels that served them
mkdir -p gpurun_out
: > gpurun_out/ring.txt
for px in grey greya rgb rgba r16 r16g16 r16g16b16 r16g16b16a16; do
  echo "## lanczos 2x $px" >> gpurun_out/ring.txt
  n=64; case $px in r16g16b16|r16g16b16a16) n=32;; esac
  timeout -k 10 200 python scripts/ab_opts.py '[{"run":2},{"run":1},{"run":2,"numerics":1},{"run":1,"numerics":1}]' '{"filter":"lanczos"}' $n:$px:3840:2160:1920:1080 3 6 >> gpurun_out/ring.txt 2>&1 || echo FAILED >> gpurun_out/ring.txt
done
for px in grey greya; do
  echo "## lanczos 4x $px" >> gpurun_out/ring.txt
  timeout -k 10 200 python scripts/ab_opts.py '[{"run":2},{"run":2,"acc":1},{"run":1},{"run":2,"numerics":1},{"run":1,"numerics":1}]' '{"filter":"lanczos"}' 64:$px:3840:2160:960:540 3 6 >> gpurun_out/ring.txt 2>&1 || echo FAILED >> gpurun_out/ring.txt
done
grep -v amdgpu gpurun_out/ring.txt
