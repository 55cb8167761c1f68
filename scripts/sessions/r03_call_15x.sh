#!/bin/bash
# r03: run kernel at 1.5x (1080p -> 720p: period 2 / 3 source pixels, 4 taps) against the accumulator kernel, 256 images
mkdir -p gpurun_out
: > gpurun_out/x15.txt
for px in grey greya rgb rgba; do
  echo "## 1.5x $px" >> gpurun_out/x15.txt
  case $px in grey|greya) pls="8 4";; *) pls="4 2";; esac
  v='[{"run":2},{"run":2,"numerics":1}'
  for pl in $pls; do v="$v,{\"run\":1,\"per_lane\":$pl},{\"run\":1,\"per_lane\":$pl,\"numerics\":1}"; done
  v="$v]"
  timeout -k 10 200 python scripts/ab_opts.py "$v" '{}' 256:$px:1920:1080:1280:720 3 8 >> gpurun_out/x15.txt 2>&1 || echo FAILED >> gpurun_out/x15.txt
done
grep -v amdgpu gpurun_out/x15.txt
