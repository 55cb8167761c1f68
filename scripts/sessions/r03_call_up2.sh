#!/bin/bash
# r03: row chunks of the 2x upscale launches (64 x 960x540 -> 1080p): fewer, longer chunks against the default
mkdir -p gpurun_out
: > gpurun_out/up2.txt
for px in grey rgb rgba; do
  echo "## $px" >> gpurun_out/up2.txt
  timeout -k 10 200 python scripts/ab_opts.py '[{},{"row_chunks":8},{"row_chunks":17},{"row_chunks":34},{"row_chunks":68},{"row_chunks":135}]' '{}' 64:$px:960:540:1920:1080 3 10 >> gpurun_out/up2.txt 2>&1
done
grep -v amdgpu gpurun_out/up2.txt
