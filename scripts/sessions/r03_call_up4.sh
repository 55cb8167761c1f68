#!/bin/bash
# r03: resize_up_kernel at 4x (480x270 -> 1080p) and r16g16b16a16 at 2x with four outputs per lane
mkdir -p gpurun_out
: > gpurun_out/up4.txt
for px in grey greya rgb rgba r16; do
  echo "## 4x $px" >> gpurun_out/up4.txt
  timeout -k 10 200 python scripts/ab_opts.py '[{"run":2},{"run":1},{"run":2,"numerics":1},{"run":1,"numerics":1}]' '{}' 512:$px:480:270:1920:1080 3 8 >> gpurun_out/up4.txt 2>&1 || echo FAILED >> gpurun_out/up4.txt
done
echo "## 2x r16g16b16a16" >> gpurun_out/up4.txt
timeout -k 10 200 python scripts/ab_opts.py '[{"run":2},{"run":1},{"run":2,"numerics":1},{"run":1,"numerics":1}]' '{}' 256:r16g16b16a16:960:540:1920:1080 3 8 >> gpurun_out/up4.txt 2>&1
grep -v amdgpu gpurun_out/up4.txt
