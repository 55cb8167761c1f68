#!/bin/bash
# r03: run kernel for grey / greya at more periodic scales (picha defaults): 4K -> 1536x864 (2.5x), 1280x720 (3x), 960x540 (4x), 768x432 (5x); 128 images
mkdir -p gpurun_out
: > gpurun_out/scales.txt
for geo in 1536:864 1280:720 960:540 768:432; do
for px in grey greya; do
  echo "## $px -> $geo" >> gpurun_out/scales.txt
  timeout -k 10 200 python scripts/ab_opts.py '[{"run":2},{"run":2,"numerics":1},{"run":1,"per_lane":8},{"run":1,"per_lane":8,"numerics":1},{"run":1,"per_lane":4},{"run":1,"per_lane":4,"numerics":1}]' '{}' 128:$px:3840:2160:$geo 3 8 >> gpurun_out/scales.txt 2>&1 || echo FAILED >> gpurun_out/scales.txt
done
done
grep -v amdgpu gpurun_out/scales.txt
