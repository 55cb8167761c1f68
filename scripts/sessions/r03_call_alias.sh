#!/bin/bash
# r03: ring-aliased tables (lanczos / box at even integer scales): accumulator kernel following the row list (default now) against
# the LDS-ring span kernel ("acc" = 2) that served them before; 64 x 4K sources
mkdir -p gpurun_out
: > gpurun_out/alias_survey.txt
for geo in 1920:1080 960:540 640:360 480:270; do
for px in grey rgb rgba r16g16b16a16; do
  for f in '{"filter":"lanczos"}' '{"filter":"box"}'; do
    echo "## $px -> $geo $f" >> gpurun_out/alias_survey.txt
    timeout -k 10 200 python scripts/ab_opts.py '[{"acc":2},{},{"acc":2,"numerics":1},{"numerics":1}]' "$f" 64:$px:3840:2160:$geo 3 6 >> gpurun_out/alias_survey.txt 2>&1 || echo "FAILED $px $f" >> gpurun_out/alias_survey.txt
  done
done
done
grep -v amdgpu.ids gpurun_out/alias_survey.txt
