#!/bin/bash
# r03: run kernel survey over formats, 4K -> 512x288 (7.5x both ways), 21 taps (picha defaults) and 31 taps (lanczos)
mkdir -p gpurun_out
: > gpurun_out/run5_survey.txt
for px in grey greya rgb rgba r16; do
  for f in '{}' '{"filter":"lanczos"}'; do
    echo "## $px $f" >> gpurun_out/run5_survey.txt
    timeout -k 10 200 python scripts/ab_opts.py '[{"run":2},{"run":1,"work":3},{"run":1,"work":3,"per_lane":2},{"run":2,"numerics":1},{"run":1,"work":3,"numerics":1},{"run":1,"work":3,"per_lane":2,"numerics":1}]' "$f" 128:$px:3840:2160:512:288 3 8 >> gpurun_out/run5_survey.txt 2>&1 || echo "FAILED $px $f" >> gpurun_out/run5_survey.txt
  done
done
grep -v amdgpu.ids gpurun_out/run5_survey.txt
