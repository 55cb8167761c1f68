#!/bin/bash
# r03: run kernel at integer scales with long windows: 4K -> 640x360 (6x: 17 / 25 taps) and -> 480x270 (8x: 23 / 33 taps)
mkdir -p gpurun_out
: > gpurun_out/run6_survey.txt
for geo in 640:360 480:270; do
for px in grey greya rgb rgba; do
  for f in '{}' '{"filter":"lanczos"}'; do
    echo "## $px -> $geo $f" >> gpurun_out/run6_survey.txt
    case $px in grey) pls="4 8";; greya) pls="2 4";; rgb) pls="2 4";; *) pls="2";; esac
    v='[{"run":2},{"run":2,"numerics":1}'
    for pl in $pls; do v="$v,{\"run\":1,\"per_lane\":$pl},{\"run\":1,\"per_lane\":$pl,\"numerics\":1}"; done
    v="$v]"
    timeout -k 10 200 python scripts/ab_opts.py "$v" "$f" 128:$px:3840:2160:$geo 3 8 >> gpurun_out/run6_survey.txt 2>&1 || echo "FAILED $px $f" >> gpurun_out/run6_survey.txt
  done
done
done
grep -v amdgpu.ids gpurun_out/run6_survey.txt
