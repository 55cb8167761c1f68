#!/bin/bash
# r04 call 14: where else does the neighbour-exchange run kernel (wave-uniform weights, optional phased stores) beat the planner's choice?
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/call14_survey_run_xch.txt; : > $out
V='[{},{"run_xch":1,"run_phase":-1},{"run_xch":1,"run_phase":12},{"numerics":1},{"numerics":1,"run_xch":1,"run_phase":-1},{"numerics":1,"run_xch":1,"run_phase":12}]'
echo "## rgba 4x default (11 taps), 128 x 4K -> 960x540" >> $out
timeout -k 10 300 python scripts/ab_opts.py "$V" '{}' 128:rgba:3840:2160:960:540 3 10 >> $out 2>&1
echo "## rgba 4x mitchel (15 taps)" >> $out
timeout -k 10 300 python scripts/ab_opts.py "$V" '{"filter":"mitchel"}' 128:rgba:3840:2160:960:540 3 10 >> $out 2>&1
echo "## rgba 3x default (8 taps), 128 x 4K -> 1280x720" >> $out
timeout -k 10 300 python scripts/ab_opts.py "$V" '{}' 128:rgba:3840:2160:1280:720 3 10 >> $out 2>&1
echo "## r16g16b16a16 4x default (11 taps), 64 x 4K -> 960x540" >> $out
timeout -k 10 300 python scripts/ab_opts.py "$V" '{}' 64:r16g16b16a16:3840:2160:960:540 3 10 >> $out 2>&1
grep -v amdgpu.ids $out
