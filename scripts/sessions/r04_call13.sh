#!/bin/bash
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/call13.txt; : > $out
echo "## cfg3b: scalar weights for 0 / 1 / 2 phases, 4 / 8 outputs per lane, chunks" >> $out
timeout -k 10 400 python scripts/ab_opts.py '[{"run_ws":2},{},{"run_ws":3},{"per_lane":8},{"per_lane":8,"run_ws":3},{"row_chunks":8},{"row_chunks":8,"run_ws":3},{"numerics":1,"run_ws":2},{"numerics":1},{"numerics":1,"run_ws":3},{"numerics":1,"per_lane":8},{"numerics":1,"row_chunks":8}]' '{"filter":"lanczos"}' 256:grey:3840:2160:512:512 4 10 >> $out 2>&1
for v in nb2; do
echo "## cfg3b, variant build $v" >> $out
PICHA_HIP_LIB=picha_amd/variants/libpicha_hip_$v.so timeout -k 10 400 python scripts/ab_opts.py '[{"run_ws":2},{},{"run_ws":3},{"row_chunks":8},{"row_chunks":8,"run_ws":3},{"numerics":1},{"numerics":1,"run_ws":3}]' '{"filter":"lanczos"}' 256:grey:3840:2160:512:512 4 10 >> $out 2>&1
done
for v in pr3 pr4; do
echo "## cfg5, variant build $v (phase buffer rows)" >> $out
PICHA_HIP_LIB=picha_amd/variants/libpicha_hip_$v.so timeout -k 10 400 python scripts/ab_opts.py '[{"run_phase":-1},{"run_phase":11},{"run_phase":12},{"numerics":1,"run_phase":12}]' '{"filter":"mitchel"}' 32:r16g16b16a16:7680:4320:1920:1080 3 10 >> $out 2>&1
done
echo "## cfg5 in-tree (8 rows)" >> $out
timeout -k 10 400 python scripts/ab_opts.py '[{"run_phase":-1},{"run_phase":11},{"run_phase":12},{"numerics":1,"run_phase":12}]' '{"filter":"mitchel"}' 32:r16g16b16a16:7680:4320:1920:1080 3 10 >> $out 2>&1
echo "## cfg2 fma: outputs per lane" >> $out
timeout -k 10 400 python scripts/ab_opts.py '[{"numerics":1},{"numerics":1,"per_lane":3},{"numerics":1,"per_lane":4},{},{"per_lane":2},{"per_lane":4}]' '{}' 256:rgba:3840:2160:1920:1080 3 10 >> $out 2>&1
grep -v amdgpu.ids $out
