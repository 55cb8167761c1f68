#!/bin/bash
# r04 call 38: cfg3b, upper bound of sharing the unpack between neighbouring lanes (-DPICHA_RUN_ABLATE=5: window pixels outside the lane's own 30 cost no unpack; wrong results)
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/call38_cfg3b_unpack_ablate.txt; : > $out
for rnd in 1 2 3; do for lib in "" picha_amd/variants/libpicha_hip_ab5.so; do
  echo "== round $rnd lib=${lib:-in-tree}" >> $out
  PICHA_HIP_LIB=$lib timeout -k 10 200 python scripts/ab_opts.py '[{}, {"numerics": 1}]' '{"filter": "lanczos"}' 256:grey:3840:2160:512:512 2 10 2>/dev/null | grep -v "^#" >> $out
done; done
cat $out
