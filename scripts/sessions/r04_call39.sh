#!/bin/bash
# r04 call 39: neighbour-exchange form of the run kernel with the window assembled through LDS (every lane writes its own dwords, reads its halo dwords) instead of 22 DPP moves
# + 22 broadcast reads per row (-DPICHA_RUN_XLDS), against the in-tree DPP form: cfg5, r16g16b16a16 / rgba 4x
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/call39_xlds.txt; : > $out
for rnd in 1 2 3; do for lib in "" picha_amd/variants/libpicha_hip_xlds.so; do
  echo "== round $rnd lib=${lib:-in-tree}" >> $out
  PICHA_HIP_LIB=$lib timeout -k 10 200 python scripts/ab_opts.py '[{}, {"numerics": 1}, {"run_phase": -1}]' '{"filter": "mitchel"}' 32:r16g16b16a16:7680:4320:1920:1080 2 10 2>/dev/null | grep -v "^#" >> $out
  PICHA_HIP_LIB=$lib timeout -k 10 200 python scripts/ab_opts.py '[{"numerics": 1}]' '' 64:rgba:7680:4320:1920:1080 2 10 2>/dev/null | grep -v "^#" >> $out
done; done
cat $out
