#!/bin/bash
# r04 call 2: ablations of the run kernel (cfg5 xch form, cfg3b) — builds with the arithmetic or the loads removed
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/call2_ablate.txt; : > $out
for rnd in 1 2; do
for lib in "" picha_amd/variants/libpicha_hip_ab1.so picha_amd/variants/libpicha_hip_ab2.so picha_amd/variants/libpicha_hip_ab3.so; do
  echo "== round $rnd lib=${lib:-in-tree}" >> $out
  PICHA_HIP_LIB=$lib timeout -k 10 200 python scripts/ab_opts.py '[{},{"numerics":1}]' '{"filter":"mitchel"}' 32:r16g16b16a16:7680:4320:1920:1080 2 10 2>/dev/null >> $out
  [ "$lib" != "picha_amd/variants/libpicha_hip_ab3.so" ] && PICHA_HIP_LIB=$lib timeout -k 10 200 python scripts/ab_opts.py '[{},{"numerics":1}]' '{"filter":"lanczos"}' 256:grey:3840:2160:512:512 2 10 2>/dev/null >> $out
done; done
cat $out
