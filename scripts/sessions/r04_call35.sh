#!/bin/bash
# r04 call 35: upper bound of handing UNPACKED pixels between lanes in the symmetric-table kernel: a build whose halo pixels cost no unpack (-DPICHA_SYM_ABLATE_HALO, wrong results)
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/call35_sym_halo_ablate.txt; : > $out
for rnd in 1 2 3; do for lib in "" picha_amd/variants/libpicha_hip_symhalo.so; do
  echo "== round $rnd lib=${lib:-in-tree}" >> $out
  PICHA_HIP_LIB=$lib timeout -k 10 200 python scripts/ab_opts.py '[{}, {"per_lane": 2}, {"per_lane": 4}, {"numerics": 1}]' '' 256:rgba:3840:2160:1920:1080 2 10 2>/dev/null | grep -v "^#" >> $out
done; done
cat $out
