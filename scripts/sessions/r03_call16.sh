#!/bin/bash
# usage: [LIBS="a.so b.so"] [OUTNAME=name] scripts/r03_call16.sh : cfg2 bench per library, alternating processes, 6 rounds
o=gpurun_out/r03/call16; mkdir -p $o
for rnd in 1 2 3 4 5 6; do for lib in ${LIBS:-picha_amd/libpicha_hip_med3.so picha_amd/libpicha_hip.so}; do
  PICHA_HIP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('round $rnd %-28s exact %.4f ms frac %.4f parity %s | fma %.4f' % ('$(basename $lib)', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['parity_max_code_diff_vs_oracle_image0'], d['other_numerics']['avg_launch_ms']))
"
done; done > $o/${OUTNAME:-ab_pack_cfg2_6rounds}.txt 2>&1
cat $o/${OUTNAME:-ab_pack_cfg2_6rounds}.txt
