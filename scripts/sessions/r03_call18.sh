#!/bin/bash
o=gpurun_out/r03/call18; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_chain_and_helpers.py -x -q > $o/pytest_chain.txt 2>&1 || { tail -30 $o/pytest_chain.txt; exit 1; }
tail -2 $o/pytest_chain.txt
for rnd in 1 2 3 4; do for lib in picha_amd/libpicha_hip_clamp.so picha_amd/libpicha_hip.so; do
  PICHA_HIP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 60 --warmup 10 --workload cfg3_4k_rgb_to_grey_to_512_lanczos_chained 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('round $rnd %-28s exact %.4f ms frac %.4f parity %s | fma %.4f' % ('$(basename $lib)', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['parity_max_code_diff_vs_oracle_image0'], d['other_numerics']['avg_launch_ms']))
"
done; done > $o/ab_noclamp_cfg3.txt 2>&1
cat $o/ab_noclamp_cfg3.txt
