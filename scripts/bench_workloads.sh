#!/bin/bash
out=gpurun_out/bench_workloads.txt; : > $out
for w in cfg2_4k_rgba_to_1080p_cubic cfg2_4k_rgba_to_1080p_cubic_scale1 cfg1_640x480_rgba_to_320x240 cfg3a_4k_rgb_to_grey cfg3b_4k_grey_to_512_lanczos cfg3_4k_rgb_to_grey_to_512_lanczos_chained cfg5_8k_rgba16_to_1080p_mitchel cc_4k_rgba_to_rgba16; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload $w "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-38s %-50s ms/step %8.4f us/img %8.2f GB/s %7.1f frac %.4f' % ('$w', d['config']['kernel'], d['ms_per_step'], d['ms_per_step']*1000/d['config']['images_per_gpu'], d['roofline']['achieved'], d['roofline']['frac']))
" >> $out
done
cat $out
