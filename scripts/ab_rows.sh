#!/bin/bash
# row-staged kernel vs the kernels it replaces: cfg3b (grey 4K -> 512^2 lanczos) and cfg3 as one call (rgb -> grey -> 512^2)
run() { timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-70s %-34s %.4f ms frac %.3f parity %s | fma %.4f ms' % (d['roofline']['kernel'][:70], ' '.join(sys.argv[1:]), d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['parity_max_code_diff_vs_oracle_image0'], d.get('other_numerics',{}).get('avg_launch_ms',0)))
    elif 'Error' in l or 'error' in l: print(l.strip())
" "$@"; }
for rnd in 1 2; do
run --workload cfg3b_4k_grey_to_512_lanczos --option rows=2
run --workload cfg3b_4k_grey_to_512_lanczos
run --workload cfg3b_4k_grey_to_512_lanczos --option per_lane=1
run --workload cfg3_4k_rgb_to_grey_to_512_lanczos_chained --option fuse=2
run --workload cfg3_4k_rgb_to_grey_to_512_lanczos_chained
run --workload cfg3_4k_rgb_to_grey_to_512_lanczos_chained --option per_lane=1
done
run --workload custom --shape 192:rgba:3840:2160:512:288 --option rows=2
run --workload custom --shape 192:rgba:3840:2160:512:288
run --workload custom --shape 192:rgba:3840:2160:512:288 --option per_lane=1
run --workload custom --shape 192:grey:3840:2160:512:288 --option rows=2
run --workload custom --shape 192:grey:3840:2160:512:288
