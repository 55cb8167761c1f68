#!/bin/bash
run() { timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-60s %-40s %.4f ms frac %.3f parity %s | fma %.4f ms' % (d['roofline']['kernel'][:60], ' '.join(sys.argv[3:]), d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['parity_max_code_diff_vs_oracle_image0'], d.get('other_numerics',{}).get('avg_launch_ms',0)))
" "$@"; }
for c in 0 5 8 10 16 32; do run --workload cfg3b_4k_grey_to_512_lanczos --option row_chunks=$c; done
for c in 0 5 8 10 16; do run --workload cfg3_4k_rgb_to_grey_to_512_lanczos_chained --option row_chunks=$c; done
