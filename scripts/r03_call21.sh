#!/bin/bash
o=gpurun_out/r03/call21; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_chain_and_helpers.py -x -q > $o/pytest_chain.txt 2>&1 || { tail -40 $o/pytest_chain.txt; exit 1; }
tail -2 $o/pytest_chain.txt
sed -i "s/'work', 'work_tail', 'work_tail_div', 'up')/'work', 'work_tail', 'work_tail_div', 'up', 'rows_pairs')/" scripts/ab_opts.py
python scripts/ab_opts.py '[{"rows_pairs":2}, {}, {"rows_pairs":2,"row_chunks":8}, {"row_chunks":8}, {"row_chunks":2}, {"numerics":1,"rows_pairs":2}, {"numerics":1}]' '{"filter":"lanczos"}' 256:grey:3840:2160:512:512 3 10 > $o/ab_cfg3b_pairs.txt 2>&1
cat $o/ab_cfg3b_pairs.txt
for rnd in 1 2 3; do for pr in 2 0; do
  python bench.py --no-cpu-baseline --steps 40 --warmup 5 --workload cfg3_4k_rgb_to_grey_to_512_lanczos_chained --option rows_pairs=$pr 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('round $rnd rows_pairs=$pr %-70s exact %.4f ms frac %.4f parity %s | fma %.4f' % (d['roofline']['kernel'][:70], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['parity_max_code_diff_vs_oracle_image0'], d['other_numerics']['avg_launch_ms']))
"
done; done > $o/ab_cfg3_pairs.txt 2>&1
cat $o/ab_cfg3_pairs.txt
