#!/bin/bash
# run bench.py over kernel knobs; one line per configuration.  usage: bench_matrix.sh [extra bench args]
out=gpurun_out/bench_matrix.txt; : > $out
for num in 0 1; do for path in 2 3; do for pl in 1 2 4; do for ch in 0 ; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --option numerics=$num --option per_lane=$pl --option path=$path --option row_chunks=$ch "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-52s chunks=$ch ms/step %.4f us/img %.2f frac %.4f parity %s' % (d['config']['kernel'], d['ms_per_step'], d['ms_per_step']*1000/d['config']['images_per_gpu'], d['roofline']['frac'], d['config']['parity_max_code_diff_vs_oracle_image0']))
" >> $out
done; done; done; done
cat $out
