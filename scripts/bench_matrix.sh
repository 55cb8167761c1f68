#!/bin/bash
# run bench.py over kernel knobs; one JSON line per configuration
out=gpurun_out/bench_matrix.txt; : > $out
for num in 0 1; do for pl in 1 2 4; do
  echo "== numerics=$num per_lane=$pl" >> $out
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --option numerics=$num --option per_lane=$pl 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['kernel'], 'ms/step', d['ms_per_step'], 'us/img', round(d['ms_per_step']*1000/256,2), 'GB/s', d['roofline']['achieved'], 'frac', d['roofline']['frac'], 'read_frac', d['roofline']['read_only_frac'])
" >> $out
done; done
cat $out
