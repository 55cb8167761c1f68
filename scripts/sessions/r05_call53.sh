#!/bin/bash
# r05 call 53: does a process that finds no fast destination placement among 10 candidates find one among 30?  bench.py --place 30 in three fresh processes and
# once under rocprofv3 --kernel-trace (call 48's profiled process had all ten candidates at 2.17-2.18 ms)
export PYTHONPATH=. TMPDIR=/tmp
root=$(pwd); o=$root/gpurun_out/r05c; mkdir -p $o
for i in 1 2 3; do python bench.py --place 30 --no-cpu-baseline > $o/place30_$i.json 2> $o/place30_$i.err || exit 1; done
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $o/place30_trace -- python3 $root/bench.py --place 30 --steps 20 --warmup 3 --no-cpu-baseline > $o/place30_prof.json 2> $o/place30_prof.err
cd $root
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05c/place30_*.json')):
    l = [x for x in open(f) if x.startswith('{"metric"')]
    if not l: print(f, 'no line'); continue
    d = json.loads(l[0]); c = d['config']['destination_placement']
    print(f.split('/')[-1], 'ms_per_step', d['ms_per_step'], 'kept', c['kept'], 'candidates ms', ' '.join('%.3f' % x[1] for x in c['candidates']))
PY
