#!/bin/bash
# r05 call 55: is the bench line a transient?  The same bench.py at 50 (default), 500 and 3000 timed steps (0.1 / 1 / 6 s of back-to-back launches), both flavours, fresh processes
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05c; mkdir -p $o
for num in exact fma; do for k in 50 500 3000 50; do
python bench.py --numerics $num --steps $k --warmup 5 --no-cpu-baseline > $o/sustain_${num}_$k.json 2> $o/sustain_${num}_$k.err || exit 1
python3 - $o/sustain_${num}_$k.json $num $k <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{"metric"')][0]); c = d['config']['destination_placement']
print(sys.argv[2], 'steps', sys.argv[3], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'kept candidate trial', c['candidates'][c['kept']][1], 'median/min launch', d['roofline'].get('median_launch_ms'), d['roofline'].get('min_launch_ms'))
PY
done; done
