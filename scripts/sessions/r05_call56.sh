#!/bin/bash
# r05 call 56: the driver's own round-end commands on the round's last commit: smoke(), the plain bench line (with the cpu baseline), and bench.py --gpus 2 over gloo on the one GPU
export TMPDIR=/tmp
o=gpurun_out/r05c; mkdir -p $o
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.log 2>&1; rc=$?; tail -2 $o/smoke.log; [ $rc = 0 ] || exit $rc
( time python bench.py ) > $o/driver_bench.json 2> $o/driver_bench.err; rc=$?; cut -c1-250 $o/driver_bench.json; tail -4 $o/driver_bench.err; [ $rc = 0 ] || exit $rc
PICHA_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $o/driver_bench_2ranks_gloo.json 2> $o/driver_bench_2ranks_gloo.err; rc=$?
python3 -c "
import json; d=json.loads([x for x in open('$o/driver_bench_2ranks_gloo.json') if x.startswith('{\"metric\"')][0]); print('n_gpus', d['n_gpus'], 'ranks_seen', d['config'].get('ranks_seen'), 'ms', d['ms_per_step'])"
exit $rc
