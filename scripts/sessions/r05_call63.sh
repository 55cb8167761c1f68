#!/bin/bash
# r05 call 63: closing check of the round's last library: the whole GPU suite, smoke(), the plain bench line
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05c; mkdir -p $o
timeout -k 10 600 python -m pytest tests -m gpu -q > $o/gputest_closing.log 2>&1; rc=$?; tail -3 $o/gputest_closing.log | cut -c1-300
[ $rc = 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $o/bench_line_closing.json 2> $o/bench_line_closing.err; rc=$?; cut -c1-220 $o/bench_line_closing.json
exit $rc
