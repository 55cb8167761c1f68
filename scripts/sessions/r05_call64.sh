#!/bin/bash
# r05 call 64: the full-size test of the long-window thumbnails at the planner's new chunking; then the whole GPU suite once more
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05c; mkdir -p $o
timeout -k 10 600 python -m pytest tests -m gpu -q > $o/gputest_final.log 2>&1; rc=$?; tail -4 $o/gputest_final.log | cut -c1-300
exit $rc
