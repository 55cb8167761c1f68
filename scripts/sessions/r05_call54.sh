#!/bin/bash
# r05 call 54 (the same as call 48, on the library of the round's last commit: edge_rows knob, RunParams grown): the whole GPU suite on the round's HEAD library, then scripts/final_r05b.sh stages 1-3 on the same
# box (bench line + cpu baseline, fma line, every workload, rocprofv3 trace + PMC passes of the BASELINE stages and the cells that changed since the last evidence run:
# the streaming kernel's whole-round chunks for small images, grey / greya default-filter thumbnails in the shared-unpack form)
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05c; mkdir -p $o
python -m pytest tests -m gpu -q > $o/gputest_head.log 2>&1; rc=$?; tail -5 $o/gputest_head.log | cut -c1-250
[ $rc = 0 ] || exit $rc
bash scripts/final_r05b.sh 1 && bash scripts/final_r05b.sh 2 && bash scripts/final_r05b.sh 3
