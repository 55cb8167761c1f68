#!/bin/bash
# r05 call 21: GPU suite with the shared-unpack form and the measured edge-source defaults; cfg3b defaults against the r03 form
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/gputest_shu.log 2>&1 || { tail -40 $o/gputest_shu.log; exit 1; }
tail -2 $o/gputest_shu.log
{
for num in exact fma; do
python scripts/ab_shared.py --case cfg3b --numerics $num r03_form=.:run_edges=1 default=.
done
python scripts/ab_shared.py --case 256:grey:3840:2160:512:288 r03_form=.:run_edges=1 default=.
python scripts/ab_shared.py --case 128:greya:3840:2160:512:512:lanczos r03_form=.:run_edges=1 default=.
} > $o/ab_run_defaults.txt 2>&1
grep -v amdgpu.ids $o/ab_run_defaults.txt | cut -c1-250
