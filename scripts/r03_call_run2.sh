#!/bin/bash
# r03: kernel-trace of the run kernel + point kernel on cfg3b
set -e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_run2 -o run2 -- python scripts/ab_opts.py '[{"run":0}]' '{"filter":"lanczos"}' 256:grey:3840:2160:512:512 2 10 > gpurun_out/run2_log.txt 2>&1
python scripts/prof_summary.py gpurun_out/prof_run2 > gpurun_out/run2_summary.txt 2>&1 || true
tail -5 gpurun_out/run2_log.txt
head -30 gpurun_out/run2_summary.txt
