#!/bin/bash
# r03 call 1: baseline bench on this box, cvt semantics probe, full PMC passes for cfg5 and cfg3a
o=gpurun_out/r03/call1; mkdir -p $o
scripts/ubench/cvt_sem > $o/cvt_sem.txt 2>&1
python bench.py --no-cpu-baseline > $o/bench_base.json 2> $o/bench_base.err
bash scripts/prof.sh r03_cfg5 --workload cfg5_8k_rgba16_to_1080p_mitchel > $o/prof_cfg5.log 2>&1
cp gpurun_out/prof_r03_cfg5/summary.txt $o/prof_cfg5_summary.txt
bash scripts/prof.sh r03_cfg3a --workload cfg3a_4k_rgb_to_grey > $o/prof_cfg3a.log 2>&1
cp gpurun_out/prof_r03_cfg3a/summary.txt $o/prof_cfg3a_summary.txt
cat $o/cvt_sem.txt $o/bench_base.json
