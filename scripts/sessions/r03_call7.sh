#!/bin/bash
o=gpurun_out/r03/call7; mkdir -p $o
bash scripts/prof.sh r03b_cfg5 --workload cfg5_8k_rgba16_to_1080p_mitchel > $o/prof_cfg5.log 2>&1
cp gpurun_out/prof_r03b_cfg5/summary.txt $o/prof_cfg5_summary.txt
python scripts/ab_opts.py '[{}, {"row_chunks":12}, {"row_chunks":20}, {"row_chunks":24}, {"work":2,"work_tail":400}, {"work":2,"work_tail_div":8}, {"work":2,"work_tail":0}]' '{"filter":"mitchel"}' 32:r16g16b16a16:7680:4320:1920:1080 3 10 > $o/ab_cfg5.txt 2>&1
bash scripts/bench_workloads.sh > /dev/null 2>&1; cp gpurun_out/bench_workloads.txt $o/bench_workloads.txt
cat $o/ab_cfg5.txt $o/bench_workloads.txt; head -50 $o/prof_cfg5_summary.txt
