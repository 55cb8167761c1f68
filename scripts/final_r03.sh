#!/bin/bash
# round-3 evidence in one GPU call: bench line (+ cpu baseline), every workload, rocprofv3 trace + FULL PMC passes for the headline
# and the cfg5 / cfg3 / cfg3b / cfg3a kernels.  Outputs under gpurun_out/r03/final/.
o=gpurun_out/r03/final; mkdir -p $o
python bench.py > $o/bench_line.json 2> $o/bench_line.err
BENCH_OUT=$o/bench_workloads.txt bash scripts/bench_workloads.sh > /dev/null 2>&1
bash scripts/prof.sh r03f_cfg2 > $o/prof_cfg2.log 2>&1
bash scripts/prof.sh r03f_cfg5 --workload cfg5_8k_rgba16_to_1080p_mitchel > $o/prof_cfg5.log 2>&1
bash scripts/prof.sh r03f_cfg3 --workload cfg3_4k_rgb_to_grey_to_512_lanczos_chained > $o/prof_cfg3.log 2>&1
bash scripts/prof.sh r03f_cfg3b --workload cfg3b_4k_grey_to_512_lanczos > $o/prof_cfg3b.log 2>&1
bash scripts/prof.sh r03f_cfg3a --workload cfg3a_4k_rgb_to_grey > $o/prof_cfg3a.log 2>&1
for t in cfg2 cfg5 cfg3 cfg3b cfg3a; do cp gpurun_out/prof_r03f_$t/summary.txt $o/prof_${t}_summary.txt 2>/dev/null; cp gpurun_out/prof_r03f_$t/trace/*/*kernel_stats.csv $o/prof_${t}_kernel_stats.csv 2>/dev/null; done
python scripts/make_traffic.py $o/prof_cfg2_summary.txt $o/bench_line.json 3 > $o/traffic_cfg2.json 2> $o/traffic.err
cat $o/bench_line.json; cat $o/bench_workloads.txt; cat $o/traffic_cfg2.json
