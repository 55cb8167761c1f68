#!/bin/bash
# copies what scripts/final_r04.sh left under gpurun_out/r04/final into profiles/ (run here, after the GPU call)
o=gpurun_out/r04/final
cp $o/bench_line.json profiles/r04_bench_line.json; cp $o/bench_line_fma.json profiles/r04_bench_line_fma.json; cp $o/bench_workloads.txt profiles/r04_bench_workloads.txt; cp $o/device.txt profiles/r04_device.txt
for t in cfg2 cfg2fma cfg5 cfg3 cfg3b cfg3a; do cp $o/prof_${t}_summary.txt profiles/r04_${t}_rocprofv3_summary.txt; done
for t in cfg2 cfg5 cfg3b; do cp $o/prof_${t}_kernel_stats.csv profiles/r04_${t}_kernel_stats.csv; done
cp $o/traffic_cfg2.json profiles/traffic_cfg2_4k_rgba_to_1080p_cubic.json; cp $o/traffic_cfg2_fma.json profiles/traffic_cfg2_4k_rgba_to_1080p_cubic.fma.json
cat $o/traffic.err; grep "Serial\|Unique" $o/device.txt
