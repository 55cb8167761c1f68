#!/bin/bash
# copies the outputs of scripts/final_r05b.sh (gpurun_out/r05/final_b/: the round's FINAL library) into profiles/.  The first session's bench line and workloads table
# (another box, the library before the shared-unpack / edges-inside work) stay as *.session1.*; its clocks, timelines, traffic / valu json describe cfg2's kernel, which did
# not change, and are kept as they are.
o=gpurun_out/r05/final_b; p=profiles
[ -e $p/r05_bench_line.session1.json ] || { cp $p/r05_bench_line.json $p/r05_bench_line.session1.json; cp $p/r05_bench_line_fma.json $p/r05_bench_line_fma.session1.json; cp $p/r05_bench_workloads.txt $p/r05_bench_workloads.session1.txt; }
cp $o/bench_line.json $p/r05_bench_line.json; cp $o/bench_line_fma.json $p/r05_bench_line_fma.json
{ echo "# scripts/bench_workloads.sh on the box of final_r05b.sh stage 1 (serial $(grep -i serial $o/device_stage1.txt | awk '{print $NF}' | head -1)), the round's final library: 20 timed launches after 3 warm-ups, bit-exact flavour | fma"; cat $o/bench_workloads.txt; } > $p/r05_bench_workloads.txt
for t in cfg2 cfg2fma cfg3 cfg3b cfg5; do
  [ -e $p/r05_${t}_rocprofv3_summary.session1.txt ] || cp $p/r05_${t}_rocprofv3_summary.txt $p/r05_${t}_rocprofv3_summary.session1.txt
  cp $o/prof_${t}_summary.txt $p/r05_${t}_rocprofv3_summary.txt
done
cp $o/prof_cfg2_kernel_stats.csv $p/r05_cfg2_kernel_stats.csv; cp $o/prof_cfg3b_kernel_stats.csv $p/r05_cfg3b_kernel_stats.csv; cp $o/prof_cfg5_kernel_stats.csv $p/r05_cfg5_kernel_stats.csv
for f in $o/prof_sec_*_summary.txt; do b=$(basename $f _summary.txt); cp $f $p/r05_${b#prof_}_rocprofv3_summary.txt; done
ls $p | grep -c r05_
