#!/bin/bash
# round-4 evidence in one GPU call: bench line (+ cpu baseline), every workload, rocprofv3 trace + FULL PMC passes for the headline (both
# numerics flavours: the fma pass feeds profiles/traffic_*.fma.json) and the cfg5 / cfg3 / cfg3b / cfg3a kernels, in-kernel clocks (probe builds),
# VALU issue models.  Outputs under gpurun_out/r04/final/.
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r04/final; mkdir -p $o
python bench.py > $o/bench_line.json 2> $o/bench_line.err
python bench.py --numerics fma --no-cpu-baseline > $o/bench_line_fma.json 2> $o/bench_line_fma.err
BENCH_OUT=$o/bench_workloads.txt bash scripts/bench_workloads.sh > /dev/null 2>&1
echo "bench done"
bash scripts/prof.sh r04f_cfg2 > $o/prof_cfg2.log 2>&1
bash scripts/prof.sh r04f_cfg2fma --numerics fma > $o/prof_cfg2fma.log 2>&1
bash scripts/prof.sh r04f_cfg5 --workload cfg5_8k_rgba16_to_1080p_mitchel > $o/prof_cfg5.log 2>&1
echo "prof cfg2 cfg5 done"
bash scripts/prof.sh r04f_cfg3 --workload cfg3_4k_rgb_to_grey_to_512_lanczos_chained > $o/prof_cfg3.log 2>&1
bash scripts/prof.sh r04f_cfg3b --workload cfg3b_4k_grey_to_512_lanczos > $o/prof_cfg3b.log 2>&1
bash scripts/prof.sh r04f_cfg3a --workload cfg3a_4k_rgb_to_grey > $o/prof_cfg3a.log 2>&1
echo "prof cfg3 done"
for t in cfg2 cfg2fma cfg5 cfg3 cfg3b cfg3a; do cp gpurun_out/prof_r04f_$t/summary.txt $o/prof_${t}_summary.txt 2>/dev/null; cp gpurun_out/prof_r04f_$t/trace/*/*kernel_stats.csv $o/prof_${t}_kernel_stats.csv 2>/dev/null; done
python scripts/make_traffic.py $o/prof_cfg2_summary.txt $o/bench_line.json 4 > $o/traffic_cfg2.json 2> $o/traffic.err
python scripts/make_traffic.py $o/prof_cfg2fma_summary.txt $o/bench_line_fma.json 4 > $o/traffic_cfg2_fma.json 2>> $o/traffic.err
PICHA_HIP_LIB=picha_amd/variants/libpicha_hip_probe.so timeout -k 10 500 python scripts/r04_clock_probe.py > $o/clock.txt 2>&1
rocm-smi --showserial --showuniqueid > $o/device.txt 2>&1
cat $o/bench_line.json | cut -c1-600; cat $o/bench_workloads.txt; grep -v amdgpu.ids $o/clock.txt | cut -c1-200
