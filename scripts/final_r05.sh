#!/bin/bash
# usage: final_r05.sh <stage 1|2|3|4>   (a gpurun call is at most 20 minutes: the evidence is taken in four)
# round-5 evidence: bench line (+ cpu baseline), every workload, rocprofv3 trace + PMC passes for the BASELINE stages and the secondary cells, in-kernel clocks
# (probe build), VALU utilisation.  Outputs under gpurun_out/r05/final/ (copied to profiles/ by hand afterwards).
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05/final; mkdir -p $o
bash scripts/unpack_variants.sh > /dev/null
stage=${1:-1}
rocm-smi --showserial --showuniqueid > $o/device_stage$stage.txt 2>&1
prof() { tag=$1; shift; bash scripts/prof.sh r05f_$tag "$@" > $o/prof_$tag.log 2>&1; cp gpurun_out/prof_r05f_$tag/summary.txt $o/prof_${tag}_summary.txt 2>/dev/null; cp gpurun_out/prof_r05f_$tag/trace/*/*kernel_stats.csv $o/prof_${tag}_kernel_stats.csv 2>/dev/null; echo "prof $tag done"; }
if [ $stage = 1 ]; then
python bench.py > $o/bench_line.json 2> $o/bench_line.err
python bench.py --numerics fma --no-cpu-baseline > $o/bench_line_fma.json 2> $o/bench_line_fma.err
BENCH_OUT=$o/bench_workloads.txt bash scripts/bench_workloads.sh > /dev/null 2>&1
echo "bench done"; cut -c1-400 $o/bench_line.json
prof cfg2
prof cfg2fma --numerics fma
PICHA_HIP_LIB=picha_amd/variants/libpicha_hip_probe.so timeout -k 10 300 python scripts/clock_probe.py "cfg2 exact" "cfg2 fma" > $o/clock_stage1.txt 2>&1
cat $o/bench_workloads.txt; grep -v amdgpu.ids $o/clock_stage1.txt | cut -c1-220
fi
if [ $stage = 2 ]; then
prof cfg5 --workload cfg5_8k_rgba16_to_1080p_mitchel
prof cfg3 --workload cfg3_4k_rgb_to_grey_to_512_lanczos_chained
prof cfg3b --workload cfg3b_4k_grey_to_512_lanczos
prof cfg3a --workload cfg3a_4k_rgb_to_grey
PICHA_HIP_LIB=picha_amd/variants/libpicha_hip_probe.so timeout -k 10 300 python scripts/clock_probe.py cfg5 cfg3b > $o/clock_stage2.txt 2>&1
grep -v amdgpu.ids $o/clock_stage2.txt | cut -c1-220
fi
if [ $stage = 3 ]; then
for w in cfg2_4k_rgba_to_1080p_cubic_scale1 cfg1_640x480_rgba_to_320x240 lanczos_4k_rgba_to_1080p lanczos_4k_rgb_to_480x270 edge_jpeg_thumb_4k_rgb_to_512x288; do prof sec_$w --workload $w; done
PICHA_HIP_LIB=picha_amd/variants/libpicha_hip_probe.so timeout -k 10 300 python scripts/clock_probe.py lanczos jpeg > $o/clock_stage3.txt 2>&1
grep -v amdgpu.ids $o/clock_stage3.txt | cut -c1-220
fi
if [ $stage = 4 ]; then
PICHA_HIP_LIB=picha_amd/variants/libpicha_hip_probe.so timeout -k 10 600 python scripts/clock_probe.py > $o/clock.txt 2>&1
grep -v amdgpu.ids $o/clock.txt | cut -c1-220
PICHA_HIP_LIB=picha_amd/variants/libpicha_hip_probe.so timeout -k 10 400 python scripts/r05_timeline.py cfg2 cfg2fma cfg3b cfg5 > $o/timeline.txt 2>&1
grep -v amdgpu.ids $o/timeline.txt | cut -c1-300
fi
