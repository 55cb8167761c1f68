#!/bin/bash
# usage: final_r05b.sh <stage 1|2|3>   — round-5 evidence on the FINAL library of the round (second session: shared-unpack run kernel, rgb thumbnails on it,
# upscales with the clipped columns inside): bench line (+ cpu baseline), every workload, rocprofv3 trace + PMC passes for the BASELINE stages and the cells
# that changed.  Outputs under gpurun_out/r05/final_b/ (copied to profiles/ by scripts/sessions/r05_collect_final_b.sh).  The in-kernel clocks and timelines of
# the first session (profiles/r05_inkernel_clock.txt, r05_timeline.txt) are not re-taken: the kernels they describe (cfg2, cfg5) did not change.
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05/final_b; mkdir -p $o
stage=${1:-1}
rocm-smi --showserial --showuniqueid > $o/device_stage$stage.txt 2>&1
prof() { tag=$1; shift; bash scripts/prof.sh r05g_$tag "$@" > $o/prof_$tag.log 2>&1; cp gpurun_out/prof_r05g_$tag/summary.txt $o/prof_${tag}_summary.txt 2>/dev/null; cp gpurun_out/prof_r05g_$tag/trace/*/*kernel_stats.csv $o/prof_${tag}_kernel_stats.csv 2>/dev/null; echo "prof $tag done"; }
if [ $stage = 1 ]; then
python bench.py > $o/bench_line.json 2> $o/bench_line.err
python bench.py --numerics fma --no-cpu-baseline > $o/bench_line_fma.json 2> $o/bench_line_fma.err
BENCH_OUT=$o/bench_workloads.txt bash scripts/bench_workloads.sh > /dev/null 2>&1
echo "bench done"; cut -c1-400 $o/bench_line.json; cat $o/bench_workloads.txt | cut -c1-220
prof cfg2
fi
if [ $stage = 2 ]; then
prof cfg3b --workload cfg3b_4k_grey_to_512_lanczos
prof cfg3 --workload cfg3_4k_rgb_to_grey_to_512_lanczos_chained
prof cfg5 --workload cfg5_8k_rgba16_to_1080p_mitchel
prof cfg2fma --numerics fma
fi
if [ $stage = 3 ]; then
for w in edge_jpeg_thumb_4k_rgb_to_512x288 up_960x540_rgba_to_1080p up_960x540_rgba16_to_1080p lanczos_4k_rgba_to_1080p; do PROF_TRACE_ONLY= prof sec_$w --workload $w; done
fi
