#!/bin/bash
# r04 call 17: in-kernel clocks of every kernel named in VERDICT r03 (probe build) + trace and PMC passes of the secondary cells
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r04/sec; mkdir -p $o
PICHA_HIP_LIB=picha_amd/variants/libpicha_hip_probe.so timeout -k 10 500 python scripts/r04_clock_probe.py > $o/clock.txt 2>&1; echo "clock rc=$?"
grep -v amdgpu.ids $o/clock.txt | cut -c1-250
for w in lanczos_4k_rgb_to_480x270 lanczos_4k_rgba_to_1080p edge_jpeg_thumb_4k_rgb_to_512x288 cfg1_640x480_rgba_to_320x240 cfg2_4k_rgba_to_1080p_cubic_scale1; do
  timeout -k 10 500 bash scripts/prof.sh r04s_$w --workload $w > $o/prof_$w.log 2>&1
  cp gpurun_out/prof_r04s_$w/summary.txt $o/prof_${w}_summary.txt 2>/dev/null
  echo "== $w"; grep "launches " $o/prof_${w}_summary.txt | cut -c1-200
done
