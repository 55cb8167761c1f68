#!/bin/bash
# copies the outputs of scripts/final_r05.sh (gpurun_out/r05/final/) into profiles/ and derives the traffic / valu json files and the utilisation table
o=gpurun_out/r05/final; p=profiles
cp $o/bench_line.json $p/r05_bench_line.json; cp $o/bench_line_fma.json $p/r05_bench_line_fma.json
{ echo "# scripts/bench_workloads.sh on the box of final_r05.sh stage 1 (serial $(grep -i serial $o/device_stage1.txt | awk '{print $NF}')): 20 timed launches after 3 warm-ups, bit-exact flavour | fma"; cat $o/bench_workloads.txt; } > $p/r05_bench_workloads.txt
for t in cfg2 cfg2fma cfg3 cfg3a cfg3b cfg5; do cp $o/prof_${t}_summary.txt $p/r05_${t}_rocprofv3_summary.txt; done
cp $o/prof_cfg2_kernel_stats.csv $p/r05_cfg2_kernel_stats.csv; cp $o/prof_cfg3b_kernel_stats.csv $p/r05_cfg3b_kernel_stats.csv; cp $o/prof_cfg5_kernel_stats.csv $p/r05_cfg5_kernel_stats.csv
for f in $o/prof_sec_*_summary.txt; do b=$(basename $f _summary.txt); cp $f $p/r05_${b#prof_}_rocprofv3_summary.txt; done
{ echo "# In-kernel shader clock of the shipped kernels under their own load (scripts/clock_probe.py on a -DPICHA_CLOCK_PROBE build; method: profiles/r04_inkernel_clock.txt), taken in the"
  echo "# same gpurun call as the rocprofv3 passes of scripts/final_r05.sh they are used with (stage 1: serial $(grep -i serial $o/device_stage1.txt | awk '{print $NF}'); stages 2-4: $(grep -i serial $o/device_stage2.txt | awk '{print $NF}'))"
  for s in 1 2 3; do echo "## stage $s"; grep -v "amdgpu.ids\|^library" $o/clock_stage$s.txt; done
  echo "## stage 4 (all cases)"; grep -v "amdgpu.ids\|^library" $o/clock.txt; } > $p/r05_inkernel_clock.txt
python scripts/make_traffic.py $o/prof_cfg2_summary.txt $o/bench_line.json 5 > $p/traffic_cfg2_4k_rgba_to_1080p_cubic.json
python scripts/make_traffic.py $o/prof_cfg2fma_summary.txt $o/bench_line_fma.json 5 > $p/traffic_cfg2_4k_rgba_to_1080p_cubic.fma.json
g() { grep "^$1" $o/$2 | head -1 | sed 's/.*median \([0-9.]*\) GHz.*/\1/'; }
python scripts/make_valu.py $o/prof_cfg2_summary.txt $o/bench_line.json $(g "cfg2 exact" clock_stage1.txt) $(g "cfg2 fma" clock_stage1.txt) r05
{ echo "# VALU utilisation of the shipped kernels (r05): dynamic VALU wave-instructions per launch (rocprofv3 PMC) x 2.05 cycles / (1024 SIMDs x kernel cycles at the IN-KERNEL clock of the"
  echo "# same gpurun call); scripts/issue_util.py, rate from profiles/r05_ubench_valu_issue.txt.  Replaces profiles/r04_issue_models.txt (priced at 2.45 cycles, several lines above 100 %)."
  echo "# resident waves / wait shares from the same passes; fabric traffic = FETCH_SIZE x 2 + WRITE_SIZE."
  python scripts/issue_util.py $o/prof_cfg2_summary.txt "resize_sym_kernel<4, false, 3, 2, 5, 2, 5, false" $(g "cfg2 exact" clock_stage1.txt) "cfg2 bit-exact (256 x 4K rgba -> 1080p)"
  python scripts/issue_util.py $o/prof_cfg2fma_summary.txt "resize_sym_kernel<4, false, 3, 2, 5, 2, 5, true" $(g "cfg2 fma" clock_stage1.txt) "cfg2 fma"
  python scripts/issue_util.py $o/prof_cfg5_summary.txt "resize_run_kernel<4, true" $(g "cfg5 exact" clock_stage2.txt) "cfg5 bit-exact (32 x 8K r16g16b16a16 -> 1080p mitchel; SDWA unpack + DPP exchange)" 0 1
  python scripts/issue_util.py $o/prof_cfg3b_summary.txt "resize_run_kernel<1, false" $(g "cfg3b exact" clock_stage2.txt) "cfg3b bit-exact (256 x 4K grey -> 512x512 lanczos)"
  python scripts/issue_util.py $o/prof_cfg3_summary.txt "resize_rows_kernel<3, false" $(g "cfg3b exact" clock_stage2.txt) "cfg3 fused rgb -> grey -> 512x512 lanczos, bit-exact (clock: cfg3b's, same call; the rows kernel has no probe)"
  python scripts/issue_util.py $o/prof_sec_lanczos_4k_rgb_to_480x270_summary.txt "resize_acc_kernel<3, false" $(g "lanczos 8x" clock_stage3.txt) "lanczos 8x rgb thumbnails (64 x 4K rgb -> 480x270)"
  python scripts/issue_util.py $o/prof_sec_lanczos_4k_rgba_to_1080p_summary.txt "resize_ring_kernel<4, false" $(g "lanczos 2x" clock_stage3.txt) "lanczos 2x rgba (64 x 4K -> 1080p, ring kernel)"
  python scripts/issue_util.py $o/prof_sec_edge_jpeg_thumb_4k_rgb_to_512x288_summary.txt "resize_acc_kernel<3, false" $(g "jpeg thumbnail" clock_stage3.txt) "jpeg thumbnail (192 x 4K rgb -> 512x288)"
  python scripts/issue_util.py $o/prof_sec_cfg2_4k_rgba_to_1080p_cubic_scale1_summary.txt "resize_stream_kernel<4, false" 1.85 "cfg2 with {filter:'cubic'} (7 taps), bit-exact (clock: assumed 1.85 GHz = cfg2's; the stream kernel's probe build is not representative, profiles/r04_inkernel_clock.txt)"
  python scripts/issue_util.py $o/prof_sec_cfg1_640x480_rgba_to_320x240_summary.txt "resize_stream_kernel<4, false" 1.85 "cfg1-size batch (256 x 640x480 rgba -> 320x240), bit-exact (clock assumed 1.85 GHz)"
} > $p/r05_issue_util.txt 2>&1
{ echo; echo "## final library (scripts/final_r05.sh stage 4): cfg2 bit-exact now launches 2-wave workgroups — 3.51 of 4 waves resident, turnover 7 % of the slot time (was 2.88 / 25 %), for -1.3 % of time"; grep -v "amdgpu.ids\|^library" $o/timeline.txt | cut -c1-330; } >> $p/r05_timeline.txt
cat $p/r05_issue_util.txt | cut -c1-220
