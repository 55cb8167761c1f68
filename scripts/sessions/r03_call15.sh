#!/bin/bash
o=gpurun_out/r03/call15; mkdir -p $o
scripts/ubench/cvt_rtz > $o/cvt_rtz.txt 2>&1
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_chain_and_helpers.py -x -q > $o/pytest.txt 2>&1 || { tail -30 $o/pytest.txt; exit 1; }
tail -2 $o/pytest.txt
bash scripts/ab_many_libs.sh "picha_amd/libpicha_hip_med3.so picha_amd/libpicha_hip.so" > $o/ab_pack_cfg2.txt 2>&1
bash scripts/ab_many_libs.sh "picha_amd/libpicha_hip_med3.so picha_amd/libpicha_hip.so" --workload cfg2_4k_rgba_to_1080p_cubic_scale1 > $o/ab_pack_cfg2_7tap.txt 2>&1
bash scripts/ab_many_libs.sh "picha_amd/libpicha_hip_med3.so picha_amd/libpicha_hip.so" --workload cfg3_4k_rgb_to_grey_to_512_lanczos_chained > $o/ab_pack_cfg3.txt 2>&1
bash scripts/ab_many_libs.sh "picha_amd/libpicha_hip_med3.so picha_amd/libpicha_hip.so" --workload edge_jpeg_4k_rgb_to_1080p > $o/ab_pack_rgb.txt 2>&1
bash scripts/ab_many_libs.sh "picha_amd/libpicha_hip_med3.so picha_amd/libpicha_hip.so" --workload cfg3a_4k_rgb_to_grey > $o/ab_pack_cfg3a.txt 2>&1
cat $o/ab_pack_*.txt
