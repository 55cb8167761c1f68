#!/bin/bash
# r05 call 12: (a) the shipped cfg2 kernel held at 3 / 2 waves per SIMD by unused LDS (its registers allow 4); (b) s_setprio raised across the load-issue window;
# (c) the three cfg2 A/Bs of r04 re-taken on ONE source / ONE destination: outputs per lane, waves per workgroup, chunk lengths
export PYTHONPATH=. TMPDIR=/tmp
out=gpurun_out/r05; mkdir -p $out
bash scripts/unpack_variants.sh
V=picha_amd/variants
for num in exact fma; do
timeout -k 10 600 python scripts/ab_shared.py --case cfg2 --numerics $num --rounds 5 --reps 10 base=. cap4=$V/libpicha_hip_cap.so:PICHA_SYM_LDS=36864 cap3=$V/libpicha_hip_cap.so:PICHA_SYM_LDS=49152 cap2=$V/libpicha_hip_cap.so:PICHA_SYM_LDS=73728 cap3_bw2=$V/libpicha_hip_cap.so:PICHA_SYM_LDS=24576,block_waves=2 cap3_bw1=$V/libpicha_hip_cap.so:PICHA_SYM_LDS=12288,block_waves=1 prio_loads=$V/libpicha_hip_prl.so prio_loads_bw2=$V/libpicha_hip_prl.so:block_waves=2 > $out/call12_cfg2_cap_prio_$num.txt 2>&1
cut -c1-200 $out/call12_cfg2_cap_prio_$num.txt
timeout -k 10 600 python scripts/ab_shared.py --case cfg2 --numerics $num --rounds 5 --reps 10 base=. p2=.:per_lane=2 p3=.:per_lane=3 p4=.:per_lane=4 bw1=.:block_waves=1 bw2=.:block_waves=2 bw3=.:block_waves=3 c9=.:row_chunks=9 c12=.:row_chunks=12 c18=.:row_chunks=18 c27=.:row_chunks=27 c36=.:row_chunks=36 c54=.:row_chunks=54 bw2_c27=.:block_waves=2,row_chunks=27 bw2_t100=.:block_waves=2,work_tail=100 bw2_t300=.:block_waves=2,work_tail=300 grid=.:work=3 grid_bw2=.:work=3,block_waves=2 > $out/call12_cfg2_retake_$num.txt 2>&1
cut -c1-200 $out/call12_cfg2_retake_$num.txt
done
