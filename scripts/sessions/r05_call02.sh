#!/bin/bash
# r05 call 2: persistent waves ("work" = 4) on cfg2 — parity tests, then ONE source / ONE destination A/B against the item list over waves per workgroup, chunk
# lengths and the short tail, both flavours; the loads+stores-only ablation (fixed: every row's loads live); occupancy timelines of both forms
export PYTHONPATH=. TMPDIR=/tmp
out=gpurun_out/r05; mkdir -p $out
bash scripts/unpack_variants.sh
V=picha_amd/variants
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "persistent_waves or work_item_list or persistent_launch_covers" 2>&1 | tail -5 &&
timeout -k 10 400 python scripts/ab_shared.py --case cfg2 --rounds 6 --reps 10 base=. bw1=.:block_waves=1 bw2=.:block_waves=2 pw=.:work=4 pw_bw1=.:work=4,block_waves=1 pw_bw2=.:work=4,block_waves=2 pw_c27=.:work=4,row_chunks=27 pw_c36=.:work=4,row_chunks=36 pw_c12=.:work=4,row_chunks=12 pw_tail0=.:work=4,work_tail=0 pw_tail400=.:work=4,work_tail=400 noarith=$V/libpicha_hip_abl3.so > $out/call02_cfg2_pw_exact.txt 2>&1 &&
timeout -k 10 400 python scripts/ab_shared.py --case cfg2 --numerics fma --rounds 6 --reps 10 base=. bw1=.:block_waves=1 pw=.:work=4 pw_bw1=.:work=4,block_waves=1 pw_c27=.:work=4,row_chunks=27 pw_c36=.:work=4,row_chunks=36 pw_c18=.:work=4,row_chunks=18 pw_tail0=.:work=4,work_tail=0 noarith=$V/libpicha_hip_abl3.so > $out/call02_cfg2_pw_fma.txt 2>&1 ;
cut -c1-250 $out/call02_cfg2_pw_exact.txt $out/call02_cfg2_pw_fma.txt &&
PICHA_HIP_LIB=$V/libpicha_hip_probe.so timeout -k 10 400 python scripts/r05_timeline.py cfg2 cfg2:work=4 cfg2:block_waves=1 cfg2fma cfg2fma:work=4 > $out/call02_timeline.txt 2>&1 ; cut -c1-420 $out/call02_timeline.txt
