#!/bin/bash
# r05 call 1: (a) VALU issue rate per SIMD by co-resident waves (start gate, HW_ID); (b) cfg2 on ONE source / ONE destination: shipped kernel, step-record
# prefetch, three ablations (no stores / no loop loads / no arithmetic), both flavours; (c) per-SIMD occupancy timeline of cfg2 / cfg2 fma / cfg3b / cfg5;
# (d) bench line of this box; (e) bench.py --gpus 2 launching its own ranks
export PYTHONPATH=. TMPDIR=/tmp
out=gpurun_out/r05; mkdir -p $out
bash scripts/unpack_variants.sh
rocm-smi --showuniqueid --showserial 2>/dev/null | grep -i -E "unique|serial" > $out/call01_device.txt
timeout -k 10 300 scripts/ubench/valu_issue > $out/call01_valu_issue.txt 2>&1 && tail -30 $out/call01_valu_issue.txt | cut -c1-260 &&
V=picha_amd/variants &&
timeout -k 10 400 python scripts/ab_shared.py --case cfg2 --rounds 6 --reps 10 base=. pf=$V/libpicha_hip_pf.so nostore=$V/libpicha_hip_abl1.so noload=$V/libpicha_hip_abl2.so noarith=$V/libpicha_hip_abl3.so > $out/call01_cfg2_ab_exact.txt 2>&1 &&
timeout -k 10 400 python scripts/ab_shared.py --case cfg2 --numerics fma --rounds 4 --reps 10 base=. pf=$V/libpicha_hip_pf.so nostore=$V/libpicha_hip_abl1.so noload=$V/libpicha_hip_abl2.so noarith=$V/libpicha_hip_abl3.so > $out/call01_cfg2_ab_fma.txt 2>&1 &&
cut -c1-330 $out/call01_cfg2_ab_exact.txt $out/call01_cfg2_ab_fma.txt &&
PICHA_HIP_LIB=$V/libpicha_hip_probe.so timeout -k 10 400 python scripts/r05_timeline.py cfg2 cfg2fma cfg3b cfg5 > $out/call01_timeline.txt 2>&1 ; cat $out/call01_timeline.txt | cut -c1-400 &&
timeout -k 10 300 python bench.py > $out/call01_bench_line.json 2> $out/call01_bench_err.txt && cut -c1-600 $out/call01_bench_line.json &&
timeout -k 10 600 python -m pytest tests/test_gpu_bench_contract.py -x -q -k "launches_its_own" 2>&1 | tail -5
