#!/bin/bash
# r05 call 15: persistent waves without the in-flight atomic: static pair assignment (no atomics), and the atomic taken only after the march
export PYTHONPATH=. TMPDIR=/tmp
out=gpurun_out/r05; mkdir -p $out
bash scripts/unpack_variants.sh
V=picha_amd/variants
for num in exact fma; do
timeout -k 10 600 python scripts/ab_shared.py --case cfg2 --numerics $num --rounds 4 --reps 10 base=. bw2=.:block_waves=2 pw=.:work=4 pw_static=$V/libpicha_hip_pwstat.so:work=4 pw_static_bw2=$V/libpicha_hip_pwstat.so:work=4,block_waves=2 pw_static_t0=$V/libpicha_hip_pwstat.so:work=4,work_tail=0 pw_late=$V/libpicha_hip_pwlate.so:work=4 pw_late_bw2=$V/libpicha_hip_pwlate.so:work=4,block_waves=2 pw_static_750=$V/libpicha_hip_pwstat.so:work=4,pw_grid=750 > $out/call15_cfg2_pw_noatomic_$num.txt 2>&1
cut -c1-200 $out/call15_cfg2_pw_noatomic_$num.txt
done
PICHA_HIP_LIB=$V/libpicha_hip_probestat.so timeout -k 10 400 python scripts/r05_timeline.py cfg2:work=4 > $out/call15_timeline_static.txt 2>&1 ; cut -c1-420 $out/call15_timeline_static.txt
