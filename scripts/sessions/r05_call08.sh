#!/bin/bash
# r05 call 8: persistent waves are all the same age — does a static priority per wave slot (s_setprio by HW_ID.wave_id & 3) break their lockstep?
export PYTHONPATH=. TMPDIR=/tmp
out=gpurun_out/r05; mkdir -p $out
bash scripts/unpack_variants.sh
V=picha_amd/variants
for num in exact fma; do
timeout -k 10 600 python scripts/ab_shared.py --case cfg2 --numerics $num --rounds 4 --reps 10 base=. pw=.:work=4 pw_prio=$V/libpicha_hip_prio.so:work=4 pw_prio_bw2=$V/libpicha_hip_prio.so:work=4,block_waves=2 pw_prio_bw1=$V/libpicha_hip_prio.so:work=4,block_waves=1 pw500_prio_bw2=$V/libpicha_hip_prio.so:work=4,block_waves=2,pw_grid=500 pw750_prio=$V/libpicha_hip_prio.so:work=4,pw_grid=750 > $out/call08_cfg2_pw_prio_$num.txt 2>&1
cut -c1-200 $out/call08_cfg2_pw_prio_$num.txt
done
