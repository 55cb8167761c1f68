#!/bin/bash
# r05 call 3: persistent waves at reduced occupancy ("pw_grid" = per mille of the resident grid): is the full chip of waves too many streams for the memory system?
export PYTHONPATH=. TMPDIR=/tmp
out=gpurun_out/r05; mkdir -p $out
bash scripts/unpack_variants.sh
V=picha_amd/variants
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "persistent_waves" 2>&1 | tail -5 &&
timeout -k 10 400 python scripts/ab_shared.py --case cfg2 --rounds 4 --reps 10 base=. pw1000=.:work=4 pw875=.:work=4,pw_grid=875 pw750=.:work=4,pw_grid=750 pw625=.:work=4,pw_grid=625 pw500=.:work=4,pw_grid=500 pw375=.:work=4,pw_grid=375 pw250=.:work=4,pw_grid=250 pw500_bw1=.:work=4,pw_grid=500,block_waves=1 pw750_bw1=.:work=4,pw_grid=750,block_waves=1 pw500_bw2=.:work=4,pw_grid=500,block_waves=2 pw500_c9=.:work=4,pw_grid=500,row_chunks=9 pw500_c36=.:work=4,pw_grid=500,row_chunks=36 > $out/call03_cfg2_pwgrid_exact.txt 2>&1 &&
timeout -k 10 400 python scripts/ab_shared.py --case cfg2 --numerics fma --rounds 4 --reps 10 base=. pw1000=.:work=4 pw750=.:work=4,pw_grid=750 pw625=.:work=4,pw_grid=625 pw500=.:work=4,pw_grid=500 pw375=.:work=4,pw_grid=375 pw250=.:work=4,pw_grid=250 pw500_c18=.:work=4,pw_grid=500,row_chunks=18 > $out/call03_cfg2_pwgrid_fma.txt 2>&1 ;
cut -c1-215 $out/call03_cfg2_pwgrid_exact.txt $out/call03_cfg2_pwgrid_fma.txt &&
PICHA_HIP_LIB=$V/libpicha_hip_probe.so timeout -k 10 400 python scripts/r05_timeline.py cfg2:work=4 cfg2:work=4,pw_grid=500 cfg2:block_waves=1 cfg2fma > $out/call03_timeline.txt 2>&1 ; cut -c1-420 $out/call03_timeline.txt
