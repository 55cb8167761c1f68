#!/bin/bash
# r05 call 14: persistent waves — is the time lost inside the marches or between them?
export PYTHONPATH=. TMPDIR=/tmp
out=gpurun_out/r05; mkdir -p $out
bash scripts/unpack_variants.sh
PICHA_HIP_LIB=picha_amd/variants/libpicha_hip_probe.so timeout -k 10 400 python scripts/r05_timeline.py cfg2 cfg2:work=4 cfg2:work=4,block_waves=2,pw_grid=500 cfg2:work=4,row_chunks=9 cfg2fma cfg2fma:work=4 > $out/call14_timeline_pw.txt 2>&1 ; cut -c1-420 $out/call14_timeline_pw.txt
