#!/bin/bash
# r05 call 31 (timing only, WRONG results): the fused rows kernel with wave-private strips, ONE LDS buffer and wave-uniform weight addresses (65 VGPRs instead of 127):
# what would 7 waves per SIMD buy cfg3?
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
bash scripts/unpack_variants.sh > /dev/null
V=picha_amd/variants
{
for num in exact fma; do
python scripts/ab_shared.py --case cfg3 --numerics $num --no-oracle rows=.:run=2 u1=$V/libpicha_hip_rowsu1.so:run=2 u4=$V/libpicha_hip_rowsu4.so:run=2 u1_c6=$V/libpicha_hip_rowsu1.so:run=2,row_chunks=6 u1_c7=$V/libpicha_hip_rowsu1.so:run=2,row_chunks=7 u1_c8=$V/libpicha_hip_rowsu1.so:run=2,row_chunks=8 u4_c8=$V/libpicha_hip_rowsu4.so:run=2,row_chunks=8 u1_c3=$V/libpicha_hip_rowsu1.so:run=2,row_chunks=3
done
} > $o/ab_rows_uniform_probe.txt 2>&1
grep -v amdgpu.ids $o/ab_rows_uniform_probe.txt | cut -c1-230
