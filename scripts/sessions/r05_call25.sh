#!/bin/bash
# r05 call 25: does cfg3b's shared-unpack kernel gain from more waves per SIMD?  both phases' weights in SGPRs + two row buffers (87 VGPRs: 5 waves), the same under a
# 6- / 8-wave register budget (80 / 64 VGPRs, a few spilled dwords), each at 8 / 12 / 16 row chunks (a launch of 4 / 6 / 8 waves per SIMD)
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
bash scripts/unpack_variants.sh > /dev/null
V=picha_amd/variants
{
for num in exact fma; do
python scripts/ab_shared.py --case cfg3b --numerics $num --rounds 5 default=.:run_shu=1 c12=.:run_shu=1,row_chunks=12 \
  ws2nb2=$V/libpicha_hip_ws2nb2.so:run_shu=1 ws2nb2_c10=$V/libpicha_hip_ws2nb2.so:run_shu=1,row_chunks=10 ws2nb2_c12=$V/libpicha_hip_ws2nb2.so:run_shu=1,row_chunks=12 \
  w6=$V/libpicha_hip_ws2nb2w6.so:run_shu=1 w6_c12=$V/libpicha_hip_ws2nb2w6.so:run_shu=1,row_chunks=12 w6_c16=$V/libpicha_hip_ws2nb2w6.so:run_shu=1,row_chunks=16 \
  w8=$V/libpicha_hip_ws2nb2w8.so:run_shu=1 w8_c12=$V/libpicha_hip_ws2nb2w8.so:run_shu=1,row_chunks=12 w8_c16=$V/libpicha_hip_ws2nb2w8.so:run_shu=1,row_chunks=16
done
} > $o/ab_shu_occupancy.txt 2>&1
grep -v amdgpu.ids $o/ab_shu_occupancy.txt | cut -c1-250
