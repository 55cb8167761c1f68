#!/bin/bash
# r05 call 27: GPU suite (SHU default for every exact grey 7.5x shape); register budgets: lanczos 2x rgba on the ring kernel at 128 VGPRs (4 waves per SIMD instead of 3),
# the 7-tap rotating streaming kernel at 80 (6 instead of 5)
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
bash scripts/unpack_variants.sh > /dev/null
V=picha_amd/variants
python -m pytest tests -m gpu -x -q > $o/gputest_shu3.log 2>&1 || { tail -40 $o/gputest_shu3.log; exit 1; }
tail -2 $o/gputest_shu3.log
{
for num in exact fma; do
python scripts/ab_shared.py --case lanczos2x --numerics $num default=. ring4=$V/libpicha_hip_ring4.so ring4_c=$V/libpicha_hip_ring4.so:row_chunks=8 default_c=.:row_chunks=8
python scripts/ab_shared.py --case cfg2_7tap --numerics $num default=. rot6=$V/libpicha_hip_strot6.so
done
python scripts/ab_shared.py --case 64:rgb:3840:2160:1920:1080:lanczos default=. ring4=$V/libpicha_hip_ring4.so
} > $o/ab_reg_budgets.txt 2>&1
grep -v amdgpu.ids $o/ab_reg_budgets.txt | cut -c1-230
