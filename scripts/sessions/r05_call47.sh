#!/bin/bash
# r05 call 47: grey / greya thumbnails at 8x / 6x with picha's default filter (23 / 17 taps): shared-unpack form against the plain run kernel / the accumulator kernel
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
{
for px in grey greya; do
python scripts/ab_shared.py --case 128:$px:3840:2160:480:270 old=.:run_shu=2 shu=.:run=1,run_shu=1 default=.
python scripts/ab_shared.py --case 128:$px:3840:2160:640:360 old=.:run_shu=2 shu=.:run=1,run_shu=1 default=.
done
python scripts/ab_shared.py --case 128:grey:3840:2160:480:270 --numerics fma old=.:run_shu=2 shu=.:run=1,run_shu=1 default=.
} > $o/ab_shu_default_filter_int.txt 2>&1
grep -v amdgpu.ids $o/ab_shu_default_filter_int.txt | grep -v "^# device" | cut -c1-200
