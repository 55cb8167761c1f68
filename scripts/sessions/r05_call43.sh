#!/bin/bash
# r05 call 43: grey / greya lanczos thumbnails at 8x / 6x on the shared-unpack run kernel (row list) against the accumulator kernel
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
{
for px in grey greya; do
for num in exact fma; do
python scripts/ab_shared.py --case 128:$px:3840:2160:480:270:lanczos --numerics $num acc=.:run=2 shu=.:run=1 default=.
python scripts/ab_shared.py --case 128:$px:3840:2160:640:360:lanczos --numerics $num acc=.:run=2 shu=.:run=1 default=.
done; done
} > $o/ab_shu_list_grey.txt 2>&1
grep -v amdgpu.ids $o/ab_shu_list_grey.txt | grep -v "^# device" | cut -c1-200
