#!/bin/bash
# r05 call 42: the shared-unpack run kernel following a row LIST (ring-aliased vertical tables: lanczos at 6x / 8x) against the accumulator kernel; GPU suite
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
python -m pytest tests -m gpu -q > $o/gputest_list.log 2>&1; tail -12 $o/gputest_list.log | cut -c1-250
{
for num in exact fma; do
python scripts/ab_shared.py --case lanczos8x --numerics $num acc=.:run=2 shu2=.:run=1 shu2_bw2=.:run=1,block_waves=2 shu2_c4=.:run=1,row_chunks=4 shu2_c8=.:run=1,row_chunks=8 default=.
python scripts/ab_shared.py --case 64:rgb:3840:2160:640:360:lanczos --numerics $num acc=.:run=2 shu2=.:run=1 shu2_bw2=.:run=1,block_waves=2 default=.
done
python scripts/ab_shared.py --case lanczos2x acc=.:run=2 default=.
} > $o/ab_shu_row_list.txt 2>&1
grep -v amdgpu.ids $o/ab_shu_row_list.txt | grep -v "^# device" | cut -c1-200
