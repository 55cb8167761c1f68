#!/bin/bash
# r05 call 28: GPU suite; the shared-unpack run kernel on rgb thumbnails (7.5x / 6x / 8x with picha's defaults) against the accumulator kernel
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/gputest_shu4.log 2>&1 || { tail -40 $o/gputest_shu4.log; exit 1; }
tail -2 $o/gputest_shu4.log
{
for num in exact fma; do
python scripts/ab_shared.py --case jpegthumb --numerics $num acc=.:run=2 shu=.:run=1 shu_bw2=.:run=1,block_waves=2 shu_bw1=.:run=1,block_waves=1 shu_c4=.:run=1,row_chunks=4 shu_c2=.:run=1,row_chunks=2 shu_c8=.:run=1,row_chunks=8
python scripts/ab_shared.py --case 128:rgb:3840:2160:640:360 --numerics $num acc=.:run=2 shu=.:run=1 shu_bw2=.:run=1,block_waves=2
python scripts/ab_shared.py --case 128:rgb:3840:2160:480:270 --numerics $num acc=.:run=2 shu=.:run=1 shu_bw2=.:run=1,block_waves=2
done
python scripts/ab_shared.py --case 32:rgb:3840:2160:512:288 acc=.:run=2 shu=.:run=1 shu_bw2=.:run=1,block_waves=2
python scripts/ab_shared.py --case 8:rgb:3840:2160:512:288 acc=.:run=2 shu=.:run=1 shu_bw2=.:run=1,block_waves=2
} > $o/ab_shu_rgb.txt 2>&1
grep -v amdgpu.ids $o/ab_shu_rgb.txt | cut -c1-230
