#!/bin/bash
# r05 call 46: rgba thumbnails on the shared-unpack run kernel (dword pixels: 60 own floats per lane, 130-150 VGPRs) against the accumulator kernel
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
{
for num in exact fma; do
python scripts/ab_shared.py --case 192:rgba:3840:2160:512:288 --numerics $num acc=.:run=2 shu=.:run=1 shu_bw2=.:run=1,block_waves=2 shu_c4=.:run=1,row_chunks=4 shu_c8=.:run=1,row_chunks=8
python scripts/ab_shared.py --case 128:rgba:3840:2160:512:288:lanczos --numerics $num acc=.:run=2 shu=.:run=1
python scripts/ab_shared.py --case 128:rgba:3840:2160:480:270 --numerics $num acc=.:run=2 shu=.:run=1
done
} > $o/ab_shu_rgba.txt 2>&1
grep -v amdgpu.ids $o/ab_shu_rgba.txt | grep -v "^# device" | cut -c1-200
python -m pytest tests -m gpu -x -q -k "fuzz or half_integer or shared_unpack" > $o/gputest_rgba_shu.log 2>&1; tail -3 $o/gputest_rgba_shu.log
