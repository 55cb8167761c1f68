#!/bin/bash
# r05 call 58: the cells at the bottom of the final cliff survey (grey / greya strong downscales with 48-80 tap windows on resize_acc_long, 0.11-0.17 of the roof at 64 images):
# workgroup width and chunk count of the long-window accumulator kernel against the planner's choice, at 64 and 256 images
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05c; mkdir -p $o
{
V="default=. lb64=.:long_block=64 lb128=.:long_block=128 lb256=.:long_block=256 lb256_c4=.:long_block=256,row_chunks=4 lb256_c8=.:long_block=256,row_chunks=8 lb64_c8=.:long_block=64,row_chunks=8 lb64_c16=.:long_block=64,row_chunks=16"
python scripts/ab_shared.py --case 64:grey:3840:2160:384:216:cubic $V
python scripts/ab_shared.py --case 256:grey:3840:2160:384:216:cubic $V
python scripts/ab_shared.py --case 64:grey:3840:2160:320:180:lanczos $V
python scripts/ab_shared.py --case 64:greya:3840:2160:333:187:mitchel $V
python scripts/ab_shared.py --case 64:grey:3840:2160:384:216:cubic --numerics fma $V
} > $o/ab_acc_long_cells.txt 2>&1
grep -v amdgpu.ids $o/ab_acc_long_cells.txt | grep -v "^# device" | cut -c1-210
