#!/bin/bash
# r05 call 41: shared unpack over TWO neighbours (lanczos thumbnails of rgb, greya 7.5x) against the accumulator / plain run kernels; parity of the run-kernel tests first
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
python -m pytest tests -m gpu -x -q -k "run_kernel or shared_unpack or half_integer or fuzz or edge_columns" > $o/gputest_shu2n.log 2>&1; tail -4 $o/gputest_shu2n.log | cut -c1-250
{
for num in exact fma; do
python scripts/ab_shared.py --case lanczos8x --numerics $num acc=.:run=2 shu2=.:run=1 shu2_bw2=.:run=1,block_waves=2 shu2_c4=.:run=1,row_chunks=4 shu2_c8=.:run=1,row_chunks=8
python scripts/ab_shared.py --case 64:rgb:3840:2160:640:360:lanczos --numerics $num acc=.:run=2 shu2=.:run=1 shu2_bw2=.:run=1,block_waves=2
python scripts/ab_shared.py --case 128:rgb:3840:2160:512:288:lanczos --numerics $num acc=.:run=2 shu2=.:run=1
python scripts/ab_shared.py --case 128:greya:3840:2160:512:512:lanczos --numerics $num plain=.:run_shu=2 shu2=.:run_shu=1
done
python scripts/ab_shared.py --case 128:greya:3840:2160:512:288 plain=.:run=1,run_shu=2 shu=.:run=1,run_shu=1 default=.
} > $o/ab_shu_two_neighbours.txt 2>&1
grep -v amdgpu.ids $o/ab_shu_two_neighbours.txt | grep -v "^# device" | cut -c1-200
