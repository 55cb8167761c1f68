#!/bin/bash
# r05 call 24: rows kernel with WAVE-PRIVATE strips (64 threads per strip, no barrier) against workgroup strips: cfg3 (fused), plain grey / rgba / greya downscales that take the rows kernel; parity of the rows tests first
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
python -m pytest tests -m gpu -x -q -k "rows or chain or fused or convert_resize or fuzz" > $o/gputest_rows_wave.log 2>&1 || { tail -40 $o/gputest_rows_wave.log; exit 1; }
tail -2 $o/gputest_rows_wave.log
{
for num in exact fma; do
python scripts/ab_shared.py --case cfg3 --numerics $num wg=.:rows_wave=2 wave=. wave_c2=.:row_chunks=2 wave_c3=.:row_chunks=3 wave_c6=.:row_chunks=6 wave_c8=.:row_chunks=8 wave_c12=.:row_chunks=12
done
python scripts/ab_shared.py --case 256:grey:3840:2160:512:512:lanczos wg=.:rows=1,run=2,rows_wave=2 wave=.:rows=1,run=2 default=.
python scripts/ab_shared.py --case 128:greya:3840:2160:512:288 wg=.:rows=1,run=2,rows_wave=2 wave=.:rows=1,run=2 default=.
} > $o/ab_rows_wave.txt 2>&1
grep -v amdgpu.ids $o/ab_rows_wave.txt | cut -c1-250
