#!/bin/bash
# r05 call 26: GPU suite with the shared-unpack shapes at both phases scalar + two row buffers + whole-round chunks; cfg3b against the r03 form and the first SHU build
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/gputest_shu2.log 2>&1 || { tail -40 $o/gputest_shu2.log; exit 1; }
tail -2 $o/gputest_shu2.log
{
for num in exact fma; do
python scripts/ab_shared.py --case cfg3b --numerics $num r03_form=.:run_edges=1 default=. shu=.:run_shu=1 shu_c8=.:run_shu=1,row_chunks=8 shu_c10=.:run_shu=1,row_chunks=10 shu_c12=.:run_shu=1,row_chunks=12
done
python scripts/ab_shared.py --case 256:grey:3840:2160:512:288 r03_form=.:run_edges=1 default=. shu=.:run_shu=1
python scripts/ab_shared.py --case 64:grey:3840:2160:512:512:lanczos r03_form=.:run_edges=1 default=. shu_c8=.:run_shu=1,row_chunks=8
} > $o/ab_shu_final.txt 2>&1
grep -v amdgpu.ids $o/ab_shu_final.txt | cut -c1-230
