#!/bin/bash
# r05 call 22: GPU suite; cfg3b defaults against the r03 form; cfg3 (fused rows kernel) with a floor under the workgroup's LDS (4 / 3 workgroups per CU at most) and chunk counts
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
bash scripts/unpack_variants.sh > /dev/null
python -m pytest tests -m gpu -x -q > $o/gputest_shu.log 2>&1 || { tail -40 $o/gputest_shu.log; exit 1; }
tail -2 $o/gputest_shu.log
{
for num in exact fma; do
python scripts/ab_shared.py --case cfg3b --numerics $num r03_form=.:run_edges=1 default=.
done
python scripts/ab_shared.py --case 256:grey:3840:2160:512:288 r03_form=.:run_edges=1 default=.
python scripts/ab_shared.py --case 128:greya:3840:2160:512:512:lanczos r03_form=.:run_edges=1 default=.
} > $o/ab_run_defaults.txt 2>&1
grep -v amdgpu.ids $o/ab_run_defaults.txt | cut -c1-250
{
for num in exact fma; do
python scripts/ab_shared.py --case cfg3 --numerics $num default=. lds40=picha_amd/variants/libpicha_hip_rl40.so lds54=picha_amd/variants/libpicha_hip_rl54.so lds54_c3=picha_amd/variants/libpicha_hip_rl54.so:row_chunks=3 \
  c8=.:row_chunks=8 lds40_c8=picha_amd/variants/libpicha_hip_rl40.so:row_chunks=8 c2=.:row_chunks=2 c16=.:row_chunks=16
done
} > $o/ab_rows_lds.txt 2>&1
grep -v amdgpu.ids $o/ab_rows_lds.txt | cut -c1-250
