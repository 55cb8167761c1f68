#!/bin/bash
# r05 call 20/21: the clipped edge columns BESIDE the march (helper stream) against after it (stash / image), and the shared-unpack form of the grey 7.5x run kernel:
# cfg3b, cfg5, greya 7.5x, a ring-kernel upscale, lanczos 2x grey; GPU suite first
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/gputest_edges.log 2>&1 || { tail -30 $o/gputest_edges.log; exit 1; }
tail -2 $o/gputest_edges.log
{
for num in exact fma; do
python scripts/ab_shared.py --case cfg3b --numerics $num stash_after=.:run_edges=1 image_after=.:run_edges=2,run_shu=2 image_beside=.:run_edges=3,run_shu=2 shu_after=.:run_edges=2 shu_beside=.:run_edges=3
python scripts/ab_shared.py --case cfg5 --numerics $num image_after=.:run_edges=2 image_beside=.:run_edges=3
done
python scripts/ab_shared.py --case 256:grey:3840:2160:512:288 stash_after=.:run_edges=1 image_beside=.:run_edges=3,run_shu=2 shu_beside=.:run_edges=3
python scripts/ab_shared.py --case 128:greya:3840:2160:512:512:lanczos stash_after=.:run_edges=1 image_after=.:run_edges=2 image_beside=.:run_edges=3
python scripts/ab_shared.py --case 64:rgba:960:540:1920:1080 stash_after=.:run_edges=1 image_after=.:run_edges=2 image_beside=.:run_edges=3
python scripts/ab_shared.py --case 64:grey:3840:2160:1920:1080:lanczos stash_after=.:run_edges=1 image_after=.:run_edges=2 image_beside=.:run_edges=3
} > $o/ab_run_edges.txt 2>&1
grep -v amdgpu.ids $o/ab_run_edges.txt | cut -c1-250
