#!/bin/bash
# r05 call 29: GPU suite; planner defaults for the rgb shared-unpack shapes against the accumulator kernel; bench workloads
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/gputest_shu5.log 2>&1 || { tail -40 $o/gputest_shu5.log; exit 1; }
tail -2 $o/gputest_shu5.log
{
for num in exact fma; do
python scripts/ab_shared.py --case jpegthumb --numerics $num acc=.:run=2 default=.
python scripts/ab_shared.py --case 128:rgb:3840:2160:640:360 --numerics $num acc=.:run=2 default=.
python scripts/ab_shared.py --case 128:rgb:3840:2160:480:270 --numerics $num acc=.:run=2 default=.
done
python scripts/ab_shared.py --case 32:rgb:3840:2160:512:288 acc=.:run=2 default=.
python scripts/ab_shared.py --case 16:rgb:3840:2160:512:288 acc=.:run=2 default=. shu=.:run=1
python scripts/ab_shared.py --case 64:rgb:1920:1080:256:144 acc=.:run=2 default=. shu=.:run=1
python scripts/ab_shared.py --case 64:rgb:7680:4320:1024:576 acc=.:run=2 default=. shu=.:run=1
} > $o/ab_shu_rgb_defaults.txt 2>&1
grep -v amdgpu.ids $o/ab_shu_rgb_defaults.txt | cut -c1-230
BENCH_OUT=$o/bench_workloads.txt bash scripts/bench_workloads.sh > /dev/null 2>&1
cat $o/bench_workloads.txt | cut -c1-230
