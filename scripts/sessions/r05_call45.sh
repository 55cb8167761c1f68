#!/bin/bash
# r05 call 45: whole-round chunking for batches of small images on the streaming kernel: new default against the old chunk counts; other stream-kernel cells must not move; GPU suite
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
python -m pytest tests -m gpu -q > $o/gputest_stream_chunks.log 2>&1; tail -5 $o/gputest_stream_chunks.log | cut -c1-250
{
for num in exact fma; do
python scripts/ab_shared.py --case cfg1 --numerics $num --rounds 8 old=.:row_chunks=15 default=.
done
python scripts/ab_shared.py --case 64:rgba:640:480:320:240 old=.:row_chunks=15 default=.
python scripts/ab_shared.py --case 1024:rgba:640:480:320:240 old=.:row_chunks=15 default=.
python scripts/ab_shared.py --case 256:rgba:1280:720:640:360 old=.:row_chunks=22 default=. c4=.:row_chunks=4 c8=.:row_chunks=8
python scripts/ab_shared.py --case 256:rgb:640:480:320:240 old=.:row_chunks=15 default=.
python scripts/ab_shared.py --case 512:grey:640:480:320:240 old=.:row_chunks=15 default=.
python scripts/ab_shared.py --case 192:rgb:3840:2160:1920:1080 default=. c2=.:row_chunks=2
} > $o/ab_stream_round.txt 2>&1
grep -v amdgpu.ids $o/ab_stream_round.txt | grep -v "^# device" | cut -c1-200
