#!/bin/bash
# r05 call 51: rows per block of the edge kernel (edge_rows; LDS sized per block): more, shorter waves for a latency-bound pass; the whole GPU suite first
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05c; mkdir -p $o
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $o/gputest_edge_rows.log 2>&1; rc=$?; tail -4 $o/gputest_edge_rows.log | cut -c1-300
[ $rc = 0 ] || exit $rc
{
V="er64=.:edge_rows=64 er32=.:edge_rows=32 er24=.:edge_rows=24 er16=.:edge_rows=16 er8=.:edge_rows=8 er4=.:edge_rows=4 noedge=.:run=3"
python scripts/ab_shared.py --case cfg3b $V
python scripts/ab_shared.py --case cfg3b --numerics fma $V
python scripts/ab_shared.py --case 192:rgb:3840:2160:512:288 $V
python scripts/ab_shared.py --case 64:rgba:3840:2160:1920:1080:lanczos $V
python scripts/ab_shared.py --case 128:greya:3840:2160:512:288 $V
python scripts/ab_shared.py --case 64:rgb:3840:2160:480:270:lanczos $V
} > $o/ab_edge_rows.txt 2>&1
grep -v amdgpu.ids $o/ab_edge_rows.txt | grep -v "^# device" | cut -c1-200
