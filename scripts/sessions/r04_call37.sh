#!/bin/bash
# r04 call 37: cfg2 bit-exact is VALU-bound: every halo row a chunk re-filters is time.  Longer row chunks / shorter tails of the work-item list
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 600 python scripts/ab_opts.py '[{}, {"row_chunks": 12}, {"row_chunks": 9}, {"row_chunks": 6}, {"row_chunks": 12, "work_tail": 100}, {"row_chunks": 9, "work_tail": 100}, {"work_tail": 100}, {"work_tail_div": 2}, {"row_chunks": 12, "work_tail_div": 2}]' '' 256:rgba:3840:2160:1920:1080 4 10 > gpurun_out/r04/call37_cfg2_chunks.txt 2>&1
grep -v "^#" gpurun_out/r04/call37_cfg2_chunks.txt | cut -c1-200
