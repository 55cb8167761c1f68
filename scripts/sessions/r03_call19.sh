#!/bin/bash
o=gpurun_out/r03/call19; mkdir -p $o
python scripts/ab_opts.py '[{}, {"xch":3}, {"xch":3,"work":3}, {"xch":3,"work":3,"row_chunks":12}, {"xch":3,"work":2,"row_chunks":24}, {"work":3,"row_chunks":12}, {"numerics":1}, {"numerics":1,"xch":3}, {"xch":2}]' '{"filter":"mitchel"}' 32:r16g16b16a16:7680:4320:1920:1080 3 10 > $o/ab_cfg5_sup.txt 2>&1
cat $o/ab_cfg5_sup.txt
