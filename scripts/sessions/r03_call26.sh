#!/bin/bash
o=gpurun_out/r03/call26; mkdir -p $o
python scripts/ab_opts.py '[{"work":3}, {}, {"work_tail":0}, {"numerics":1,"work":3}, {"numerics":1}, {"numerics":1,"work_tail":0}, {"numerics":1,"per_lane":3}]' '' 256:rgba:3840:2160:1920:1080 4 10 > $o/ab_cfg2_tail16.txt 2>&1
cat $o/ab_cfg2_tail16.txt
