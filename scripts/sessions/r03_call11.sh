#!/bin/bash
o=gpurun_out/r03/call11; mkdir -p $o
python scripts/ab_opts.py '[{}, {"xch":3}, {"xch":3,"per_lane":2}, {"xch":3,"per_lane":4}, {"xch":1}, {"numerics":1}, {"numerics":1,"xch":3}, {"numerics":1,"xch":3,"per_lane":3}]' '' 256:rgba:3840:2160:1920:1080 3 10 > $o/ab_cfg2_xf.txt 2>&1
python scripts/ab_opts.py '[{}, {"xch":3,"per_lane":2}, {"xch":3,"per_lane":3}, {"numerics":1}, {"numerics":1,"xch":3}]' '{"filter":"cubic"}' 256:rgba:3840:2160:1920:1080 3 10 > $o/ab_cfg2_7tap_xf.txt 2>&1
cat $o/*.txt
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $o/pytest_gpu_full.txt 2>&1; tail -5 $o/pytest_gpu_full.txt
