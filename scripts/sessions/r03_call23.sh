#!/bin/bash
o=gpurun_out/r03/call23; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > $o/pytest.txt 2>&1 || { tail -30 $o/pytest.txt; exit 1; }
tail -2 $o/pytest.txt
python scripts/ab_opts.py '[{"work":3}, {}, {"work":1,"work_tail":300}, {"numerics":1,"work":3}, {"numerics":1}]' '{"filter":"cubic"}' 256:rgba:3840:2160:1920:1080 3 10 > $o/ab_7tap_items.txt 2>&1
python scripts/ab_opts.py '[{"work":3}, {}]' '' 192:rgb:3840:2160:1920:1080 3 10 > $o/ab_rgb_items.txt 2>&1
python scripts/ab_opts.py '[{"work":3}, {}]' '' 192:grey:3840:2160:1920:1080 3 10 > $o/ab_grey_items.txt 2>&1
python scripts/ab_opts.py '[{"work":3}, {}]' '' 2048:rgba:640:480:320:240 3 10 > $o/ab_small_items.txt 2>&1
cat $o/ab_*.txt
