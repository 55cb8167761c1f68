#!/bin/bash
# r03 call 2: neighbour-exchange (xch) variants against the r02 defaults, every variant checked against the oracle
o=gpurun_out/r03/call2; mkdir -p $o
python scripts/ab_opts.py '[{}, {"xch":1}, {"xch":1,"block_waves":4}, {"xch":1,"per_lane":2}, {"xch":1,"per_lane":4}, {"xch":1,"per_lane":4,"block_waves":4}, {"numerics":1}, {"numerics":1,"xch":1}, {"numerics":1,"xch":1,"per_lane":3}, {"numerics":1,"xch":1,"per_lane":4}, {"numerics":1,"xch":1,"per_lane":3,"block_waves":2}]' '' 256:rgba:3840:2160:1920:1080 3 10 > $o/ab_cfg2.txt 2>&1
python scripts/ab_opts.py '[{}, {"xch":1,"per_lane":2}, {"xch":1,"per_lane":3}, {"sym":2,"xch":1,"per_lane":4}, {"numerics":1}, {"numerics":1,"xch":1}, {"numerics":1,"xch":1,"per_lane":3}]' '{"filter":"cubic"}' 256:rgba:3840:2160:1920:1080 3 10 > $o/ab_cfg2_7tap.txt 2>&1
python scripts/ab_opts.py '[{}, {"xch":1}, {"xch":1,"row_chunks":8}, {"xch":1,"row_chunks":32}, {"numerics":1}, {"numerics":1,"xch":1}]' '{"filter":"mitchel"}' 32:r16g16b16a16:7680:4320:1920:1080 3 10 > $o/ab_cfg5.txt 2>&1
python scripts/ab_opts.py '[{}, {"xch":1}, {"sym":2,"xch":1}, {"numerics":1}, {"numerics":1,"xch":1}]' '' 64:rgba:7680:4320:1920:1080 3 10 > $o/ab_rgba_4x.txt 2>&1
python scripts/ab_opts.py '[{}, {"xch":1}, {"sym":2,"xch":1}, {"numerics":1}, {"numerics":1,"xch":1}]' '' 128:rgba:3840:2160:1280:720 3 10 > $o/ab_rgba_3x.txt 2>&1
cat $o/*.txt
