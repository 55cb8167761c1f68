#!/bin/bash
o=gpurun_out/r03/call13; mkdir -p $o
for px in rgba rgb grey; do
  python scripts/ab_opts.py '[{}, {"up":1,"work":3}, {"row_chunks":4}, {"row_chunks":40}]' '' 256:$px:960:540:1920:1080 3 10
  python scripts/ab_opts.py '[{}, {"up":1,"work":3}]' '' 64:$px:1920:1080:3840:2160 3 10
done > $o/upscale_big.txt 2>&1
python scripts/ab_opts.py '[{}, {"per_lane":1}, {"per_lane":2,"sym":2}, {"row_chunks":4}, {"row_chunks":8}, {"numerics":1}]' '' 256:rgba:640:480:320:240 3 20 > $o/cfg1_size.txt 2>&1
python scripts/ab_opts.py '[{}, {"per_lane":1}, {"row_chunks":4}, {"row_chunks":8}]' '' 2048:rgba:640:480:320:240 3 10 >> $o/cfg1_size.txt 2>&1
cat $o/upscale_big.txt $o/cfg1_size.txt
