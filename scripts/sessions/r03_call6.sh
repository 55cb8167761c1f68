#!/bin/bash
o=gpurun_out/r03/call6; mkdir -p $o
for px in rgb grey rgba r16 r16g16 r16g16b16 r16g16b16a16; do
  python scripts/ab_opts.py '[{"up":2}, {"work":3}, {"work":1}, {"work":2}]' '' 64:$px:960:540:1920:1080 3 20
done > $o/survey_upscale_work.txt 2>&1
cat $o/survey_upscale_work.txt
