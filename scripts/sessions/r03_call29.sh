#!/bin/bash
o=gpurun_out/r03/call29; mkdir -p $o
for px in grey rgb greya rgba; do
  python scripts/ab_opts.py '[{}, {"row_chunks":45}, {"row_chunks":90}, {"row_chunks":180}, {"row_chunks":360}]' '' 64:$px:960:540:1920:1080 3 20
  python scripts/ab_opts.py '[{}, {"row_chunks":45}, {"row_chunks":90}]' '' 256:$px:960:540:1920:1080 3 10
done > $o/upscale_chunks.txt 2>&1
cat $o/upscale_chunks.txt
