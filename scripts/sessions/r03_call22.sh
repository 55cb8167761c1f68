#!/bin/bash
o=gpurun_out/r03/call22; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "upscale or subdword or four_outputs" > $o/pytest.txt 2>&1 || { tail -30 $o/pytest.txt; exit 1; }
tail -2 $o/pytest.txt
for px in grey greya rgb; do
  python scripts/ab_opts.py '[{}, {"numerics":1}]' '' 64:$px:960:540:1920:1080 3 20
  python scripts/ab_opts.py '[{}]' '{"filter":"cubic"}' 64:$px:960:540:1920:1080 3 20
done > $o/upscale_packed.txt 2>&1
cat $o/upscale_packed.txt
