#!/bin/bash
# r03 call 5: new parity tests (work items, persistent, upscale accumulator shapes), fuzz with the new knobs, upscale survey
o=gpurun_out/r03/call5; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "work_item or persistent or upscales" > $o/pytest_new.txt 2>&1 || { tail -30 $o/pytest_new.txt; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -x -q > $o/pytest_fuzz.txt 2>&1 || { tail -30 $o/pytest_fuzz.txt; exit 1; }
for px in rgb grey greya rgba r16; do
  python scripts/ab_opts.py '[{"up":2}, {}, {"numerics":1,"up":2}, {"numerics":1}]' '' 64:$px:960:540:1920:1080 3 20
done > $o/survey_upscale.txt 2>&1
python scripts/ab_opts.py '[{"up":2}, {}]' '{"filter":"cubic"}' 64:rgba:960:540:1920:1080 3 20 >> $o/survey_upscale.txt 2>&1
python scripts/ab_opts.py '[{"up":2}, {}]' '' 64:rgb:640:360:1920:1080 3 20 >> $o/survey_upscale.txt 2>&1
tail -3 $o/pytest_new.txt $o/pytest_fuzz.txt; cat $o/survey_upscale.txt
