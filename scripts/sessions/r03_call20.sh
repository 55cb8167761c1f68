#!/bin/bash
o=gpurun_out/r03/call20; mkdir -p $o
python scripts/ab_opts.py '[{}, {"xch":1}, {"xch":1,"work":2}, {"xch":3,"row_chunks":18}, {"xch":2}, {"numerics":1}, {"numerics":1,"xch":1}]' '{"filter":"mitchel"}' 32:r16g16b16a16:7680:4320:1920:1080 3 10 > $o/ab_cfg5_rounds.txt 2>&1
python scripts/ab_opts.py '[{}, {"xch":1}, {"xch":1,"work":2}]' '{"filter":"mitchel"}' 128:r16g16b16a16:7680:4320:1920:1080 2 5 > $o/ab_cfg5_128.txt 2>&1
cat $o/ab_cfg5_rounds.txt $o/ab_cfg5_128.txt
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -k "cfg5 or work_item or persistent or selection" > $o/pytest.txt 2>&1; tail -3 $o/pytest.txt
