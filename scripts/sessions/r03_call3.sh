#!/bin/bash
# r03 call 3: work-item list / persistent launch against the plain grid; clocks and power while the exact and fma kernels run
o=gpurun_out/r03/call3; mkdir -p $o
python scripts/ab_opts.py '[{}, {"work":1}, {"work":2}, {"work":2,"row_chunks":9}, {"work":2,"row_chunks":36}, {"work":1,"row_chunks":9}, {"work":2,"block_waves":4}, {"work":2,"block_waves":1}, {"numerics":1}, {"numerics":1,"work":1}, {"numerics":1,"work":2}, {"numerics":1,"work":2,"row_chunks":18}, {"numerics":1,"work":2,"block_waves":2}]' '' 256:rgba:3840:2160:1920:1080 3 10 > $o/ab_cfg2.txt 2>&1
python scripts/ab_opts.py '[{}, {"xch":1}, {"xch":1,"work":1}, {"xch":1,"work":2}, {"xch":1,"work":2,"row_chunks":8}, {"xch":1,"work":2,"row_chunks":32}, {"work":2}, {"numerics":1,"xch":1}, {"numerics":1,"xch":1,"work":2}]' '{"filter":"mitchel"}' 32:r16g16b16a16:7680:4320:1920:1080 3 10 > $o/ab_cfg5.txt 2>&1
# clocks / power under load
( for i in 1 2 3 4 5 6; do sleep 1.5; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power" ; echo ---; done ) > $o/smi_exact.txt &
python bench.py --no-cpu-baseline --steps 3000 --warmup 3 --no-parity > $o/bench_long_exact.json 2>/dev/null
wait
( for i in 1 2 3 4 5 6; do sleep 1.5; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power" ; echo ---; done ) > $o/smi_fma.txt &
python bench.py --no-cpu-baseline --steps 3000 --warmup 3 --no-parity --numerics fma > $o/bench_long_fma.json 2>/dev/null
wait
cat $o/ab_cfg2.txt $o/ab_cfg5.txt; tail -12 $o/smi_exact.txt; tail -12 $o/smi_fma.txt
