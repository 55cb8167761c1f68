#!/bin/bash
# r03 call 4: item-list parameters, SLP-vectorised build vs scalar build, store-shaped convert for 3-byte destinations
o=gpurun_out/r03/call4; mkdir -p $o
python -m pytest tests/test_gpu_parity.py -x -q -k "convert or color" > $o/pytest_convert.txt 2>&1
python scripts/ab_opts.py '[{}, {"work":1}, {"work":1,"block_waves":4}, {"work":1,"work_tail":250}, {"work":1,"work_tail":60}, {"work":1,"work_tail_div":8}, {"work":1,"work_tail_div":2}, {"work":1,"row_chunks":27}, {"work":1,"row_chunks":14}, {"numerics":1,"work":1,"block_waves":2}, {"numerics":1,"work":1,"row_chunks":27}]' '' 256:rgba:3840:2160:1920:1080 3 10 > $o/ab_cfg2_items.txt 2>&1
bash scripts/ab_many_libs.sh "picha_amd/libpicha_hip.so picha_amd/libpicha_hip_slp.so" > $o/ab_slp_cfg2.txt 2>&1
bash scripts/ab_many_libs.sh "picha_amd/libpicha_hip.so picha_amd/libpicha_hip_slp.so" --workload cfg5_8k_rgba16_to_1080p_mitchel --option xch=1 --option work=2 > $o/ab_slp_cfg5.txt 2>&1
bash scripts/ab_many_libs.sh "picha_amd/libpicha_hip.so picha_amd/libpicha_hip_slp.so" --workload cfg2_4k_rgba_to_1080p_cubic_scale1 > $o/ab_slp_cfg2_7tap.txt 2>&1
for w in edge_webp_4k_grey_to_rgb cfg3a_4k_rgb_to_grey cc_4k_rgba_to_rgba16; do python bench.py --no-cpu-baseline --workload $w 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-34s %-44s %.4f ms frac %.4f parity %s' % ('$w', d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['parity_max_code_diff_vs_oracle_image0']))
"; done > $o/convert_bench.txt 2>&1
tail -3 $o/pytest_convert.txt; cat $o/ab_cfg2_items.txt $o/ab_slp_*.txt $o/convert_bench.txt
