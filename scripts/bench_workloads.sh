#!/bin/bash
# one bench.py line per BASELINE.json config, the colour-convert and codec-edge workloads: kernel, ms, frac, parity, fma flavour
out=${BENCH_OUT:-gpurun_out/bench_workloads.txt}; : > $out
for w in cfg2_4k_rgba_to_1080p_cubic cfg2_4k_rgba_to_1080p_cubic_scale1 cfg1_640x480_rgba_to_320x240 cfg3a_4k_rgb_to_grey cfg3b_4k_grey_to_512_lanczos cfg3_4k_rgb_to_grey_to_512_lanczos_chained cfg5_8k_rgba16_to_1080p_mitchel cc_4k_rgba_to_rgba16 edge_jpeg_4k_rgba_to_rgb edge_jpeg_4k_rgba16_to_rgb edge_webp_4k_grey_to_rgb edge_png_4k_rgba16_to_rgba edge_jpeg_thumb_4k_rgb_to_512x288 edge_jpeg_4k_rgb_to_1080p edge_deep_png_4k_rgba16_to_1080p_rgba up_960x540_rgba_to_1080p up_960x540_r16_to_1080p up_960x540_rgba16_to_1080p up_480x270_rgba_to_1080p lanczos_4k_rgba_to_1080p lanczos_4k_rgb_to_480x270; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload $w "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); o=d.get('other_numerics') or {}
        print('%-44s %-62s ms/launch %8.4f GB/s %7.1f frac %.4f parity %s | fma %s' % ('$w', d['config']['kernel'][:62], d['roofline']['avg_launch_ms'], d['roofline']['achieved'], d['roofline']['frac'], d['config']['parity_max_code_diff_vs_oracle_image0'], ('%.4f ms frac %.4f' % (o['avg_launch_ms'], o['roofline_frac'])) if o else '-'))
" >> $out
done
cat $out
