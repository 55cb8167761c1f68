#!/bin/bash
# r04 call 15: what do the output stores cost in cfg3b (run kernel) and in the fused cfg3 (row-staged kernel)?  builds without them
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/call15_nostore.txt; : > $out
for rnd in 1 2; do for lib in "" picha_amd/variants/libpicha_hip_nostore.so; do
  echo "== round $rnd lib=${lib:-in-tree}" >> $out
  PICHA_HIP_LIB=$lib timeout -k 10 200 python scripts/ab_opts.py '[{},{"numerics":1}]' '{"filter":"lanczos"}' 256:grey:3840:2160:512:512 2 10 2>/dev/null >> $out
  for w in cfg3_4k_rgb_to_grey_to_512_lanczos_chained edge_jpeg_thumb_4k_rgb_to_512x288; do
  PICHA_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload $w 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); o=d.get('other_numerics') or {}
        print('%-44s %-62s ms/launch %8.4f frac %.4f parity %s | fma %s' % ('$w', d['config']['kernel'][:62], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['parity_max_code_diff_vs_oracle_image0'], ('%.4f ms frac %.4f' % (o['avg_launch_ms'], o['roofline_frac'])) if o else '-'))
" >> $out
  done
done; done
cat $out
