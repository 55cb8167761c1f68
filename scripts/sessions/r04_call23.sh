#!/bin/bash
# r04 call 23: what do the output stores cost in the accumulator kernel's thumbnail workloads (rgb)?  build without them
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/call23_acc_nostore.txt; : > $out
for rnd in 1 2; do for lib in "" picha_amd/variants/libpicha_hip_accns.so; do
  echo "== round $rnd lib=${lib:-in-tree}" >> $out
  for w in edge_jpeg_thumb_4k_rgb_to_512x288 lanczos_4k_rgb_to_480x270; do
  PICHA_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --workload $w 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); o=d.get('other_numerics') or {}
        print('%-44s %-50s ms/launch %8.4f frac %.4f | fma %s' % ('$w', d['config']['kernel'][:50], d['roofline']['avg_launch_ms'], d['roofline']['frac'], ('%.4f ms frac %.4f' % (o['avg_launch_ms'], o['roofline_frac'])) if o else '-'))
" >> $out
  done
done; done
cat $out
