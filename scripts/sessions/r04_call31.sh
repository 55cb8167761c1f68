#!/bin/bash
# r04 call 31: ring kernel (lanczos 2x rgba, 131 VGPRs = 3 waves per SIMD) built for 4 waves per SIMD (amdgpu_waves_per_eu(4,8): 128 VGPRs, 16 B scratch)
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/call31_ring_w4.txt; : > $out
for rnd in 1 2 3; do for lib in "" picha_amd/variants/libpicha_hip_ringw4.so; do
  echo "== round $rnd lib=${lib:-in-tree}" >> $out
  for w in lanczos_4k_rgba_to_1080p up_960x540_rgba_to_1080p; do
  PICHA_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload $w 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); o=d.get('other_numerics') or {}
        print('%-30s %-66s ms/launch %8.4f frac %.4f parity %s | fma %s' % ('$w', d['config']['kernel'][:66], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('parity',{}).get('max_code_diff'), ('%.4f ms frac %.4f' % (o['avg_launch_ms'], o['roofline_frac'])) if o else '-'))
" >> $out
  done
done; done
cat $out
