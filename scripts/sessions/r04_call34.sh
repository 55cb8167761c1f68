#!/bin/bash
# r04 call 34: 16-bit samples unpacked without the SDWA multiply of the denormal form too (nodenorm build: and / shift + v_cvt_f32_u32 + v_mul_f32) against the build that keeps it for 16-bit (greycvt)
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/call34_u16_unpack.txt; : > $out
for rnd in 1 2; do for lib in picha_amd/variants/libpicha_hip_greycvt.so picha_amd/variants/libpicha_hip_nodenorm.so; do
  echo "== round $rnd lib=$lib" >> $out
  for w in cfg5_8k_rgba16_to_1080p_mitchel edge_deep_png_4k_rgba16_to_1080p_rgba up_960x540_r16_to_1080p; do
  PICHA_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload $w 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); o=d.get('other_numerics') or {}
        print('%-44s %-66s ms/launch %8.4f frac %.4f parity %s | fma %s' % ('$w', d['config']['kernel'][:66], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('parity',{}).get('max_code_diff'), ('%.4f ms frac %.4f' % (o['avg_launch_ms'], o['roofline_frac'])) if o else '-'))
" >> $out
  done
  for shape in 64:r16g16b16a16:3840:2160:1920:1080 128:r16:3840:2160:1920:1080 128:r16g16:3840:2160:1920:1080 64:r16g16b16:3840:2160:1920:1080 64:r16g16b16a16:3840:2160:1280:720 64:r16g16b16:3840:2160:512:288 128:r16:3840:2160:512:288; do
  PICHA_HIP_LIB=$lib timeout -k 10 200 python scripts/ab_opts.py '[{}]' '' $shape 2 10 2>/dev/null | grep -v "^#" | sed "s/^/$shape  /" >> $out
  done
done; done
cat $out
