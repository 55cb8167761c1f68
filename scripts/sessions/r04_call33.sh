#!/bin/bash
# r04 call 33: 8-bit grey unpacked by v_cvt_f32_ubyteN + v_mul_f32 (greycvt build) instead of the one-instruction SDWA multiply of the denormal form (base build):
# the r04 VALU ubench says an SDWA instruction costs ~8.7 cycles of issue against 2.45 + a conversion that overlaps (profiles/r04_ubench_valu_rates.txt)
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/call33_grey_unpack.txt; : > $out
for rnd in 1 2; do for lib in picha_amd/variants/libpicha_hip_base.so picha_amd/variants/libpicha_hip_greycvt.so; do
  echo "== round $rnd lib=$lib" >> $out
  for w in cfg3b_4k_grey_to_512_lanczos cfg3_4k_rgb_to_grey_to_512_lanczos_chained; do
  PICHA_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload $w 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); o=d.get('other_numerics') or {}
        print('%-44s %-66s ms/launch %8.4f frac %.4f parity %s | fma %s' % ('$w', d['config']['kernel'][:66], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('parity',{}).get('max_code_diff'), ('%.4f ms frac %.4f' % (o['avg_launch_ms'], o['roofline_frac'])) if o else '-'))
" >> $out
  done
  for shape in 256:grey:3840:2160:1920:1080 192:grey:3840:2160:512:288 128:grey:3840:2160:1280:720 256:grey:1920:1080:1280:720 128:grey:3840:2160:960:540 256:grey:960:540:1920:1080; do
  PICHA_HIP_LIB=$lib timeout -k 10 200 python scripts/ab_opts.py '[{}, {"numerics": 1}]' '' $shape 2 10 2>/dev/null | grep -v "^#" | sed "s/^/$shape  /" >> $out
  done
done; done
cat $out
