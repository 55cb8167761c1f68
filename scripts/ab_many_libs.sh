#!/bin/bash
# usage: ab_many_libs.sh "<lib1> <lib2> ..." [bench args]: cfg2 bench line per library, two rounds, both numerics
libs=$1; shift
for rnd in 1 2; do for lib in $libs; do for num in exact fma; do
  PICHA_HIP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --numerics $num "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('round $rnd %-40s %-6s %.4f ms  frac %.4f' % ('$(basename $lib)', '$num', d['roofline']['avg_launch_ms'], d['roofline']['frac']))
"
done; done; done
