#!/bin/bash
# usage: r04_build_variant.sh <tag> "<extra -D flags>" [pixels...]: rebuilds the run-kernel TUs of the given pixel formats (default 2 7) with extra
# flags (and the symmetric-table TUs of the formats in $SYM_PIX) and links them with the in-tree objects into picha_amd/variants/libpicha_hip_<tag>.so (A/B of BUILDS on one box: PICHA_HIP_LIB)
set -e
tag=$1; flags=$2; shift 2; pix=${@:-2 7}
cd "$(dirname "$0")/../picha_amd/csrc"
mkdir -p build/var_$tag ../variants
objs=""
for n in 0 1 2 3 4 5 6 7; do
  if [[ " $pix " == *" $n "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -fno-slp-vectorize -DPICHA_RUN_PIXEL=$n $flags -c kernels_run.hip -o build/var_$tag/run_p$n.o &
    objs="$objs build/var_$tag/run_p$n.o"
  else objs="$objs build/run_p$n.o"; fi
done
symobjs=""
for n in 0 1 2 3 4 5 6 7; do
  if [[ " $SYM_PIX " == *" $n "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -fno-slp-vectorize -DPICHA_SYM_PIXEL=$n $flags -c kernels_sym.hip -o build/var_$tag/sym_p$n.o &
    symobjs="$symobjs build/var_$tag/sym_p$n.o"
  else symobjs="$symobjs build/sym_p$n.o"; fi
done
rowsobjs=""
for n in 0 1 2 3 4 5 6 7 8 9 10 11; do
  if [[ " $ROWS_VAR " == *" $n "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -fno-slp-vectorize -DPICHA_ROWS_VARIANT=$n $flags -c kernels_rows.hip -o build/var_$tag/rows_v$n.o &
    rowsobjs="$rowsobjs build/var_$tag/rows_v$n.o"
  else rowsobjs="$rowsobjs build/rows_v$n.o"; fi
done
wait
rest="$(ls build/*.o | grep -v "build/run_p" | grep -v "build/sym_p" | grep -v "build/rows_v") $symobjs $rowsobjs"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libpicha_hip_$tag.so $rest $objs
ls -la ../variants/libpicha_hip_$tag.so
