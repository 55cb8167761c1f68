#!/bin/bash
# usage: build_variant.sh <tag> "<extra -D flags>" [pixels...]: rebuilds the run-kernel TUs of the given pixel formats (default 2 7) with extra
# flags (and the symmetric-table TUs of the formats in $SYM_PIX) and links them with the in-tree objects into picha_amd/variants/libpicha_hip_<tag>.so (A/B of BUILDS on one box: PICHA_HIP_LIB)
set -e
tag=$1; flags=$2; shift 2; pix=${@:-2 7}
cd "$(dirname "$0")/../picha_amd/csrc"
mkdir -p build/var_$tag ../variants
objs=""
for n in 0 1 2 3 4 5 6 7; do
  if [[ " $pix " == *" $n "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -fno-slp-vectorize -mllvm -pragma-unroll-threshold=200000 -DPICHA_RUN_PIXEL=$n $flags -c kernels_run.hip -o build/var_$tag/run_p$n.o &
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
streamobjs=""
for n in 0 1 2 3 4 5 6 7; do
  if [[ " $STREAM_PIX " == *" $n "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -fno-slp-vectorize -DPICHA_STREAM_PIXEL=$n $flags -c kernels_stream.hip -o build/var_$tag/stream_p$n.o &
    streamobjs="$streamobjs build/var_$tag/stream_p$n.o"
  else streamobjs="$streamobjs build/stream_p$n.o"; fi
done
spanobjs=""
for n in 0 1 2 3 4 5 6 7; do
  if [[ " $SPAN_PIX " == *" $n "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -fno-slp-vectorize -DPICHA_SPAN_PIXEL=$n $flags -c kernels_span.hip -o build/var_$tag/span_p$n.o &
    spanobjs="$spanobjs build/var_$tag/span_p$n.o"
  else spanobjs="$spanobjs build/span_p$n.o"; fi
done
kobj=build/kernels.o
if [ -n "$KERNELS_TU" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -fno-slp-vectorize $flags -c kernels.hip -o build/var_$tag/kernels.o &
  kobj=build/var_$tag/kernels.o
fi
rowsobjs=""
for n in 0 1 2 3 4 5 6 7 8 9 10 11; do
  if [[ " $ROWS_VAR " == *" $n "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -fno-slp-vectorize -DPICHA_ROWS_VARIANT=$n $flags -c kernels_rows.hip -o build/var_$tag/rows_v$n.o &
    rowsobjs="$rowsobjs build/var_$tag/rows_v$n.o"
  else rowsobjs="$rowsobjs build/rows_v$n.o"; fi
done
wait
rest="$(ls build/*.o | grep -v "build/run_p" | grep -v "build/sym_p" | grep -v "build/rows_v" | grep -v "build/stream_p" | grep -v "build/span_p" | grep -v "build/kernels.o") $symobjs $rowsobjs $streamobjs $spanobjs $kobj"
# the library is linked into variants_stash/ (not shipped: gpurun's snapshot limit is 512 MiB and a build is 80 MB) and shipped as variants/<name>.so.xz (8 MB);
# scripts/unpack_variants.sh restores variants/*.so on the GPU box
mkdir -p ../../gpurun_out/variants_stash
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_out/variants_stash/libpicha_hip_$tag.so $rest $objs
xz -1 -T4 -k -c ../../gpurun_out/variants_stash/libpicha_hip_$tag.so > ../variants/libpicha_hip_$tag.so.xz
ls -la ../variants/libpicha_hip_$tag.so.xz
