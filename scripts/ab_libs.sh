#!/bin/bash
# A/B of two builds of libpicha_hip.so on one box: alternates the per-workload bench between $1 (old) and the in-tree build.
old=$1; shift
for rnd in 1 2; do
  for lib in "$old" ""; do
    echo "== round $rnd lib=${lib:-in-tree}"
    PICHA_HIP_LIB=$lib bash scripts/bench_workloads.sh "$@" || exit 1
    cp gpurun_out/bench_workloads.txt gpurun_out/ab_libs_${rnd}_$( [ -z "$lib" ] && echo new || echo old ).txt
  done
done
