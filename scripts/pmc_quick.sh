#!/bin/bash
# usage: scripts/pmc_quick.sh <tag> [bench.py args...]: a few PMC passes (one rocprofv3 run each, --pmc with --kernel-trace only) for comparing two launch forms of one workload:
# fabric traffic, L2 hit / miss, wave residency and waits, L1 stalls -> gpurun_out/pmcq_<tag>/summary.txt
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmcq_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="--steps 3 --warmup 1 --no-cpu-baseline --no-parity --prewarm-ms 5 $@"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" $PMCQ_EXTRA ; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc$i -- python3 $root/bench.py $args > $out/pmc$i.log 2>&1 || echo "pass $i ($set) failed" >> $out/fail.txt
done
cd $root && python3 scripts/prof_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
