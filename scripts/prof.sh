#!/bin/bash
# usage: scripts/prof.sh <tag> [bench.py args...]   -> gpurun_out/prof_<tag>/{trace,pmcN}/...csv + summary
# Counter passes are separate runs (--pmc with --kernel-trace only), as the MI355X guide prescribes.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="--steps 3 --warmup 1 --no-cpu-baseline $@"
# the trace pass runs the bench's default step counts so that its average kernel duration is comparable with the bench line
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline $@ > $out/trace.log 2>&1 || exit 1
[ -n "$PROF_TRACE_ONLY" ] && { cd $root && python3 scripts/prof_summary.py $out > $out/summary.txt 2>&1; cat $out/summary.txt; exit 0; }
i=0
for set in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32" \
  "FETCH_SIZE" "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_64B_sum" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum" ; do   # r05: the memory-side requests by size and target (DRAM-typed vs all; no MALL counter exists)
  i=$((i+1))
  echo "[prof] pass $i: $set"
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc$i -- python3 $root/bench.py $args > $out/pmc$i.log 2>&1 || echo "pass $i failed" >> $out/fail.txt
done
cd $root && python3 scripts/prof_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
