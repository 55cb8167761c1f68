#!/bin/bash
# r05 call 13: memory-side counters of cfg2 (list form): are the writes whole 64-byte requests?  how many read requests go to DRAM (vs the fabric total)?
export PYTHONPATH=. TMPDIR=/tmp
out=gpurun_out/r05; mkdir -p $out
root=$(pwd)
cd /tmp
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCC_EA0_WRREQ_WRITE_DRAM_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $root/gpurun_out/pmcq_mem/pmc$i -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --prewarm-ms 5 > $root/gpurun_out/pmcq_mem/pmc$i.log 2>&1 || echo "pass $i ($set) failed" >> $root/gpurun_out/pmcq_mem/fail.txt
done
cd $root && python3 scripts/prof_summary.py gpurun_out/pmcq_mem > $out/call13_mem_counters.txt 2>&1
cat $out/call13_mem_counters.txt | cut -c1-140; cat gpurun_out/pmcq_mem/fail.txt 2>/dev/null
