#!/bin/bash
# r05 call 4: why are persistent waves slower?  Counters of the two launch forms of cfg2 (fabric traffic, L2, waits); list of memory-side counters this rocprofv3 knows
export PYTHONPATH=. TMPDIR=/tmp
out=gpurun_out/r05; mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -o -i -E "\b(TCC_[A-Z0-9_]*(DRAM|MALL|EA0|IO|GMI)[A-Z0-9_]*|[A-Z_]*MALL[A-Z_0-9]*|[A-Z_]*DRAM[A-Z_0-9]*)\b" | sort -u > $out/call04_counters.txt; wc -l $out/call04_counters.txt
bash scripts/pmc_quick.sh list > /dev/null 2>&1; grep -v "^==\|dispatch\|launches" gpurun_out/pmcq_list/summary.txt | grep -A40 "false, false, fals" | head -40
bash scripts/pmc_quick.sh pw --option work=4 > /dev/null 2>&1; grep -v "^==\|dispatch\|launches" gpurun_out/pmcq_pw/summary.txt | grep -A40 "false, false, fals" | head -40
cat gpurun_out/pmcq_*/fail.txt 2>/dev/null
