#!/bin/bash
# r05 call 40: translation-cache counters of cfg2's kernel writing into a slow and into a fast destination placement (same process, same source)
export TMPDIR=/tmp
root=$(pwd); o=$root/gpurun_out/r05b/tlb; mkdir -p $o
cd /tmp
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum"; do
  i=$((i+1))
  PYTHONPATH=$root timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $o/pmc$i -- python3 $root/scripts/r05_placement_tlb.py 10 > $o/pmc$i.log 2>&1
  grep "candidates ms" $o/pmc$i.log
  python3 $root/scripts/r05_placement_tlb_parse.py $o/pmc$i
done
