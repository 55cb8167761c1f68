#!/bin/bash
# r03: edge kernel block size (PICHA_EDGE_ROWS, measurement only)
set -e
for er in 64 32 16; do
  PICHA_EDGE_ROWS=$er PROF_TRACE_ONLY=1 bash scripts/prof.sh run_er$er --workload cfg3b_4k_grey_to_512_lanczos | grep "launches" 
done
