#!/bin/bash
# r05 call 32: GPU suite on the restored tree; what the edge kernel costs the upscale / ring / run cells (run = 3: no edge pass, measurement only, edge columns wrong)
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/gputest_restored.log 2>&1 || { tail -40 $o/gputest_restored.log; exit 1; }
tail -2 $o/gputest_restored.log
{
python scripts/ab_shared.py --case 64:rgba:960:540:1920:1080 --no-oracle default=. noedge=.:run=3
python scripts/ab_shared.py --case 64:r16:960:540:1920:1080 --no-oracle default=. noedge=.:run=3
python scripts/ab_shared.py --case 256:rgba:480:270:1920:1080 --no-oracle default=. noedge=.:run=3
python scripts/ab_shared.py --case lanczos2x --no-oracle default=. noedge=.:run=3
python scripts/ab_shared.py --case cfg3b --no-oracle default=. noedge=.:run=3
python scripts/ab_shared.py --case cfg5 --no-oracle default=. noedge=.:run=3
python scripts/ab_shared.py --case jpegthumb --no-oracle default=. noedge=.:run=3
} > $o/ab_edge_cost.txt 2>&1
grep -v amdgpu.ids $o/ab_edge_cost.txt | cut -c1-200
