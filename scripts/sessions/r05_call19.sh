#!/bin/bash
# r05 call 19: 3x upscales on the ring kernel (new instantiations): parity test, then against the kernels they ran on before ("run" = 2), five formats x filters, 64 images 960x540 -> 2880x1620
export PYTHONPATH=. TMPDIR=/tmp
out=gpurun_out/r05; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_chain_and_helpers.py -x -q -k "three_times_upscale or integer_upscale_kernel" 2>&1 | tail -8
for pix in rgba grey greya r16 r16g16; do for filt in "" cubic; do
timeout -k 10 300 python scripts/ab_shared.py --case 64:$pix:960:540:2880:1620:$filt --rounds 3 --reps 10 before=.:run=2 up3=. up3_pl3=.:per_lane=3 up3_pl6=.:per_lane=6 up3_pl12=.:per_lane=12 2>&1 | grep -v amdgpu.ids | cut -c1-260
done; done > $out/call19_up3.txt 2>&1
cat $out/call19_up3.txt
