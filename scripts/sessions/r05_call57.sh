#!/bin/bash
# r05 call 57: cliff survey of the final library (default options): filters x scale ladder x all eight formats, 64 images, sorted by roof fraction, with the kernel that ran
export PYTHONPATH=. TMPDIR=/tmp
o=gpurun_out/r05c; mkdir -p $o
timeout -k 10 800 python scripts/survey_cliffs.py 64 grey,greya,rgb,rgba,r16,r16g16,r16g16b16,r16g16b16a16 > $o/survey_cliffs_final.txt 2>&1; rc=$?
grep -v amdgpu.ids $o/survey_cliffs_final.txt | head -45 | cut -c1-230
exit $rc
