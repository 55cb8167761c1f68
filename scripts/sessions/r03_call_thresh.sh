#!/bin/bash
# r03: where the run kernel's short-window shapes start to pay: batch sizes 16 / 32 / 64 (the planner asks for >= 96 images)
mkdir -p gpurun_out
: > gpurun_out/thresh.txt
for n in 16 32 64; do
  for spec in "grey:1920:1080:1280:720" "greya:1920:1080:1280:720" "grey:3840:2160:1536:864" "grey:3840:2160:1280:720" "greya:3840:2160:960:540" "grey:3840:2160:768:432" "grey:3840:2160:640:360"; do
    echo "## n=$n $spec" >> gpurun_out/thresh.txt
    timeout -k 10 120 python scripts/ab_opts.py '[{"run":2},{"run":1}]' '{}' $n:$spec 3 10 >> gpurun_out/thresh.txt 2>&1 || echo FAILED >> gpurun_out/thresh.txt
  done
done
grep -v amdgpu gpurun_out/thresh.txt | awk '/^##/{h=$0; next} {print h " | " $1 " min " $(NF-4)}' | cut -c1-150
