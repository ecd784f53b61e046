#!/bin/bash
# masks_kernel under rocprofv3 (kernel durations, not launch-to-launch times); PI3_MASKS_ROWS / PI3_MASKS_LDS_KB sweep the strip height
mkdir -p gpurun_out/m
timeout -k 10 100 python tools/dev_masks.py 2>&1 | grep "masks 100\|DIFF"
rocprofv3 --kernel-trace --stats -d gpurun_out/m/prof --output-format csv -- python tools/dev_masks.py > /dev/null 2>&1
cat gpurun_out/m/prof/*/*kernel_stats.csv | grep "masks_kernel\|Name"
# HBM-side traffic of the kernel (separate --pmc passes, as the guide prescribes)
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/m/pmc_$tag --output-format csv -- python tools/dev_masks.py > /dev/null 2>&1 || echo "pmc $tag failed"
done
python tools/pmc_summary.py masks_kernel 0.04 gpurun_out/m/pmc_* > gpurun_out/m/masks_pmc.csv 2>&1
cat gpurun_out/m/masks_pmc.csv
