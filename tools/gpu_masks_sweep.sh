#!/bin/bash
# masks_kernel under rocprofv3 (kernel durations, not launch-to-launch times); PI3_MASKS_ROWS / PI3_MASKS_LDS_KB sweep the strip height
mkdir -p gpurun_out/m
timeout -k 10 100 python tools/dev_masks.py 2>&1 | grep "masks 100\|DIFF"
rocprofv3 --kernel-trace --stats -d gpurun_out/m/prof --output-format csv -- python tools/dev_masks.py > /dev/null 2>&1
cat gpurun_out/m/prof/*/*kernel_stats.csv | grep "masks_kernel\|Name"
