#!/bin/bash
# MoGe forward alone: times + rocprofv3 per-kernel stats
mkdir -p gpurun_out/mg
timeout -k 10 200 python tools/dev_moge_time.py > gpurun_out/mg/time.log 2>&1; cat gpurun_out/mg/time.log | grep "eager\|graph"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/mg/prof --output-format csv -- python tools/dev_moge_time.py > /dev/null 2>&1
cp gpurun_out/mg/prof/*/*kernel_stats.csv gpurun_out/mg/kernel_stats.csv
head -30 gpurun_out/mg/kernel_stats.csv | cut -c1-180
