#!/bin/bash
# kernel stats of the bundle adjustment at N = 100, K = 200
mkdir -p gpurun_out/ba
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ba/prof -o ba -- python3 $R/tools/dev_ba_full.py > $R/gpurun_out/ba/run.log 2>&1
cd $R
grep -v "^$" gpurun_out/ba/run.log | tail -6 | cut -c1-300
f=$(find gpurun_out/ba/prof -name "*kernel_stats.csv" | head -1)
head -16 $f | cut -c1-170
find gpurun_out/ba -name "*_kernel_trace.csv" -size +40M -delete
