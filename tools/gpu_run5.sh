#!/bin/bash
# online stream under rocprofv3 kernel stats: graph replay against eager launches
set -e
mkdir -p gpurun_out/gr
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PI3_DEV_GRAPH=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gr/prof_graph -o g -- python3 $R/tools/dev_online_stream.py 600 > $R/gpurun_out/gr/pg.log 2>&1
PI3_DEV_GRAPH=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gr/prof_eager -o e -- python3 $R/tools/dev_online_stream.py 600 > $R/gpurun_out/gr/pe.log 2>&1
cd $R
for f in $(find gpurun_out/gr/prof_graph gpurun_out/gr/prof_eager -name "*kernel_stats.csv"); do echo == $f; head -8 $f | cut -c1-150; done
find gpurun_out/gr -name "*_kernel_trace.csv" -size +60M -delete
grep RESULT gpurun_out/gr/pg.log gpurun_out/gr/pe.log
