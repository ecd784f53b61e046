#!/bin/bash
# end of round 3: rocprofv3 kernel stats of the bench command + the default bench line (PMC passes: gpu_profile_r03.sh)
set -o pipefail
mkdir -p gpurun_out/r3f
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r3f/prof --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3f/prof_bench.json 2> gpurun_out/r3f/prof_bench.err || echo "kernel-trace run failed"
find gpurun_out/r3f/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3f/bench_kernel_stats.csv \;
head -8 gpurun_out/r3f/bench_kernel_stats.csv | cut -c1-150
timeout -k 10 900 python bench.py > gpurun_out/r3f/bench_line.json 2> gpurun_out/r3f/bench_line.err || echo "bench failed"
tail -c 300 gpurun_out/r3f/bench_line.json
find gpurun_out/r3f -name "*kernel_trace.csv" -size +30M -delete
