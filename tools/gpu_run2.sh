#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2b/tests.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r2b/tests.log
tail -5 gpurun_out/r2b/tests.log
timeout -k 10 120 python tools/dev_gemm.py > gpurun_out/r2b/gemm.log 2>&1; tail -7 gpurun_out/r2b/gemm.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err; tail -c 1500 gpurun_out/r2b/bench.json
