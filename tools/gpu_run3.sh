#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2c
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2c/tests.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r2c/tests.log
tail -5 gpurun_out/r2c/tests.log
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err; echo "bench rc=$?"; tail -c 3500 gpurun_out/r2c/bench.json; tail -5 gpurun_out/r2c/bench.err
