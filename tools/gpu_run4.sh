#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2d
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2d/tests.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r2d/tests.log
tail -4 gpurun_out/r2d/tests.log
python tools/dev_fromdisk.py > gpurun_out/r2d/fromdisk.log 2>&1; grep -n "###\|host time" gpurun_out/r2d/fromdisk.log | tail -4
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err; echo "bench rc=$?"; tail -c 3600 gpurun_out/r2d/bench.json
