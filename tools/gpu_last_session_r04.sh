#!/bin/bash
# round 4, last GPU session: long seeded fuzz of the post-network entry points against the oracle,
# the randomised shape sweep of GEMM / attention / LayerNorm, and race screens of every GEMM form that changed this round
# (shipped kernel with the new epilogues; the four-wave variants).  Usage: bash tools/gpu_last_session_r04.sh [a|b]
set -o pipefail
mkdir -p gpurun_out/r4last
part=${1:-a}
if [ "$part" = "a" ]; then
  FUZZ_SECONDS=480 timeout -k 10 700 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -s > gpurun_out/r4last/fuzz_oracle.log 2>&1 || { tail -5 gpurun_out/r4last/fuzz_oracle.log; exit 1; }
  grep "^fuzz\|passed\|failed" gpurun_out/r4last/fuzz_oracle.log
  timeout -k 10 300 python tools/fuzz_kernels.py 150 > gpurun_out/r4last/fuzz_kernels.log 2>&1 || { tail -5 gpurun_out/r4last/fuzz_kernels.log; exit 1; }
  tail -3 gpurun_out/r4last/fuzz_kernels.log
else
  timeout -k 10 300 python tools/dev_gemm_race.py > gpurun_out/r4last/race_gemm256.log 2>&1 || { tail -5 gpurun_out/r4last/race_gemm256.log; exit 1; }
  tail -1 gpurun_out/r4last/race_gemm256.log
  PI3_GEMM_4W=2 timeout -k 10 300 python tools/dev_gemm_race.py > gpurun_out/r4last/race_gemm4w_ilv.log 2>&1 || { tail -5 gpurun_out/r4last/race_gemm4w_ilv.log; exit 1; }
  tail -1 gpurun_out/r4last/race_gemm4w_ilv.log
  PI3_GEMM_4W=1 timeout -k 10 300 python tools/dev_gemm_race.py > gpurun_out/r4last/race_gemm4w.log 2>&1 || { tail -5 gpurun_out/r4last/race_gemm4w.log; exit 1; }
  tail -1 gpurun_out/r4last/race_gemm4w.log
  timeout -k 10 300 python tools/dev_qkv_race.py > gpurun_out/r4last/race_qkv.log 2>&1 || { tail -5 gpurun_out/r4last/race_qkv.log; exit 1; }
  tail -1 gpurun_out/r4last/race_qkv.log
fi
