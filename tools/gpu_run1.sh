#!/bin/bash
# round-2 GPU check #1: full GPU test suite, GEMM baseline timing + PMC passes for gemm256 (evidence the judge asked for)
set -o pipefail
mkdir -p gpurun_out/r2a
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2a/tests.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r2a/tests.log
tail -5 gpurun_out/r2a/tests.log
timeout -k 10 120 python tools/dev_gemm.py > gpurun_out/r2a/gemm_base.log 2>&1; tail -6 gpurun_out/r2a/gemm_base.log
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/r2a/pmc_$tag --output-format csv -- python tools/dev_gemm.py > gpurun_out/r2a/pmc_$tag.log 2>&1 || echo "pmc $tag failed"
done
python tools/pmc_summary.py gemm256 0.05 gpurun_out/r2a/pmc_* > gpurun_out/r2a/gemm256_pmc_base.csv 2>&1
cat gpurun_out/r2a/gemm256_pmc_base.csv | cut -c1-220
