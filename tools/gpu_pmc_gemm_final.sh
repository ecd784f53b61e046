#!/bin/bash
# end of round 3: PMC pass over the block GEMMs (tools/dev_gemm.py) after the two-phase K loop
mkdir -p gpurun_out/r3g
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/r3g/pmcg_$tag --output-format csv -- python tools/dev_gemm.py > gpurun_out/r3g/pmcg_$tag.log 2>&1 || echo "pmc gemm $tag failed"
done
python tools/pmc_summary.py gemm256 0.2 gpurun_out/r3g/pmcg_* > gpurun_out/r3g/gemm256_pmc.csv 2>&1
grep "MFMA_BUSY\|GRBM_GUI\|WAIT" gpurun_out/r3g/gemm256_pmc.csv | cut -c1-170
find gpurun_out/r3g -name "*kernel_trace.csv" -size +30M -delete
