#!/bin/bash
# MFMA-busy and clock of the global attention: three-phase kernel vs the software-pipelined one (PI3_ATTN_PIPE=1)
mkdir -p gpurun_out/pp
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pipe in 0 1; do
  PI3_ATTN_PIPE=$pipe timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $R/gpurun_out/pp/pmc_$pipe --output-format csv -- python3 $R/tools/dev_attn.py > $R/gpurun_out/pp/p$pipe.log 2>&1
  grep "S=64300" $R/gpurun_out/pp/p$pipe.log
done
cd $R
python3 tools/pmc_summary.py attn_fwd64 5.0 gpurun_out/pp/pmc_0 gpurun_out/pp/pmc_1 | cut -c1-170
find gpurun_out/pp -name "*kernel_trace.csv" -delete
