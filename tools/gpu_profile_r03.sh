#!/bin/bash
# round-3 profiles: rocprofv3 kernel stats of the bench command, PMC passes for the GEMMs (fused qkv epilogue included)
# and for the global attention (HBM traffic).  Summaries are copied into profiles/ by hand afterwards.
set -o pipefail
mkdir -p gpurun_out/r3p
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r3p/prof --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3p/prof_bench.json 2> gpurun_out/r3p/prof_bench.err || echo "kernel-trace run failed"
find gpurun_out/r3p/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3p/bench_kernel_stats.csv \;
head -12 gpurun_out/r3p/bench_kernel_stats.csv | cut -c1-150
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/r3p/pmcg_$tag --output-format csv -- python tools/dev_gemm.py > gpurun_out/r3p/pmcg_$tag.log 2>&1 || echo "pmc gemm $tag failed"
done
python tools/pmc_summary.py gemm256 0.2 gpurun_out/r3p/pmcg_* > gpurun_out/r3p/gemm256_pmc.csv 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/r3p/pmca_$tag --output-format csv -- python tools/dev_attn.py > gpurun_out/r3p/pmca_$tag.log 2>&1 || echo "pmc attn $tag failed"
done
python tools/pmc_summary.py attn_fwd64 5.0 gpurun_out/r3p/pmca_* > gpurun_out/r3p/attention_pmc.csv 2>&1
cat gpurun_out/r3p/attention_pmc.csv | cut -c1-200
tail -c 600 gpurun_out/r3p/prof_bench.json
echo
timeout -k 10 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r3p/bench_line.json 2> gpurun_out/r3p/bench_line.err || echo "bench failed"
tail -c 400 gpurun_out/r3p/bench_line.json
find gpurun_out/r3p -name "*kernel_trace.csv" -size +30M -delete
