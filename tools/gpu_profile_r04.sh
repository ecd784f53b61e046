#!/bin/bash
# round-4 profiles: rocprofv3 kernel stats of the bench command, PMC passes for the block GEMMs (fused qkv epilogue and
# the new GELU included) and for the global attention (HBM traffic: FETCH_SIZE / WRITE_SIZE in separate passes).
# Summaries are copied into profiles/r04_* afterwards.  Steps are chained with && : a killed step ends the session.
set -o pipefail
mkdir -p gpurun_out/r4p
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r4p/prof --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r4p/prof_bench.json 2> gpurun_out/r4p/prof_bench.err || { echo "kernel-trace run failed"; exit 1; }
find gpurun_out/r4p/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r4p/bench_kernel_stats.csv \;
head -12 gpurun_out/r4p/bench_kernel_stats.csv | cut -c1-150
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/r4p/pmcg_$tag --output-format csv -- python tools/dev_gemm.py > gpurun_out/r4p/pmcg_$tag.log 2>&1 || { echo "pmc gemm $tag failed"; exit 1; }
  echo "pmc gemm $tag done"
done
python tools/pmc_summary.py gemm256 0.15 gpurun_out/r4p/pmcg_* > gpurun_out/r4p/gemm256_pmc.csv 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/r4p/pmca_$tag --output-format csv -- python tools/dev_attn.py > gpurun_out/r4p/pmca_$tag.log 2>&1 || { echo "pmc attn $tag failed"; exit 1; }
  echo "pmc attn $tag done"
done
python tools/pmc_summary.py attn_fwd64 5.0 gpurun_out/r4p/pmca_* > gpurun_out/r4p/attention_pmc.csv 2>&1
cat gpurun_out/r4p/attention_pmc.csv | cut -c1-200
tail -c 500 gpurun_out/r4p/prof_bench.json
find gpurun_out/r4p -name "*kernel_trace.csv" -size +20M -delete
find gpurun_out/r4p -name "*counter_collection.csv" -size +20M -delete
