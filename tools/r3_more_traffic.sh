#!/bin/bash
# Dev aid (GPU box): PMC traffic passes for the workloads that had no counter record (bf16 tables, uniform ids, config 3), added to
# profiles/traffic.json by profiles/summarize.py; results gathered under gpurun_out/traffic2/.
set -o pipefail
out=gpurun_out/traffic2
mkdir -p $out
bash tools/pmc_traffic.sh r03_c4_fused_bf16 c4_fused_bf16 --workload c4 --dtype bf16 > $out/c4b.log 2>&1 &&
bash tools/pmc_traffic.sh r03_c5_mean_bf16 c5_fused_bf16 --workload c5 --dtype bf16 > $out/c5b.log 2>&1 &&
bash tools/pmc_traffic.sh r03_c4_fused_uniform c4_fused_uniform --workload c4 --uniform-ids > $out/c4u.log 2>&1 &&
bash tools/pmc_traffic.sh r03_c3_fused c3_fused --workload c3 > $out/c3.log 2>&1 || { tail -3 $out/*.log; exit 1; }
cp profiles/traffic.json profiles/r03_c4_fused_bf16_pmc.json profiles/r03_c4_fused_bf16_kernel_stats.csv profiles/r03_c5_mean_bf16_pmc.json \
   profiles/r03_c5_mean_bf16_kernel_stats.csv profiles/r03_c4_fused_uniform_pmc.json profiles/r03_c4_fused_uniform_kernel_stats.csv \
   profiles/r03_c3_fused_pmc.json profiles/r03_c3_fused_kernel_stats.csv $out/
cat $out/*.log | tail -8
