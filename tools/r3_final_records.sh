#!/bin/bash
# Dev aid (GPU box): the records of the fused front-end on the final build of the round -- the three PMC traffic figures
# (profiles/traffic.json is keyed to the kernel sources) and the bench lines -- gathered under gpurun_out/final/ for copying into profiles/.
set -o pipefail
out=gpurun_out/final
mkdir -p $out
bash tools/pmc_traffic.sh r03_c4_fused c4_fused --workload c4 > $out/pmc_c4.log 2>&1 &&
bash tools/pmc_traffic.sh r03_c2_fused c2_fused --workload c2 > $out/pmc_c2.log 2>&1 &&
bash tools/pmc_traffic.sh r03_c5_mean c5_fused --workload c5 > $out/pmc_c5.log 2>&1 || exit 1
cp profiles/traffic.json profiles/r03_c4_fused_pmc.json profiles/r03_c4_fused_kernel_stats.csv profiles/r03_c2_fused_pmc.json profiles/r03_c2_fused_kernel_stats.csv \
   profiles/r03_c5_mean_pmc.json profiles/r03_c5_mean_kernel_stats.csv $out/
timeout -k 10 600 python3 bench.py > $out/r03_c4_fused_bench.json 2> $out/c4.err || exit 1
timeout -k 10 300 python3 bench.py --dtype bf16 --backward --no-cpu-baseline --no-extra > $out/r03_c4_fused_bf16_bench.json 2> $out/c4b.err || exit 1
timeout -k 10 300 python3 bench.py --workload c2 --no-cpu-baseline --no-extra > $out/r03_c2_fused_bench.json 2> $out/c2.err || exit 1
timeout -k 10 300 python3 bench.py --workload c5 --no-cpu-baseline --no-extra > $out/r03_c5_mean_bench.json 2> $out/c5.err || exit 1
timeout -k 10 300 python3 bench.py --workload c5 --dtype bf16 --no-cpu-baseline --no-extra > $out/r03_c5_mean_bf16_bench.json 2> $out/c5b.err || exit 1
tail -c 600 $out/r03_c4_fused_bench.json
