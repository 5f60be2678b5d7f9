#!/bin/bash
# round 3: the records kept under profiles/ (run on the GPU box from the repo root)
set -uo pipefail
tools/pmc_traffic.sh r03_c4_fused c4_fused --workload c4
tools/pmc_traffic.sh r03_c2_fused c2_fused --workload c2
tools/pmc_traffic.sh r03_c5_mean c5_fused --workload c5
tools/pmc_traffic.sh r03_c4_fused_bf16 c4_fused_bf16 --workload c4 --dtype bf16
python3 bench.py --workload c4 --backward --no-cpu-baseline > profiles/r03_c4_fused_bench.json 2> gpurun_out/r03_c4_bench.err
python3 bench.py --workload c2 --no-cpu-baseline --no-extra > profiles/r03_c2_fused_bench.json 2>> gpurun_out/r03_c4_bench.err
python3 bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline --no-extra > profiles/r03_c5_mean_bench.json 2>> gpurun_out/r03_c4_bench.err
python3 bench.py --workload c5 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline --no-extra > profiles/r03_c5_mean_bf16_bench.json 2>> gpurun_out/r03_c4_bench.err
python3 bench.py --workload c4 --dtype bf16 --backward --no-cpu-baseline --no-extra > profiles/r03_c4_fused_bf16_bench.json 2>> gpurun_out/r03_c4_bench.err
ls -la profiles/r03_*
