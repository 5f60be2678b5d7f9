set -o pipefail
out=gpurun_out/final2; mkdir -p $out
timeout -k 10 300 python3 bench.py --dtype bf16 --backward --no-cpu-baseline --no-extra > $out/r03_c4_fused_bf16_bench.json 2>/dev/null || exit 1
timeout -k 10 300 python3 bench.py --workload c5 --dtype bf16 --no-cpu-baseline --no-extra > $out/r03_c5_mean_bf16_bench.json 2>/dev/null || exit 1
timeout -k 10 300 python3 bench.py --workload c3 --no-cpu-baseline --no-extra > $out/r03_c3_fused_bench.json 2>/dev/null || exit 1
timeout -k 10 300 python3 bench.py --uniform-ids --no-cpu-baseline --no-extra > $out/r03_c4_fused_uniform_bench.json 2>/dev/null || exit 1
