set -o pipefail
out=gpurun_out/records; mkdir -p $out; : > $out/char_swa_bench.json
for v in fp32 bf16 bf16-fp32mm; do timeout -k 10 300 python3 tools/bench_swa.py 8 8192 $v 2>/dev/null >> $out/char_swa_bench.json || exit 1; done
timeout -k 10 300 python3 tools/bench_swa.py 8 8192 bf16 kv-cache 2>/dev/null >> $out/char_swa_bench.json || exit 1
timeout -k 10 300 python3 tools/bench_swa.py 8 8192 fp32 kv-cache 2>/dev/null >> $out/char_swa_bench.json || exit 1
bash tools/prof_cmd.sh rec_swa16 "" python3 tools/bench_swa.py 8 8192 bf16 > $out/char_swa_bf16_kernel_stats.txt 2>&1
