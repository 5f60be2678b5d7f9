#!/bin/bash
# Dev aid (GPU box): the round-3 records of the two attention mixins on the final build -- bench lines, rocprofv3 kernel
# statistics and the counter passes -- written to gpurun_out/records/, copied into profiles/ by hand.
set -o pipefail
out=gpurun_out/records
mkdir -p $out
: > $out/cross_attn_bench.json; : > $out/cross_attn_two_ids_bench.json; : > $out/char_swa_bench.json
for bw in "" "--backward"; do
  for v in "" "--bf16 --matmul fp32" "--bf16"; do
    timeout -k 10 300 python3 tools/bench_cross_attn.py $bw $v 2>/dev/null >> $out/cross_attn_bench.json || exit 1
  done
done
timeout -k 10 300 python3 tools/bench_cross_attn.py --bf16 --kv-cache 2>/dev/null >> $out/cross_attn_bench.json || exit 1
for bw in "" "--backward"; do
  for v in "" "--bf16"; do
    timeout -k 10 300 python3 tools/bench_cross_attn.py --dual $bw $v 2>/dev/null >> $out/cross_attn_two_ids_bench.json || exit 1
  done
done
for v in fp32 bf16 bf16-fp32mm; do
  timeout -k 10 300 python3 tools/bench_swa.py 8 8192 $v 2>/dev/null >> $out/char_swa_bench.json || exit 1
done
timeout -k 10 300 python3 tools/bench_swa.py 8 8192 bf16 kv-cache 2>/dev/null >> $out/char_swa_bench.json || exit 1
timeout -k 10 300 python3 tools/bench_swa.py 8 8192 fp32 kv-cache 2>/dev/null >> $out/char_swa_bench.json || exit 1
bash tools/prof_cmd.sh rec_fwd16 "" python3 tools/bench_cross_attn.py --bf16 > $out/cross_attn_bf16_fwd_kernel_stats.txt 2>&1 &&
bash tools/prof_cmd.sh rec_bwd16 "" python3 tools/bench_cross_attn.py --backward --bf16 > $out/cross_attn_bf16_fwd_bwd_kernel_stats.txt 2>&1 &&
bash tools/prof_cmd.sh rec_bwd32 "" python3 tools/bench_cross_attn.py --backward > $out/cross_attn_fp32_fwd_bwd_kernel_stats.txt 2>&1 &&
bash tools/prof_cmd.sh rec_dual16 "" python3 tools/bench_cross_attn.py --dual --backward --bf16 > $out/cross_attn_two_ids_bf16_kernel_stats.txt 2>&1 &&
bash tools/prof_cmd.sh rec_swa16 "" python3 tools/bench_swa.py 8 8192 bf16 > $out/char_swa_bf16_kernel_stats.txt 2>&1 &&
bash tools/r3_attn_pmc.sh --backward > $out/cross_attn_pmc.txt 2>&1
