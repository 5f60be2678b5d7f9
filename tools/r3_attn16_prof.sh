#!/bin/bash
# Dev aid (GPU box): the round-3 records of the cross-attention mixin (bench lines + rocprofv3 kernel statistics), written to
# gpurun_out/attn16/ and copied into profiles/ by hand.
set -o pipefail
mkdir -p gpurun_out/attn16
: > gpurun_out/attn16/bench.log
for bw in "" "--backward"; do
  for v in "" "--bf16 --matmul fp32" "--bf16"; do
    timeout -k 10 300 python3 tools/bench_cross_attn.py $bw $v 2>/dev/null >> gpurun_out/attn16/bench.log || exit 1
  done
done
cat gpurun_out/attn16/bench.log | cut -c1-120
bash tools/prof_cmd.sh attn16_fwd "" python3 tools/bench_cross_attn.py --bf16 > gpurun_out/attn16/prof_fwd.txt 2>&1 &&
bash tools/prof_cmd.sh attn16_bwd "" python3 tools/bench_cross_attn.py --backward --bf16 > gpurun_out/attn16/prof_bwd.txt 2>&1
