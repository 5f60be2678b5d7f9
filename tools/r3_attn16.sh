#!/bin/bash
# Dev aid (GPU box): parity of the cross-attention mixin, then its time at T = 65 536 with fp32 tables, with bf16 tables on the
# fp32 MFMA (round 2) and with bf16 tables and the products over the tokens on the bf16 MFMA (round 3), forward and forward + backward.
set -o pipefail
mkdir -p gpurun_out/attn16
timeout -k 10 600 python -m pytest tests/test_gpu_attn.py tests/test_capi_load.py -x -q > gpurun_out/attn16/tests.log 2>&1; rc=$?
tail -5 gpurun_out/attn16/tests.log
[ $rc -ne 0 ] && exit $rc
for bw in "" "--backward"; do
  for v in "" "--bf16 --matmul fp32" "--bf16"; do
    timeout -k 10 300 python3 tools/bench_cross_attn.py $bw $v 2>/dev/null | tee -a gpurun_out/attn16/bench.log
  done
done
