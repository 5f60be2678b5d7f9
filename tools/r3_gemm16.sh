#!/bin/bash
# Dev aid (GPU box): parity of everything that runs through the bf16 product kernels, then the character mixer, the cross-attention
# mixin and the composed bf16 concat with the 256 x 256 kernel against the 128 x 128 one (build/variants/g16_dev.so is a
# -DMOT_DEV_ABLATION build: MOT_GEMM16_OLD=1 selects the old kernel).
set -o pipefail
mkdir -p gpurun_out/gemm16
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 700 python -m pytest tests/test_gpu_attn.py tests/test_gpu_swa.py tests/test_gpu_bf16.py -x -q > gpurun_out/gemm16/tests.log 2>&1; rc=$?
tail -4 gpurun_out/gemm16/tests.log
[ $rc -ne 0 ] && exit $rc
fi
export MOT_DEV=1 MOT_DEV_LIB=$PWD/build/variants/g16_dev.so
for old in "" 1; do
    if [ -n "$old" ]; then export MOT_GEMM16_OLD=1; else unset MOT_GEMM16_OLD; fi
    echo "== old=$old" | tee -a gpurun_out/gemm16/ab.log
    timeout -k 10 200 python3 tools/bench_swa.py 8 8192 bf16 2>/dev/null | cut -c1-100 | tee -a gpurun_out/gemm16/ab.log
    timeout -k 10 200 python3 tools/bench_cross_attn.py --bf16 2>/dev/null | cut -c1-200 | tee -a gpurun_out/gemm16/ab.log
    timeout -k 10 200 python3 tools/bench_cross_attn.py --bf16 --backward 2>/dev/null | cut -c1-200 | tee -a gpurun_out/gemm16/ab.log
    timeout -k 10 200 python3 tools/bench_cross_attn.py --bf16 --backward --dual 2>/dev/null | cut -c1-200 | tee -a gpurun_out/gemm16/ab.log
done
