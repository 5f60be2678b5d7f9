#!/bin/bash
# Dev aid (GPU box): the full GPU suite, then the fp32 paths that run through launch_gemm_rows with the 256 x 256 kernel against the
# 128 x 128 one (build/variants/g16_dev.so is a -DMOT_DEV_ABLATION build: MOT_GEMM32_OLD=1 selects the old kernel).
set -o pipefail
mkdir -p gpurun_out/gemm32
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gemm32/tests.log 2>&1; rc=$?
tail -4 gpurun_out/gemm32/tests.log
[ $rc -ne 0 ] && exit $rc
fi
export MOT_DEV=1 MOT_DEV_LIB=$PWD/build/variants/g16_dev.so
for old in "" 1; do
    if [ -n "$old" ]; then export MOT_GEMM32_OLD=1; else unset MOT_GEMM32_OLD; fi
    echo "== old=$old" | tee -a gpurun_out/gemm32/ab.log
    timeout -k 10 200 python3 bench.py --workload c2l --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('c2l fp32 ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))
" | tee -a gpurun_out/gemm32/ab.log
    timeout -k 10 200 python3 tools/bench_swa.py 8 8192 fp32 2>/dev/null | cut -c1-100 | tee -a gpurun_out/gemm32/ab.log
    timeout -k 10 200 python3 tools/bench_cross_attn.py 2>/dev/null | cut -c1-200 | tee -a gpurun_out/gemm32/ab.log
    timeout -k 10 200 python3 tools/bench_cross_attn.py --backward 2>/dev/null | cut -c1-200 | tee -a gpurun_out/gemm32/ab.log
done
