#!/bin/bash
# Dev aid (GPU box): parity of the bf16 gather-GEMM paths, then kernel time of the sub-tile kernel against the round-2 kernel
# (build/variants/c16_dev.so is a -DMOT_DEV_ABLATION build: MOT_C16_OLD=1 selects the round-2 kernel).
set -o pipefail
mkdir -p gpurun_out/c16
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 500 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "concat or gather" > gpurun_out/c16/tests.log 2>&1; rc=$?
tail -5 gpurun_out/c16/tests.log
[ $rc -ne 0 ] && exit $rc
fi
for rep in 1 2; do
for old in "" 1; do
    if [ -n "$old" ]; then export MOT_C16_OLD=1; else unset MOT_C16_OLD; fi
    MOT_DEV=1 MOT_DEV_LIB=$PWD/build/variants/c16_dev.so timeout -k 10 200 python3 bench.py --workload c2l --dtype bf16 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('old=$old', 'kernel_ms %.4f  ms_per_step %.4f frac %.3f' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac']))
" | tee -a gpurun_out/c16/ab.log
done
done
