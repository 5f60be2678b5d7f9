#!/bin/bash
# Dev aid (GPU box): parity of the bf16 gather-GEMM paths, then whole-call time of the working tree's kernel against the round-2
# kernel (build/variants/c16_r2.so = the library at the last commit that still had it), interleaved, one process each.
set -o pipefail
mkdir -p gpurun_out/c16
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 500 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "concat or gather" > gpurun_out/c16/tests.log 2>&1; rc=$?
tail -5 gpurun_out/c16/tests.log
[ $rc -ne 0 ] && exit $rc
fi
for rep in 1 2; do
for lib in "" ${AB_LIBS:-c16_r2}; do
    if [ -n "$lib" ]; then export MOT_DEV=1 MOT_DEV_LIB=$PWD/build/variants/$lib.so; else unset MOT_DEV MOT_DEV_LIB; fi
    for wl in ${AB_WL:-c2l}; do
    timeout -k 10 200 python3 bench.py --workload $wl --dtype bf16 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('${lib:-tree} $wl', 'kernel_ms %.4f  ms_per_step %.4f frac %.3f' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac']))
" | tee -a gpurun_out/c16/ab.log
    done
done
done
