#!/bin/bash
# round 3: backward tests on the new plain kernel + A/B of pipeline depth (dev build: MOT_BWD_ABL 0 = two rows ahead, 32 = one row ahead, 16 = lc kernel)
set -uo pipefail
out=gpurun_out/r3b; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_backward.py -x -q > $out/pytest_bwd.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_bwd.log
for abl in 0 32 16; do
  MOT_DEV=1 MOT_DEV_LIB=$PWD/build/variants/bwd_dev.so MOT_BWD_ABL=$abl timeout -k 10 300 python3 bench.py --workload c4 --backward --steps 200 --warmup 20 --no-cpu-baseline --no-extra 2> $out/bench_abl$abl.err | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('abl $abl backward_ms %.4f fwd_ms %.4f' % (d['backward']['kernel_ms'], d['roofline']['kernel_ms']))
"
done
