#!/bin/bash
# dev: timing ablations of embed_mix_bwd_plain_kernel (MOT_BWD_ABL bits, see the kernel), one bench.py --backward run each
out=gpurun_out/r3c; mkdir -p $out
for abl in "$@"; do
  MOT_DEV=1 MOT_DEV_LIB=$PWD/build/variants/bwd_dev.so MOT_BWD_ABL=$abl timeout -k 10 300 python3 bench.py --workload c4 --backward --steps 80 --warmup 10 --no-cpu-baseline --no-extra 2> $out/bench_abl$abl.err | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('abl $abl backward_ms %.4f' % (d['backward']['kernel_ms']))
"
done
