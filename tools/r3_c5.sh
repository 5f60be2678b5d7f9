#!/bin/bash
# dev: config-5 kernel variants (build/variants/<name>.so): bench.py c5 fp32 + bf16, and tools/c5_floor.py (random / sequential / one token)
out=gpurun_out/r3e; mkdir -p $out
for name in "$@"; do
  for dt in f32 bf16; do
    MOT_DEV=1 MOT_DEV_LIB=$PWD/build/variants/$name.so timeout -k 10 300 python3 bench.py --workload c5 --dtype $dt --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2> $out/$name.$dt.err | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('$name $dt kernel_ms %.3f frac %.3f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))
"
  done
  MOT_DEV=1 MOT_DEV_LIB=$PWD/build/variants/$name.so timeout -k 10 300 python3 tools/c5_floor.py 2> $out/$name.floor.err | sed "s/^/$name floor: /"
done
