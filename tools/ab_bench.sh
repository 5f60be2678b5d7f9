#!/bin/bash
# Dev aid (GPU box): bench.py kernel time for every named build/variants/<name>.so, one process each.
# usage: tools/ab_bench.sh "<bench.py args>" name ...
args=$1; shift
for name in "$@"; do
    MOT_DEV=1 MOT_DEV_LIB=$PWD/build/variants/$name.so timeout -k 10 200 python3 bench.py $args --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('$name', 'kernel_ms %.4f  ms_per_step %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step']))
"
done
