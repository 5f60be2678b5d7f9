#!/bin/bash
# Dev aid (GPU box): bench.py with every named build/variants/<name>.so.  usage: tools/ab_bench.sh "<bench args>" name...
args=$1; shift
for name in "$@"; do
    MOT_DEV_LIB=$PWD/build/variants/$name.so timeout -k 10 300 python bench.py $args --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    try:
        j = json.loads(ln); r = j['roofline']; print('%-10s kernel_ms %.4f  frac %.3f' % ('$name', r['kernel_ms'], r['frac']))
    except Exception: pass
"
done
