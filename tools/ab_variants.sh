#!/bin/bash
# Dev aid (GPU box): times tools/bench_shard.py for every build/variants/*.so (and MOT_UNIT values), one process each.
# usage: tools/ab_variants.sh "<rows...>" variant[:unit] ...
rows=$1; shift
for spec in "$@"; do
    name=${spec%%:*}; unit=${spec#*:}; [ "$unit" = "$spec" ] && unit=""
    echo "== $name unit=${unit:-default}"
    MOT_DEV=1 MOT_DEV_LIB=$PWD/build/variants/$name.so MOT_UNIT=$unit timeout -k 10 200 python tools/bench_shard.py $rows 2>&1 | grep -v Warning | python -c "
import sys, json
for ln in sys.stdin:
    try:
        n, j = ln.split(' ', 1); d = json.loads(j)
        print('   %7s fused %7.2f us (%.3f)  given %7.2f (%.3f)  noop %7.2f' % (n, d['fused_us'], d['fused_frac'], d['given_us'], d['given_frac'], d['noop_us']))
    except Exception: print('   ' + ln.rstrip())
"
done
