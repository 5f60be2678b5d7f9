#!/bin/bash
# Dev aid (GPU box): rocprofv3 kernel trace (+ optional PMC passes) of any python command of this repo.
# usage: tools/prof_cmd.sh <tag> "<pmc counters or empty>" python3 <script> [args]      (results under gpurun_out/prof_<tag>/)
set -uo pipefail
tag=$1; pmc=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
prog=$1; shift
args=()
for a in "$@"; do case "$a" in /*|-*) args+=("$a");; *) if [ -e "$root/$a" ]; then args+=("$root/$a"); else args+=("$a"); fi;; esac; done
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- "$prog" "${args[@]}" > "$out/kt.out" 2> "$out/kt.err"
if [ -n "$pmc" ]; then
    rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc" -- "$prog" "${args[@]}" > "$out/pmc.out" 2> "$out/pmc.err"
fi
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for fn in glob.glob(out + "/kt/**/*_kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(fn)))[:14]:
        print("%-90s calls %5s avg %10.1f us  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
for fn in glob.glob(out + "/pmc/**/*_counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        if "mot::" in k: print("%-60s %-28s n=%4d mean=%.5g" % (k, c, len(v), sum(v) / len(v)))
PY
