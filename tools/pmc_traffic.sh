#!/bin/bash
# Measures what profiles/traffic.json holds for one bench.py workload: the rocprofv3 kernel trace (per-kernel average duration)
# and the HBM bytes per launch from the TCC counters, FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit
# one pass on gfx950; MI355X_MICROARCH.md, rocprofv3 PMC slots), then condenses them with profiles/summarize.py.
# Run on the GPU box from the repo root:  tools/pmc_traffic.sh <tag> <key> [bench.py args...]
#   e.g. tools/pmc_traffic.sh r02_c4_fused c4_fused --workload c4
set -euo pipefail
tag=$1; key=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
common=(python3 "$root/bench.py" --steps 20 --warmup 4 --no-cpu-baseline --no-extra "$@")
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- "${common[@]}" > "$out/kt.json" 2> "$out/kt.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- "${common[@]}" > "$out/fetch.json" 2> "$out/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- "${common[@]}" > "$out/write.json" 2> "$out/write.err"
cd "$root"
python3 profiles/summarize.py "$tag" "$out/kt" "$out/fetch" "$out/write" --key "$key"
