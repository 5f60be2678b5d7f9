#!/bin/bash
# Runs GPU steps one after another on the gpurun box, each under its own `timeout -k 10`, logging into gpurun_out/<tag>/.
# A step that TIMES OUT or is KILLED ends the sequence (a hung kernel must not be followed by more GPU work); a step that
# merely fails (test assertion, non-zero exit) is recorded and the next one still runs.
# usage: tools/gpu_steps.sh <tag> "<seconds> <name> <command...>" ...
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
for spec in "$@"; do
    secs=${spec%% *}; rest=${spec#* }; name=${rest%% *}; cmd=${rest#* }
    echo "=== $name (limit ${secs}s): $cmd"
    timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.log" 2> "$out/$name.err"
    rc=$?
    echo "    rc=$rc"; echo "rc=$rc" >> "$out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "    $name timed out / was killed: stopping here"; exit $rc; fi
done
exit 0
