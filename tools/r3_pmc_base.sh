#!/bin/bash
# round 3, first GPU call: counter records of the kernels VERDICT r2 names (backward of config 4, config 5 mean kernel)
set -uo pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3a
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
b4=(python3 "$root/bench.py" --steps 20 --warmup 4 --no-cpu-baseline --no-extra --workload c4 --backward)
c5=(python3 "$root/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --no-extra --workload c5)
run() { tag=$1; shift; pmc=$1; shift; if [ -z "$pmc" ]; then rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$tag" -- "$@" > "$out/$tag.json" 2> "$out/$tag.err"; else rocprofv3 --pmc $pmc --output-format csv -d "$out/$tag" -- "$@" > "$out/$tag.json" 2> "$out/$tag.err"; fi; echo "$tag rc=$?"; }
run b4_kt "" "${b4[@]}"
run b4_fetch "FETCH_SIZE" "${b4[@]}"
run b4_write "WRITE_SIZE" "${b4[@]}"
run b4_sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "${b4[@]}"
run b4_tcc "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "${b4[@]}"
run c5_kt "" "${c5[@]}"
run c5_fetch "FETCH_SIZE" "${c5[@]}"
run c5_write "WRITE_SIZE" "${c5[@]}"
run c5_tcc "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_HIT_sum TCC_MISS_sum" "${c5[@]}"
run c5_sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR" "${c5[@]}"
cd "$root"
for t in b4_fetch b4_write b4_sq b4_tcc c5_fetch c5_write c5_tcc c5_sq; do echo "== $t"; python3 tools/pmc_summary.py "$out/$t" 2>&1 | grep -v "zero_i32\|rows_rnorm" ; done > "$out/summary.txt"
for t in b4_kt c5_kt; do echo "== $t"; f=$(find "$out/$t" -name "*_kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"; done >> "$out/summary.txt"
python3 bench.py --workload c4 --backward --no-cpu-baseline --no-extra > "$out/bench_c4_bwd.json" 2> "$out/bench_c4_bwd.err"
python3 bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline --no-extra > "$out/bench_c5.json" 2> "$out/bench_c5.err"
tail -c 1500 "$out/summary.txt"
