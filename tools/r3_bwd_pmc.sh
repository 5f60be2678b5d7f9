#!/bin/bash
# round 3: counter record of the backward of config 4 (embed_mix_bwd_plain_kernel + the token-order kernels) -> profiles/r03_c4_backward_pmc.json
set -uo pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3_bwd_pmc
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
b4=(python3 "$root/bench.py" --steps 20 --warmup 4 --no-cpu-baseline --no-extra --workload c4 --backward)
run() { tag=$1; shift; pmc=$1; shift; if [ -z "$pmc" ]; then rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$tag" -- "$@" > "$out/$tag.json" 2> "$out/$tag.err"; else rocprofv3 --pmc $pmc --output-format csv -d "$out/$tag" -- "$@" > "$out/$tag.json" 2> "$out/$tag.err"; fi; echo "$tag rc=$?"; }
run kt "" "${b4[@]}"
run fetch "FETCH_SIZE" "${b4[@]}"
run write "WRITE_SIZE" "${b4[@]}"
run sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "${b4[@]}"
run tcc "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "${b4[@]}"
cd "$root"
python3 - "$out" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
res = {"command": "bench.py --workload c4 --backward (256 x 2048 tokens, d 768, bpt 16, norm_out; fp32)", "kernels": {}}
def pmc(d):
    fn = glob.glob(f"{out}/{d}/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        agg[(r["Kernel_Name"].split("(")[0][:80], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
allc = {}
for d in ("fetch", "write", "sq", "tcc"):
    allc.update(pmc(d))
stats = glob.glob(f"{out}/kt/**/*_kernel_stats.csv", recursive=True)[0]
dur = {r["Name"].split("(")[0][:80]: (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(open(stats))}
for (k, c), v in sorted(allc.items()):
    if "mot::" in k and ("bwd" in k):
        res["kernels"].setdefault(k, {})[c] = v
for k in res["kernels"]:
    if k in dur:
        res["kernels"][k]["avg_us_kernel_trace"], res["kernels"][k]["calls"] = dur[k]
N, D, bpt = 256 * 2048, 768, 16
main = next(k for k in res["kernels"] if "plain" in k or "lc_kernel" in k)
m = res["kernels"][main]
fetch_b, write_b, atom_b = 2 * m["FETCH_SIZE"] * 1024, m["WRITE_SIZE"] * 1024, m["TCC_EA0_ATOMIC_sum"] * 64
alg = {"grad_out_rows": N * D * 4, "byte_ids_int64": N * bpt * 8, "sorted_pairs": N * 8}
res["traffic"] = {"kernel": main, "fetch_bytes_x2_corrected": fetch_b, "write_bytes": write_b, "atomic_request_bytes_64B_each": atom_b,
                  "algorithmic_read_bytes_without_token_rows": sum(alg.values()), "algorithmic_detail": alg,
                  "fetch_over_algorithmic_read": fetch_b / sum(alg.values()),
                  "note": "FETCH_SIZE x 2 (gfx950 counts a 16 B/lane stream at half, MI355X_MICROARCH.md); the remainder over the "
                          "algorithmic read is one 3 KB token row per run of equal tokens (~53 000 runs: 0.16 GB) and the rows read again "
                          "by the atomic / slow paths; WRITE + atomic requests = the d_tok row-adds (one 3 KB row per run) + the byte table"}
wc = m["SQ_WAVE_CYCLES"]
res["wave_cycle_shares"] = {"wait_any": m["SQ_WAIT_ANY"] / wc, "wait_inst_any": m["SQ_WAIT_INST_ANY"] / wc, "active_inst_any": m["SQ_ACTIVE_INST_ANY"] / wc,
                            "valu_insts_per_position": m["SQ_INSTS_VALU"] / N, "lds_insts_per_position": m["SQ_INSTS_LDS"] / N,
                            "lds_bank_conflict_over_idx_active": m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]}
json.dump(res, open("profiles/r03_c4_backward_pmc.json", "w"), indent=1)
print(json.dumps(res["traffic"], indent=1)); print(json.dumps(res["wave_cycle_shares"], indent=1))
PY
cp profiles/r03_c4_backward_pmc.json gpurun_out/
f=$(find "$out/kt" -name "*_kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-160 > gpurun_out/r03_c4_backward_kernel_stats.txt; cat gpurun_out/r03_c4_backward_kernel_stats.txt
