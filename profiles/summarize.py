#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof/...) into the small files kept under profiles/.

usage: python profiles/summarize.py <tag> <kernel-trace dir> [<fetch pmc dir> <write pmc dir>] [--key c4_fused]

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, kernel names cut
to 100 chars), profiles/<tag>_pmc.json (per-kernel mean FETCH_SIZE / WRITE_SIZE in KB as reported),
and updates profiles/traffic.json[key] = HBM bytes per launch of the dominant mot:: kernel:
    2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024
(MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports exactly half the bytes of a 16 B/lane
coalesced read stream; WRITE_SIZE is exact for 16 B/lane streaming stores; separate --pmc passes).
"""
import collections, csv, glob, json, sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    key = "c4_fused"
    if "--key" in sys.argv:
        key = sys.argv[sys.argv.index("--key") + 1]
        args = [a for a in args if a != key]
    tag, kt = args[0], args[1]
    stats = glob.glob(f"{kt}/**/*_kernel_stats.csv", recursive=True)[0]
    rows = list(csv.reader(open(stats)))
    with open(HERE / f"{tag}_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        for r in rows:
            r[0] = r[0][:100]
            w.writerow(r)
    if len(args) >= 4:
        pmc = {}
        for kind, d in (("FETCH_SIZE", args[2]), ("WRITE_SIZE", args[3])):
            fn = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(fn)):
                if r["Counter_Name"] == kind:
                    agg[r["Kernel_Name"][:100]].append(float(r["Counter_Value"]))
            pmc[kind] = {k: {"launches": len(v), "mean_KB": sum(v) / len(v), "min_KB": min(v), "max_KB": max(v)}
                         for k, v in agg.items() if k.startswith(("void mot::", "mot::", "_ZN3mot"))}   # (kernels with __bf16 template arguments come out mangled)
        (HERE / f"{tag}_pmc.json").write_text(json.dumps(pmc, indent=1) + "\n")
        kern = max(pmc["WRITE_SIZE"], key=lambda k: pmc["WRITE_SIZE"][k]["mean_KB"])
        fetch, write = pmc["FETCH_SIZE"][kern]["mean_KB"], pmc["WRITE_SIZE"][kern]["mean_KB"]
        tfile = HERE / "traffic.json"
        t = json.loads(tfile.read_text()) if tfile.exists() else {}
        t[key] = int(2 * fetch * 1024 + write * 1024)
        t[key + "_detail"] = {"kernel": kern, "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
                              "fetch_bytes_corrected_x2": int(2 * fetch * 1024), "write_bytes": int(write * 1024),
                              "profile": f"profiles/{tag}_pmc.json",
                              # bench.py refuses the figure once the kernel's sources differ from what was profiled
                              "source_sha16": __import__("bench").source_sha16()}
        tfile.write_text(json.dumps(t, indent=1) + "\n")
        print(key, t[key])


if __name__ == "__main__":
    main()
