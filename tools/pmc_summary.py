#!/usr/bin/env python3
"""Dev aid: mean of each PMC counter per kernel from a rocprofv3 --pmc CSV directory (kernel names cut to 70 chars)."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    fn = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"][:70]
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for (k, c), v in sorted(agg.items()):
        if k.startswith("void mot::"):
            print(f"{k:70s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}  dur_us={sum(dur[k])/len(dur[k])/1e3:.1f}")
