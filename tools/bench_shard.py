#!/usr/bin/env python3
"""Dev aid: where the time of the fused SUM launch goes at small per-GPU shards (what each GPU runs at 8-way strong scaling).
Times, per token count: the fused launch (ids pulled in-kernel), the same with ids given, the tokens-only launch (NOOP: row
gather + norm + store, no index phase), and a plain device copy of the launch's algorithmic bytes."""
import json, sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
import bench
import mixture_of_tokenizers_amd as mot
from mixture_of_tokenizers_amd import data_creation as dc

dev = torch.device("cuda", 0)
res = {}
only = [int(a) for a in sys.argv[1:]] or (32, 64, 128, 256)
for rows in only:
    inp = bench.make_inputs("c4", dev, 12345, False, rows=rows)
    T, D, bpt = 2048, 768, 16
    toks, tab = torch.from_numpy(inp["toks"]).to(dev), torch.from_numpy(inp["tab"]).to(dev)
    out = torch.empty((rows, T, D), device=dev)
    ids = dc.pull_from_left(dc.tokens_to_bytes(toks, tab), bpt, 456, 457)
    plans = {"fused": mot.embed_mix_plan(toks, inp["tok_table"], inp["byte_table"], mode="sum", bpt=bpt, ttb=tab, pull="left", norm_out=True, out=out),
             "given": mot.embed_mix_plan(toks, inp["tok_table"], inp["byte_table"], mode="sum", bpt=bpt, ids_a=ids, norm_out=True, out=out),
             "noop": mot.embed_mix_plan(toks, inp["tok_table"], mode="noop", norm_tok=True, out=out)}
    n = rows * T
    r = {}
    for k, p in plans.items():
        ms = bench.timed_launches(p, 200, warm=20)
        r[k + "_us"] = ms * 1e3
        r[k + "_frac"] = (6180 if k != "noop" else 6148) * n / (ms * 1e-3) / 8e12
    src = torch.empty(n * 6180 // 8, dtype=torch.uint8, device=dev); dst = torch.empty_like(src)
    ms = bench.timed_launches(lambda: dst.copy_(src), 200, warm=20)
    r["copy_same_bytes_us"] = ms * 1e3
    res[n] = r
    print(n, json.dumps(r), flush=True)
