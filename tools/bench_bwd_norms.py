#!/usr/bin/env python3
"""Dev aid (GPU box): mot_embed_mix_bwd of the headline batch (256 x 2048 tokens, SUM) with the output norm only and with all three
norms, fp32 and bf16 tables, token order given.  One line per case."""
import json, sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
import golden_inputs as gi
import mixture_of_tokenizers_amd as mot
from mixture_of_tokenizers_amd import data_creation as dc
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 2048)
dev = torch.device("cuda", 0)
V, Vb, bpt, D, Db = 50257, 458, 16, 768, 48
g = torch.Generator(device=dev).manual_seed(3)
toks = torch.from_numpy(gi.fineweb_like_tokens(11, B, T, vocab=V)).to(dev)
tab = torch.from_numpy(gi.widen_left_pad(gi.load_real_ttb8(), bpt).astype(np.int32)).to(dev)
ids = dc.pull_from_left(dc.tokens_to_bytes(toks, tab), bpt, 456, 457)
for dtype in (torch.float32, torch.bfloat16):
    Et, Eb = torch.randn((V, D), generator=g, device=dev).to(dtype), torch.randn((Vb, Db), generator=g, device=dev).to(dtype)
    go = torch.randn((B, T, D), generator=g, device=dev).to(dtype)
    order = mot.functional.token_order(toks, V)
    for name, kw in (("none", {}), ("norm_out", dict(norm_out=True)), ("norm_tok+byte", dict(norm_tok=True, norm_byte=True)),
                     ("all three", dict(norm_tok=True, norm_byte=True, norm_out=True))):
        into = {"tok_table": torch.zeros((V, D), device=dev), "byte_table": torch.zeros((Vb, Db), device=dev)}
        fkw = dict(mode="sum", bpt=bpt, ids_a=ids, **kw)
        out = mot.functional.embed_mix(toks, Et, Eb, **fkw) if kw.get("norm_out") else None
        run = lambda: mot.functional.embed_mix_backward(go, toks, Et, Eb, into=into, token_order=order, out=out, **fkw)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        print(json.dumps({"dtype": str(dtype).split(".")[1], "norms": name, "tokens": B * T, "ms": round(e0.elapsed_time(e1) / 20, 4)}), flush=True)
