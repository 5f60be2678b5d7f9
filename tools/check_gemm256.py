#!/usr/bin/env python3
"""Dev aid (GPU box): the 256 x 256 product kernel against the numpy oracle of the character mixer at several widths
(dim = heads * 64 = the products' Nc and R), fp32 and bf16 tables, rows not a multiple of 256."""
import sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
from oracle import oracle as orc
import mixture_of_tokenizers_amd as mot
dev = torch.device("cuda", 0)
for d, H in ((256, 4), (512, 8), (768, 12)):
    for bf in (False, True):
        rs = np.random.RandomState(d + bf)
        B, T, cv, Vt, Vc, hd = 2, 330, 8, 300, 132, 64
        f = lambda *s: rs.standard_normal(s).astype(np.float32)
        c = dict(toks=rs.randint(0, Vt, (B, T)).astype(np.int32), cid=rs.randint(0, Vc, (B, T, cv)).astype(np.int64), Et=f(Vt, d), Ec=f(Vc, d),
                 wa=1 + 0.1 * f(d), wc=1 + 0.1 * f(d), wq=f(H * hd, d) / d ** 0.5, wk=f(H * hd, d) / d ** 0.5, wv=f(H * hd, d) / d ** 0.5, wo=f(d, H * hd) / d ** 0.5)
        if bf:
            c = {k: (orc.bf16_round(v) if v.dtype == np.float32 else v) for k, v in c.items()}
        ref = orc.char_swa(c["toks"], c["cid"], c["Et"], c["Ec"], c["wa"], c["wc"], c["wq"], c["wk"], c["wv"], c["wo"], n_heads=H, head_dim=hd, window=8,
                           version="no_residual", round_token_products_bf16=bf)
        t = lambda a: torch.from_numpy(a).to(dev)
        tt = (lambda a: t(a).bfloat16()) if bf else t
        x = mot.functional.char_swa(t(c["toks"]), t(c["cid"]), tt(c["Et"]), tt(c["Ec"]), attn_norm_w=tt(c["wa"]), char_norm_w=tt(c["wc"]), wq=tt(c["wq"]),
                                    wk=tt(c["wk"]), wv=tt(c["wv"]), wo=tt(c["wo"]), n_heads=H, head_dim=hd, window=8, version="no_residual")
        got = x.float().cpu().numpy().astype(np.float64)
        err = np.abs(got - ref).max() / np.abs(ref).max()
        print(f"d={d} bf16={bf}: max rel err {err:.3e}", flush=True)
