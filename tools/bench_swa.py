#!/usr/bin/env python3
"""Times the character mixer call (sliding-window attention + residuals, mot_char_swa_fwd) at config-5 dims:
Llama-3.2-1B hidden 2048, 32 heads x 64, 8 character slots, window 8, vocab 128 256.  usage: bench_swa.py [B T [fp32 | bf16 | bf16-fp32mm]]
(fp32 tables; bf16 tables with the two token products on the bf16 MFMA; bf16 tables with everything on the fp32 kernels)"""
import json, sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
import bench
import mixture_of_tokenizers_amd as mot

B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 8192)
variant = sys.argv[3] if len(sys.argv) > 3 else "fp32"
cache = {} if (len(sys.argv) > 4 and sys.argv[4] == "kv-cache") else None   # inference: the per-character K / V tables kept across calls
dev = torch.device("cuda", 0)
d, H, hd, cv, Vt, Vc = 2048, 32, 64, 8, 128256, 132
g = torch.Generator(device=dev).manual_seed(1)
r = lambda *s: torch.randn(s, generator=g, device=dev)
Et, Ec = r(Vt, d), r(Vc, d)
w = lambda o, i: r(o, i) / i ** 0.5
wq, wk, wv, wo = w(H * hd, d), w(H * hd, d), w(H * hd, d), w(d, H * hd)
wa, wc = 1 + 0.1 * r(d), 1 + 0.1 * r(d)
toks = torch.randint(0, Vt, (B, T), generator=g, device=dev, dtype=torch.int32)
cid = torch.randint(0, Vc, (B, T, cv), generator=g, device=dev)
lt, lc = torch.ones(1, device=dev), torch.ones(1, device=dev)
matmul = None
if variant != "fp32":
    Et, Ec, wq, wk, wv, wo, wa, wc, lt, lc = (t.bfloat16() for t in (Et, Ec, wq, wk, wv, wo, wa, wc, lt, lc))
    matmul = "fp32" if variant == "bf16-fp32mm" else None
step = lambda: mot.functional.char_swa(toks, cid, Et, Ec, attn_norm_w=wa, char_norm_w=wc, wq=wq, wk=wk, wv=wv, wo=wo, n_heads=H, head_dim=hd,
                                       lambda_tok=lt, lambda_char=lc, matmul=matmul, kv_cache=cache)
ms = bench.timed_launches(step, 10, warm=2)
N = B * T
flop = 2 * 2 * d * H * hd * N + 2 * 2 * 64 * hd * H * N          # q and o projections + 64 keys x (score, value) per head
print(json.dumps({"variant": variant, "kv_cache": cache is not None, "tokens": N, "ms": ms, "tokens_per_s": N / (ms * 1e-3), "dense_TFLOPs": flop / (ms * 1e-3) / 1e12,
                  "note": "whole call: gather+RMSNorm, wq GEMM, 132-row K/V tables, char_swa_kernel, residual (MEAN kernel), wo GEMM"}))
