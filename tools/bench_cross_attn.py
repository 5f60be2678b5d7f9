#!/usr/bin/env python3
"""Time the cross-attention mixin forward (functional.cross_attn) at config-2 size as one row:
T = 65 536 tokens (the reference asserts batch 1), d 768 (6 heads), bpt 16.  Dev tool; prints one JSON line."""
import argparse
import json
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
import golden_inputs as gi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=65536)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--bpt", type=int, default=16)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--layout", default="as_viewed")
    ap.add_argument("--dual", action="store_true")
    ap.add_argument("--backward", action="store_true", help="time forward + backward through autograd")
    ap.add_argument("--kv-cache", action="store_true", help="inference: keep the per-byte-row K/V tables across calls")
    ap.add_argument("--bf16", action="store_true", help="bfloat16 tables (the production cast)")
    ap.add_argument("--matmul", default=None, choices=["fp32", "bf16"], help="where the products over the tokens run (default: bf16 MFMA with bf16 tables)")
    a = ap.parse_args()
    import mixture_of_tokenizers_amd as mot
    from mixture_of_tokenizers_amd.modules import Rotary
    dev = torch.device("cuda", 0)
    T, D, bpt, H = a.tokens, a.dim, a.bpt, a.dim // 128
    g = torch.Generator(device=dev).manual_seed(1)
    Et = torch.randn((50257, D), generator=g, device=dev)
    Eb = torch.randn((458, D), generator=g, device=dev)
    if a.bf16:
        Et, Eb = Et.bfloat16(), Eb.bfloat16()
    bound = (3 ** 0.5) * 0.5 * D ** -0.5
    q_w = (torch.rand((D, D), generator=g, device=dev) * 2 - 1) * bound
    kv_w = (torch.rand((2, D, D), generator=g, device=dev) * 2 - 1) * bound
    p_w = (torch.rand((D, D), generator=g, device=dev) * 2 - 1) * bound
    toks = torch.from_numpy(gi.fineweb_like_tokens(12345, 1, T, vocab=50257)).to(dev)
    tab = torch.from_numpy(gi.widen_left_pad(gi.load_real_ttb8(), bpt)).to(dev)
    from mixture_of_tokenizers_amd import data_creation as dc
    padded = dc.tokens_to_bytes(toks, tab)
    pulled = dc.pull_from_left(padded, bpt, 456, 457)
    rq, rk = Rotary(128, T).to(dev), Rotary(128, T * bpt).to(dev)
    kw = dict(q_w=q_w, kv_w=kv_w, proj_w=p_w, lambda_factor=torch.tensor(0.5, device=dev), cos_q=rq.cos, sin_q=rq.sin,
              cos_k=rk.cos, sin_k=rk.sin, bpt=bpt, n_heads=H, head_layout=a.layout, ids_b=padded if a.dual else None, matmul=a.matmul)
    if a.backward:
        for t in (Et, Eb, q_w, kv_w, p_w, kw["lambda_factor"]):
            t.requires_grad_(True)
        go = torch.randn((1, T, D), generator=g, device=dev).to(Et.dtype)

        def run():
            x = mot.functional.cross_attn(toks, pulled, Et, Eb, **kw)
            x.backward(go)
            return x
    else:
        cache = {} if a.kv_cache else None

        def run():
            with torch.no_grad():
                return mot.functional.cross_attn(toks, pulled, Et, Eb, kv_cache=cache, **kw)
    for _ in range(3):
        x = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        x = run()
    e1.record(); torch.cuda.synchronize()
    mot.check_status()
    ms = e0.elapsed_time(e1) / a.steps
    flops = 2.0 * T * D * D * 2 + (2.0 * T * bpt * D * D * 2 if a.dual else 2.0 * 458 * D * D * 2)
    print(json.dumps({"backward": a.backward, "tables": str(Et.dtype), "matmul": a.matmul or ("bf16" if a.bf16 else "fp32"),
                      "workload": f"cross_attn T={T} d={D} bpt={bpt} heads={H} layout={a.layout} dual={a.dual}", "ms": ms,
                      "tokens_per_s": T / (ms * 1e-3), "gemm_tflops": flops / (ms * 1e-3) / 1e12,
                      "reference_flops_ratio": (2.0 * T * D * D * 2 + 2.0 * T * bpt * D * D * 2) / flops,
                      "finite": bool(torch.isfinite(x).all())}))


if __name__ == "__main__":
    main()
