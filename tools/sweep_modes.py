#!/usr/bin/env python3
"""Dev aid (GPU box): times functional.embed_mix forward and forward + backward over the matrix of modes and options at one
batch size, to find combinations that are far off the others (round 3: the two-id cross-attention backward was 50x off before
anyone had timed it).  One line per case: ms forward, ms forward + backward, GB/s of algorithmic traffic of the forward.
usage: sweep_modes.py [B T]"""
import itertools
import json
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
import golden_inputs as gi  # noqa: E402
import mixture_of_tokenizers_amd as mot  # noqa: E402

B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 1024)
dev = torch.device("cuda", 0)
V, Vb, bpt = 50257, 458, 16
g = torch.Generator(device=dev).manual_seed(3)
toks_np = gi.fineweb_like_tokens(11, B, T, vocab=V)
toks = torch.from_numpy(toks_np).to(dev)
tab_np = gi.widen_left_pad(gi.load_real_ttb8(), bpt)
tab16, tab32 = torch.from_numpy(tab_np.astype(np.int16)).to(dev), torch.from_numpy(tab_np.astype(np.int32)).to(dev)
from mixture_of_tokenizers_amd import data_creation as dc  # noqa: E402
padded = dc.tokens_to_bytes(toks, tab32)
pulled = dc.pull_from_left(padded, bpt, 456, 457)
N = B * T


def timed(fn, steps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def case(name, dtype, D, Db, mode, kw, Dt=None, Dm=None):
    Dt = Dt or D
    Et = torch.randn((V, Dt), generator=g, device=dev).to(dtype)
    Eb = torch.randn((Vb, Db), generator=g, device=dev).to(dtype) if mode != "noop" else None
    kw = dict(kw)
    if mode == "concat_linear":
        K = Dt + bpt * Db
        kw["weight"] = ((torch.rand((Dm, K), generator=g, device=dev) * 2 - 1) * (3 ** 0.5) * 0.5 * K ** -0.5).to(dtype)
    es = torch.finfo(dtype).bits // 8
    out_d = Dm or D
    try:
        with torch.no_grad():
            fwd = timed(lambda: mot.functional.embed_mix(toks, Et, Eb, mode=mode, bpt=bpt if mode != "noop" else 0, **kw))
        for t in (Et, Eb, kw.get("weight")):
            if t is not None:
                t.requires_grad_(True)
        go = torch.randn((B, T, out_d), generator=g, device=dev).to(dtype)

        def fb():
            x = mot.functional.embed_mix(toks, Et, Eb, mode=mode, bpt=bpt if mode != "noop" else 0, **kw)
            x = x[0] if isinstance(x, tuple) else x
            x.backward(go)
        try:
            both = timed(fb)
        except RuntimeError as e:
            both = None if "not built" in str(e) or "forward" in str(e) else float("nan")
        mot.check_status()
        gbs = N * (4 + es * (Dt + out_d)) / (fwd * 1e-3) / 1e9
        print(json.dumps({"case": name, "dtype": str(dtype).split(".")[1], "fwd_ms": round(fwd, 4), "fwd_bwd_ms": None if both is None else round(both, 4),
                          "fwd_GBps": round(gbs)}), flush=True)
    except Exception as e:  # noqa: BLE001 -- a sweep: report and go on
        print(json.dumps({"case": name, "dtype": str(dtype).split(".")[1], "error": f"{type(e).__name__}: {e}"[:160]}), flush=True)


for dtype in (torch.float32, torch.bfloat16):
    case("noop", dtype, 768, 48, "noop", {})
    case("noop norm_tok", dtype, 768, 48, "noop", dict(norm_tok=True))
    for pull, norms, two in itertools.product(("left", "right", None), (False, True), (False, True)):
        kw = dict(ttb=tab16, pull=pull, add_padded=two)
        if norms:
            kw.update(norm_tok=True, norm_byte=True, norm_out=True)
        case(f"sum ttb16 pull={pull} norms={norms} add_padded={two}", dtype, 768, 48, "sum", kw)
    case("sum ttb32 pull=left", dtype, 768, 48, "sum", dict(ttb=tab32, pull="left"))
    case("sum ids given", dtype, 768, 48, "sum", dict(ids_a=pulled))
    case("sum ids given a+b", dtype, 768, 48, "sum", dict(ids_a=pulled, ids_b=padded))
    s1, s2 = torch.tensor([0.7], device=dev), torch.tensor([1.2], device=dev)
    case("sum scales", dtype, 768, 48, "sum", dict(ttb=tab16, pull="left", scale_tok=s1, scale_byte=s2))
    case("mean ids given d768", dtype, 768, 768, "mean", dict(ids_a=pulled))
    for norms, two in itertools.product((False, True), (False, True)):
        kw = dict(ttb=tab16, pull="left", add_padded=two)
        if norms:
            kw.update(norm_tok=True, norm_byte=True, norm_out=True)
        case(f"concat_linear K768 norms={norms} add_padded={two}", dtype, 768, 32, "concat_linear", kw, Dt=256, Dm=768)
    case("concat_linear ids given bytes_first", dtype, 768, 32, "concat_linear", dict(ids_a=pulled, bytes_first=True), Dt=256, Dm=768)
    case("concat_linear K1024 Dm1024", dtype, 1024, 48, "concat_linear", dict(ttb=tab16, pull="left", norm_tok=True, norm_byte=True, norm_out=True), Dt=256, Dm=1024)
