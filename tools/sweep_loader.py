#!/usr/bin/env python3
"""Dev aid (GPU box): times the loader-side calls (tokens_to_bytes, pull_from_left / right, create_batch, tokens_to_digits,
the character matrix) at the headline batch and at one long row, and prints the bytes each moves per second."""
import json, sys, time
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
import golden_inputs as gi
import mixture_of_tokenizers_amd as mot
from mixture_of_tokenizers_amd import data_creation as dc
dev = torch.device("cuda", 0)


def timed(fn, steps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


for B, T in ((256, 2048), (1, 524288), (8, 65536)):
    bpt = 16
    toks = torch.from_numpy(gi.fineweb_like_tokens(11, B, T, vocab=50257)).to(dev)
    tabL = torch.from_numpy(gi.widen_left_pad(gi.load_real_ttb8(), bpt).astype(np.int32)).to(dev)
    padded = dc.tokens_to_bytes(toks, tabL)
    N = B * T
    res = {"B": B, "T": T}
    res["tokens_to_bytes_ms"] = timed(lambda: dc.tokens_to_bytes(toks, tabL))
    res["pull_left_ms"] = timed(lambda: dc.pull_from_left(padded, bpt, 456, 457))
    res["pull_right_ms"] = timed(lambda: dc.pull_from_right(padded, bpt, 456, 457))
    try:
        res["create_batch_ms"] = timed(lambda: dc.create_batch(toks, bpt, 456, 457, tabL, tabL))
    except Exception as e:  # noqa: BLE001
        res["create_batch_ms"] = f"{type(e).__name__}: {e}"[:80]
    dtab = dc.make_digit_table(3).to(dev)
    res["tokens_to_digits_ms"] = timed(lambda: dc.tokens_to_digits(toks.clamp(max=dtab.shape[0] - 1), dtab))
    res["pull_GBps"] = round(N * bpt * 16 / (res["pull_left_ms"] * 1e-3) / 1e9)
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()}), flush=True)
