#!/usr/bin/env python3
"""Time forward + backward of the fused CONCAT_LINEAR front-end (config-2 concat dims: 64x1024 tokens, Dt 256, Db 32,
bpt 16, Dm 768) through autograd.  Dev tool; one JSON line.  --dtype bf16 runs the bf16 tables."""
import argparse, json, sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
import golden_inputs as gi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f32")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--ids", default="fused", choices=["fused", "given"], help="byte ids pulled in-kernel from the token->byte table, or precomputed (the module seam)")
a = ap.parse_args()
import mixture_of_tokenizers_amd as mot
dev = torch.device("cuda", 0)
B, T, Vt, bpt, Dm, Db, Dt = 64, 1024, 50257, 16, 768, 32, 256
dt = torch.float32 if a.dtype == "f32" else torch.bfloat16
g = torch.Generator(device=dev).manual_seed(1)
P = lambda *s: torch.nn.Parameter(torch.randn(s, generator=g, device=dev).to(dt))
Et, Eb = P(Vt, Dt), P(458, Db)
K = Dt + bpt * Db
W = torch.nn.Parameter(((torch.rand((Dm, K), generator=g, device=dev) * 2 - 1) * (3 ** 0.5) * 0.5 * K ** -0.5).to(dt))
toks = torch.from_numpy(gi.fineweb_like_tokens(12345, B, T, vocab=Vt)).to(dev)
tab = torch.from_numpy(gi.widen_left_pad(gi.load_real_ttb8(), bpt)).to(dev)
go = torch.randn((B, T, Dm), generator=g, device=dev).to(dt)
if a.ids == "given":
    from mixture_of_tokenizers_amd import data_creation as dc
    src = dict(ids_a=dc.pull_from_left(dc.tokens_to_bytes(toks, tab), bpt, 456, 457))
else:
    src = dict(ttb=tab, pull="left")
def step():
    x = mot.embed_mix(toks, Et, Eb, mode="concat_linear", bpt=bpt, weight=W, norm_tok=True, norm_byte=True, norm_out=True, **src)
    x.backward(go)
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.steps): step()
e1.record(); torch.cuda.synchronize()
print(json.dumps({"workload": f"concat fwd+bwd {B}x{T} Dt{Dt} Db{Db} Dm{Dm} {a.dtype} ids {a.ids}", "ms_fwd_bwd": e0.elapsed_time(e1) / a.steps}))
