#!/usr/bin/env python3
"""Dev aid (GPU box): forward + backward of the headline SUM mix through autograd (256 x 2048 tokens), for a kernel-level profile
of everything a training step spends around the two kernels.  usage: bench_sum_fb.py [f32|bf16]"""
import json, sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
import golden_inputs as gi
import mixture_of_tokenizers_amd as mot
dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
dev = torch.device("cuda", 0)
B, T, V, Vb, bpt, D, Db = 256, 2048, 50257, 458, 16, 768, 48
g = torch.Generator(device=dev).manual_seed(3)
toks = torch.from_numpy(gi.fineweb_like_tokens(11, B, T, vocab=V)).to(dev)
tab = torch.from_numpy(gi.widen_left_pad(gi.load_real_ttb8(), bpt).astype(np.int16)).to(dev)
Et = torch.nn.Parameter(torch.randn((V, D), generator=g, device=dev).to(dt))
Eb = torch.nn.Parameter(torch.randn((Vb, Db), generator=g, device=dev).to(dt))
go = torch.randn((B, T, D), generator=g, device=dev).to(dt)
def step():
    Et.grad = None; Eb.grad = None
    x = mot.functional.embed_mix(toks, Et, Eb, mode="sum", bpt=bpt, ttb=tab, pull="left", norm_out=True)
    x.backward(go)
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): step()
e1.record(); torch.cuda.synchronize()
print(json.dumps({"dtype": str(dt), "ms_fwd_bwd": e0.elapsed_time(e1) / 20}))
