"""Dev aid (GPU box): config 5 (MEAN, 256 x 8192 tokens, d 2048) with random token ids, sequential token ids and ONE token id --
what the gather costs and what the 17.2 GB of output writes cost on their own.  usage: python3 tools/c5_floor.py"""
import sys, torch, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import mixture_of_tokenizers_amd as mot
dev = torch.device("cuda:0")
B, T, V, bpt, D, Vb = 256, 8192, 128256, 8, 2048, 132
g = torch.Generator(device=dev).manual_seed(1)
Et = torch.randn((V, D), generator=g, device=dev); Eb = torch.randn((Vb, D), generator=g, device=dev)
chars = torch.randint(0, Vb, (B, T * bpt), device=dev, generator=g)
lt, lc = torch.tensor(1.0, device=dev), torch.tensor(0.5, device=dev)
out = torch.empty((B, T, D), device=dev)
def run(toks, name):
    plan = mot.embed_mix_plan(toks, Et, Eb, mode="mean", bpt=bpt, ids_a=chars, scale_tok=lt, scale_byte=lc, out=out)
    for _ in range(3): plan()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): plan()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    nbytes = B * T * (4 + 8 * bpt + 2 * 4 * D)
    print(name, "ms", round(ms, 3), "TB/s", round(nbytes / ms / 1e9, 3), "frac", round(nbytes / ms / 1e9 / 8, 3))
rnd = torch.randint(0, V, (B, T), device=dev, generator=g, dtype=torch.int32)
seq = (torch.arange(B * T, device=dev, dtype=torch.int64) % V).to(torch.int32).view(B, T)
run(rnd, "random tokens")
run(seq, "sequential tokens")
run(torch.zeros_like(rnd), "one token")
