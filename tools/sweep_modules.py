#!/usr/bin/env python3
"""Dev aid (GPU box): forward and forward + backward through the reference-shaped modules (FlexibleEmbedding + ByteMixin of
scaled-pre-train for every mixin method; the mathblations DigitFrontEnd) at 64 x 1024 tokens (cross-attention: 1 x 65 536), fp32 and
bf16 embedding tables, to find module-level paths that are far off what their kernels take."""
import json, sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
import golden_inputs as gi
import mixture_of_tokenizers_amd as mot
from mixture_of_tokenizers_amd import modules as M, data_creation as dc
dev = torch.device("cuda", 0)


def timed(fn, steps=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


class Host(torch.nn.Module):
    def __init__(self, dims, vocab, bp, fused):
        super().__init__()
        self.embed = M.FlexibleEmbedding(dims, vocab, bp, fused=fused)
        self.byte_mixin = M.ByteMixin(dims, 65536 + 8, bp)

    def forward(self, t, a, b):
        xt, xb = self.embed(tokens=t, byte_tensor=a, byte_tensor_pulled=b)
        return self.byte_mixin(xt, xb)


V, bpt = 50257, 16
tab = torch.from_numpy(gi.widen_left_pad(gi.load_real_ttb8(), bpt).astype(np.int32)).to(dev)
for method, B, T, dims in (("noop", 64, 1024, dict(model_dim=768, byte_dim=48, token_dim=768)),
                           ("concat", 64, 1024, dict(model_dim=768, byte_dim=32, token_dim=256)),
                           ("cross_attn", 1, 65536, dict(model_dim=768, byte_dim=768, token_dim=768))):
    toks = torch.from_numpy(gi.fineweb_like_tokens(5, B, T, vocab=V)).to(dev)
    padded = dc.tokens_to_bytes(toks, tab)
    pulled = dc.pull_from_left(padded, bpt, 456, 457)
    for two in ((False, True) if method != "noop" else (False,)):
        for fused in (True, False):
            for dt in (torch.float32, torch.bfloat16):
                name = f"{method} two_ids={two} fused={fused} {str(dt).split('.')[1]}"
                try:
                    bp = M.ByteHyperparameters(bytes_per_token=bpt, vocab_size=458, byte_mixin_method=method, pull_in=True, add_padded_and_pulled=two)
                    net = Host(M.ModelDims(**dims), V, bp, fused).to(dev)
                    if dt == torch.bfloat16:
                        for m in net.modules():
                            if isinstance(m, torch.nn.Embedding): m.bfloat16()     # train_gpt.py:1124-1126
                    args = (toks, padded if method != "noop" else None, pulled if method != "noop" else None)
                    with torch.no_grad():
                        fwd = timed(lambda: net(*args))
                    go = None
                    def fb():
                        global go
                        x = net(*args)
                        if go is None or go.shape != x.shape or go.dtype != x.dtype: go = torch.randn_like(x)
                        x.backward(go)
                    both = timed(fb)
                    mot.check_status()
                    print(json.dumps({"case": name, "fwd_ms": round(fwd, 3), "fwd_bwd_ms": round(both, 3)}), flush=True)
                except Exception as e:  # noqa: BLE001
                    print(json.dumps({"case": name, "error": f"{type(e).__name__}: {e}"[:140]}), flush=True)
# mathblations front-end (model.py:304-327)
for method in ("noop", "concat", "cross_attn"):
    for dt in (torch.float32,):
        try:
            cfg = M.GPTConfig(vocab_size=1003, n_embd_tok=768, n_embd_digit=768, length_factor=3, digit_mixin_method=method, n_head=6)
            fe = M.DigitFrontEnd(cfg).to(dev)
            Bm, Tm = (64, 1024) if method != "cross_attn" else (1, 65536)
            toks = torch.randint(0, 1003, (Bm, Tm), device=dev)
            digs = dc.tokens_to_digits(toks, dc.make_digit_table(3).to(dev)) if method != "noop" else None
            a = (toks, digs) if method != "noop" else (toks,)
            with torch.no_grad():
                fwd = timed(lambda: fe(*a))
            def fb2():
                x = fe(*a); x.backward(torch.ones_like(x))
            both = timed(fb2)
            print(json.dumps({"case": f"mathblations {method}", "fwd_ms": round(fwd, 3), "fwd_bwd_ms": round(both, 3)}), flush=True)
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"case": f"mathblations {method}", "error": f"{type(e).__name__}: {e}"[:160]}), flush=True)
