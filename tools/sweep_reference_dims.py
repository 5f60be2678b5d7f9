#!/usr/bin/env python3
"""Dev aid (GPU box): does every dimension combination of the reference's experiment scripts run, forward AND backward, through the
reference-shaped modules?  (scaled-pre-train/experiments*.sh, slices.sh: model_dim 1024, 16 bytes per token, token_dim
128 ... 1024, byte_dim 32 ... 128 for the concat mixin; byte_dim = token_dim = model_dim for the cross-attention run.)  A superset
of the combinations the scripts name; prints one line per failure and a summary."""
import itertools, json, sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
import golden_inputs as gi
import mixture_of_tokenizers_amd as mot
from mixture_of_tokenizers_amd import modules as M, data_creation as dc
dev = torch.device("cuda", 0)
V, bpt, B, T = 50257, 16, 4, 256
tab = torch.from_numpy(gi.widen_left_pad(gi.load_real_ttb8(), bpt).astype(np.int32)).to(dev)
toks = torch.from_numpy(gi.fineweb_like_tokens(5, B, T, vocab=V)).to(dev)
padded = dc.tokens_to_bytes(toks, tab)
pulled = dc.pull_from_left(padded, bpt, 456, 457)


class Host(torch.nn.Module):
    def __init__(self, dims, bp):
        super().__init__()
        self.embed = M.FlexibleEmbedding(dims, V, bp)
        self.byte_mixin = M.ByteMixin(dims, B * T * bpt + 8, bp)

    def forward(self, t, a, b):
        return self.byte_mixin(*self.embed(tokens=t, byte_tensor=a, byte_tensor_pulled=b))


cases = [("concat", td, bd, two) for td, bd in itertools.product((128, 256, 512, 768, 896, 1024), (32, 48, 56, 64, 128)) for two in (False, True)]
cases += [("cross_attn", 1024, 1024, False), ("cross_attn", 1024, 1024, True), ("noop", 1024, 48, False)]
ok = bad = 0
for method, td, bd, two in cases:
    for dt in (torch.float32, torch.bfloat16):
        name = f"{method} token_dim={td} byte_dim={bd} two_ids={two} {str(dt).split('.')[1]}"
        try:
            bp = M.ByteHyperparameters(bytes_per_token=bpt, vocab_size=458, byte_mixin_method=method, pull_in=True, add_padded_and_pulled=two)
            net = Host(M.ModelDims(model_dim=1024, byte_dim=bd, token_dim=td), bp).to(dev)
            if dt == torch.bfloat16:
                for m in net.modules():
                    if isinstance(m, torch.nn.Embedding): m.bfloat16()
            tk = toks if method != "cross_attn" else toks.reshape(1, -1)
            pa, pu = (padded, pulled) if method != "cross_attn" else (padded.reshape(1, -1), pulled.reshape(1, -1))
            x = net(tk, None if method == "noop" else pa, None if method == "noop" else pu)
            x.backward(torch.randn_like(x))
            mot.check_status()
            assert bool(torch.isfinite(x.float()).all()) and all(p.grad is not None and bool(torch.isfinite(p.grad.float()).all()) for p in net.parameters() if p.requires_grad)
            ok += 1
        except Exception as e:  # noqa: BLE001
            bad += 1
            print(json.dumps({"case": name, "error": f"{type(e).__name__}: {e}"[:200]}), flush=True)
print(json.dumps({"ran": ok + bad, "ok": ok, "failed": bad}))
