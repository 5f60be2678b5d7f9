import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import mixture_of_tokenizers_amd as mot
dev = torch.device("cuda", 0)
Vt, D, Db, bpt = 50257, 768, 48, 16
B, T = int(sys.argv[1]), 2048
toks = torch.randint(0, Vt, (B, T), dtype=torch.int32, device=dev)
ids = torch.randint(0, 458, (B, T * bpt), dtype=torch.int64, device=dev)
Et = torch.randn(Vt, D, device=dev); Eb = torch.randn(458, Db, device=dev); g = torch.randn(B, T, D, device=dev)
r = mot.functional.embed_mix_backward(g, toks, Et, Eb, mode="sum", bpt=bpt, ids_a=ids, norm_out=True)
torch.cuda.synchronize()
ws = list(mot.functional._workspaces.values())[0]
w32 = ws.view(torch.int32)
N = B * T
base = 460
counts, cursor, starts = w32[base:base+Vt], w32[base+Vt:base+2*Vt], w32[base+2*Vt:base+3*Vt]
pos = w32[base+3*Vt: base+3*Vt+N]
ref_counts = torch.bincount(toks.view(-1).long(), minlength=Vt).int()
print("counts ok", torch.equal(counts, ref_counts), "cursor ok", torch.equal(cursor, ref_counts))
print("starts ok", torch.equal(starts.long(), torch.cumsum(ref_counts.long(), 0) - ref_counts.long()))
print("pos range", int(pos.min()), int(pos.max()), "perm", torch.equal(torch.sort(pos.long())[0], torch.arange(N, device=dev)))
st = toks.view(-1)[pos.long()]
print("sorted by token", bool((st[1:] >= st[:-1]).all()))
