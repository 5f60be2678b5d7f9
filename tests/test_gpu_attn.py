"""-m gpu: the cross-attention byte mixin (train_gpt.py:243-300, 446-464) through the module interface
(FlexibleEmbedding -> ByteMixin(byte_mixin_method="cross_attn")) and through functional.cross_attn.

Tolerance (fp32 path with two D x D contractions, a per-head norm and a softmax in between): measured against the
float64 evaluation, relative to the largest output entry --
    max|hip - ref64| <= 2 * max(max|ref32 - ref64|, 1e-6 * max|ref64|)
where ref32 / ref64 are the reference's own fp32 and float64 outputs (tests/golden/cross_attn.npz); i.e. the kernel
may be at most twice as far from exact as the reference's fp32 CPU run (same criterion as CONCAT_LINEAR).  Against
the float64 oracle at sizes without a golden the bar is 5e-6 of max|ref64|."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import oracle as orc
from util_gpu import DEV, dev, f32, host, rel

pytestmark = pytest.mark.gpu
G = gi.GOLDEN_DIR


@pytest.fixture(scope="module")
def mot():
    import mixture_of_tokenizers_amd as m
    return m


def build(M, Vt, D, bpt, T, seed, mode):
    bp = M.ByteHyperparameters(bytes_per_token=bpt, vocab_size=gi.BYTE_VOCAB, byte_mixin_method="cross_attn", pull_in=True,
                               add_padded_and_pulled=mode == "padded_and_pulled")
    dims = M.ModelDims(model_dim=D, byte_dim=D, token_dim=D)
    embed, mixin = M.FlexibleEmbedding(dims, Vt, bp).to(DEV), M.ByteMixin(dims, T, bp).to(DEV)
    q_w, kv_w, p_w = gi.cross_weights(seed + 3, D)
    ca = mixin.mixin.mixin
    with torch.no_grad():
        embed.embed_tokens.weight.copy_(dev(f32(gi.normal_table(seed + 1, Vt, D))))
        embed.embed_bytes.weight.copy_(dev(f32(gi.normal_table(seed + 2, gi.BYTE_VOCAB, D))))
        ca.q_w.copy_(dev(q_w)); ca.kv_w.copy_(dev(kv_w)); ca.c_proj.weight.copy_(dev(p_w)); ca.lambda_factor.fill_(0.7)
    return embed, mixin


@pytest.mark.parametrize("mode", ["pulled", "padded_and_pulled"])
@pytest.mark.parametrize("case", gi.CROSS_CASES, ids=lambda c: c[0])
def test_cross_attn_modules_vs_reference(mot, case, mode):
    from mixture_of_tokenizers_amd import modules as M
    name, Vt, D, bpt, T, seed = case
    z = np.load(G / "cross_attn.npz")
    embed, mixin = build(M, Vt, D, bpt, T, seed, mode)
    assert sorted(mixin.state_dict()) == ["mixin.mixin.c_proj.weight", "mixin.mixin.kv_w", "mixin.mixin.lambda_factor", "mixin.mixin.q_w"]
    with torch.no_grad():
        x = mixin(*embed(tokens=dev(z[f"{name}/tokens"]), byte_tensor=dev(z[f"{name}/padded"]), byte_tensor_pulled=dev(z[f"{name}/pulled"])))
    assert x.shape == (1, T, D) and x.dtype == torch.float32
    r32, r64 = z[f"{name}/{mode}/f32/x"], z[f"{name}/{mode}/f64/x"]
    scale = np.abs(r64).max()
    bar = 2 * max(np.abs(r32.astype(np.float64) - r64).max(), 1e-6 * scale)
    assert np.abs(host(x).astype(np.float64) - r64).max() <= bar


@pytest.mark.parametrize("D,bpt,Vt,T,layout,dual,seed", [
    (768, 16, 4096, 300, "as_viewed", False, 9901),      # C2 dims: 6 heads
    (768, 16, 4096, 300, "per_token", False, 9902),
    (1024, 16, 2048, 130, "as_viewed", False, 9903),     # production dims: 8 heads
    (256, 5, 512, 77, "as_viewed", True, 9904),          # two id tensors: keys depend on the id pair
    (384, 7, 300, 33, "per_token", True, 9905),
    (768, 20, 512, 50, "as_viewed", False, 9906),        # more than 16 keys: the key-loop attention kernel
    (768, 11, 512, 50, "as_viewed", False, 9907),        # an odd key count in the lane-per-(key, quarter) kernel
])
def test_cross_attn_vs_oracle(mot, D, bpt, Vt, T, layout, dual, seed):
    from mixture_of_tokenizers_amd.modules import Rotary
    H = D // 128
    tab = gi.synth_ttb(seed + 1, Vt, bpt, "left", mean_valid=min(4.4, bpt / 2))
    toks = gi.fineweb_like_tokens(seed, 1, T, vocab=Vt, eot_p=0.01)
    Et, Eb = f32(gi.normal_table(seed + 2, Vt, D)), f32(gi.normal_table(seed + 3, gi.BYTE_VOCAB, D))
    q_w, kv_w, p_w = gi.cross_weights(seed + 4, D)
    padded = orc.tokens_to_bytes(toks, tab.astype(np.float32))
    pulled = orc.pull_from_left(padded, bpt, gi.PAD, gi.EOT)
    rq, rk = Rotary(128, T + 3), Rotary(128, T * bpt + 5)       # longer buffers than needed, as with max_seq_len
    rot = [rq.cos, rq.sin, rk.cos, rk.sin]
    d64 = lambda a: np.asarray(a, dtype=np.float64)
    ref = orc.cross_attn(toks[0], pulled[0], padded[0] if dual else None, d64(Et), d64(Eb), d64(q_w), d64(kv_w), d64(p_w), 0.35,
                         *[r.numpy() for r in rot], bpt=bpt, n_heads=H, dtype=np.float64, head_layout=0 if layout == "as_viewed" else 1)
    x = mot.functional.cross_attn(dev(toks), dev(pulled), dev(Et), dev(Eb), ids_b=dev(padded) if dual else None,
                                  q_w=dev(f32(q_w)), kv_w=dev(f32(kv_w)), proj_w=dev(f32(p_w)), lambda_factor=torch.tensor(0.35, device=DEV),
                                  cos_q=rot[0].to(DEV), sin_q=rot[1].to(DEV), cos_k=rot[2].to(DEV), sin_k=rot[3].to(DEV),
                                  bpt=bpt, n_heads=H, head_layout=layout)
    mot.check_status()
    assert np.abs(host(x)[0].astype(np.float64) - ref).max() <= 5e-6 * np.abs(ref).max()


def test_cross_attn_errors(mot):
    from mixture_of_tokenizers_amd import modules as M
    Vt, D, bpt, T = 97, 256, 8, 24
    embed, mixin = build(M, Vt, D, bpt, T, 702, "pulled")
    toks = torch.zeros((1, T), dtype=torch.int32, device=DEV)
    ids = torch.zeros((1, T * bpt), dtype=torch.int64, device=DEV)
    with torch.no_grad():
        with pytest.raises(AssertionError, match="batch size = 1"):    # train_gpt.py:275
            mixin(*embed(toks.repeat(2, 1), ids.repeat(2, 1), ids.repeat(2, 1)))
        with pytest.raises(AssertionError):                            # Rotary.forward's length assert, line 200
            mixin(*embed(toks.repeat(1, 2), ids.repeat(1, 2), ids.repeat(1, 2)))
        bad = ids.clone(); bad[0, 5] = 999
        mixin(*embed(toks, bad, bad))
        with pytest.raises(IndexError):
            mot.check_status()


# ------------------------------------------------------------------------------------------------
# backward (loss.backward() through the modules, train_gpt.py:1319).  Bar: per gradient tensor,
# max|hip - ref64| <= 5e-5 * max|ref64| -- a chain of five fp32 contractions, a softmax and two norms, accumulated with
# float atomics; the reference's own fp32 run is not closer to its float64 gradients.
# ------------------------------------------------------------------------------------------------
GTOL = 5e-5


grel = rel   # self-describing on failure (util_gpu.RelErr)


# (the fixture holds the reference's autograd run for both embeddings of the one-head case and the one-id embedding of the two-head case:
#  oracle/gen_golden.py, gen_cross_attn_grads)
@pytest.mark.parametrize("case,mode", [(gi.CROSS_CASES[0], "pulled"), (gi.CROSS_CASES[0], "padded_and_pulled"), (gi.CROSS_CASES[1], "pulled")],
                         ids=lambda v: v if isinstance(v, str) else v[0])
def test_cross_attn_backward_vs_reference_autograd(mot, case, mode):
    from mixture_of_tokenizers_amd import modules as M
    name, Vt, D, bpt, T, seed = case
    z, zg = np.load(G / "cross_attn.npz"), np.load(G / "cross_attn_grads.npz")
    embed, mixin = build(M, Vt, D, bpt, T, seed, mode)
    x = mixin(*embed(tokens=dev(z[f"{name}/tokens"]), byte_tensor=dev(z[f"{name}/padded"]), byte_tensor_pulled=dev(z[f"{name}/pulled"])))
    assert x.requires_grad
    (x * dev(zg[f"{name}/g"])).sum().backward()
    ca = mixin.mixin.mixin
    for key, p in (("d_tok", embed.embed_tokens.weight), ("d_byte", embed.embed_bytes.weight)):
        full = np.zeros(tuple(p.shape), np.float64)
        full[zg[f"{name}/{mode}/{key}_rows"]] = zg[f"{name}/{mode}/{key}_vals"]
        assert grel(host(p.grad), full) < GTOL, key
    assert grel(host(ca.q_w.grad), zg[f"{name}/{mode}/d_qw"]) < GTOL
    assert grel(host(ca.kv_w.grad), zg[f"{name}/{mode}/d_kvw"]) < GTOL
    assert grel(host(ca.c_proj.weight.grad), zg[f"{name}/{mode}/d_pw"]) < GTOL
    assert abs(float(ca.lambda_factor.grad) - zg[f"{name}/{mode}/d_lambda"][0]) < GTOL * max(1.0, abs(zg[f"{name}/{mode}/d_lambda"][0]))


@pytest.mark.parametrize("D,bpt,Vt,T,layout,norms,seed,dual", [
    (768, 16, 4096, 150, "as_viewed", True, 9951, False),      # C2 dims
    (768, 16, 4096, 150, "per_token", True, 9952, False),
    (256, 5, 300, 77, "as_viewed", False, 9953, False),        # no embedding norms
    (1024, 8, 512, 64, "as_viewed", True, 9954, False),        # production dims
    (640, 7, 300, 53, "as_viewed", True, 9955, False),         # 5 heads: the per-head-slice variant of the table-row reduction
    (640, 7, 300, 53, "per_token", True, 9956, False),
    (384, 3, 300, 41, "per_token", False, 9957, False),        # 3 heads, bpt smaller than the head count
    (768, 4, 300, 45, "as_viewed", True, 9958, False),         # 6 heads, 4 keys: the slots of a position run over more than two queries
    (256, 8, 300, 640, "as_viewed", True, 9967, False),        # 640 tokens: dy and dxq on the LDS-DMA product kernel (transposed weights)
    (256, 4, 300, 160, "as_viewed", True, 9968, True),         # ... and dxkv, over the 640 kv positions of two id tensors
    (256, 2, 300, 19, "as_viewed", True, 9966, False),         # two keys per token: two live quads in the lane-per-key kernels (with ONE key the
                                                               # token-table and q_w gradients are exactly zero in the reference: nothing to scale a bar by)
    (768, 20, 300, 37, "as_viewed", True, 9959, False),        # more than 16 keys: the key-loop attention kernels
    (256, 20, 300, 37, "per_token", True, 9960, True),
    # two id tensors (add_padded_and_pulled, train_gpt.py:364-372): key / value rows per kv position
    (768, 16, 4096, 150, "as_viewed", True, 9961, True),
    (768, 16, 4096, 150, "per_token", True, 9962, True),
    (256, 5, 300, 77, "as_viewed", False, 9963, True),
    (640, 7, 300, 53, "as_viewed", True, 9965, True),
    (1024, 8, 512, 64, "per_token", True, 9964, True),
])
def test_cross_attn_backward_vs_oracle(mot, D, bpt, Vt, T, layout, norms, seed, dual):
    from mixture_of_tokenizers_amd.modules import Rotary
    H = D // 128
    tab = gi.synth_ttb(seed + 1, Vt, bpt, "left", mean_valid=min(4.4, bpt / 2))
    toks = gi.fineweb_like_tokens(seed, 1, T, vocab=Vt, eot_p=0.01)
    Et, Eb = f32(gi.normal_table(seed + 2, Vt, D)), f32(gi.normal_table(seed + 3, gi.BYTE_VOCAB, D))
    q_w, kv_w, p_w = (f32(a) for a in gi.cross_weights(seed + 4, D))
    padded = orc.tokens_to_bytes(toks, tab.astype(np.float32))
    pulled = orc.pull_from_left(padded, bpt, gi.PAD, gi.EOT)
    g = f32(np.random.RandomState(seed + 5).standard_normal((1, T, D)))
    rq, rk = Rotary(128, T), Rotary(128, T * bpt)
    rot = [rq.cos, rq.sin, rk.cos, rk.sin]
    d64 = lambda a: np.asarray(a, dtype=np.float64)
    ref = orc.cross_attn_bwd(toks[0], pulled[0], padded[0] if dual else None, d64(Et), d64(Eb), d64(q_w), d64(kv_w), d64(p_w), 0.35, *[r.numpy() for r in rot],
                             d64(g), bpt=bpt, n_heads=H, norm_tok=norms, norm_byte=norms, head_layout=0 if layout == "as_viewed" else 1)
    P = lambda a: torch.nn.Parameter(dev(a))
    pEt, pEb, pq, pkv, pp, plam = P(Et), P(Eb), P(q_w), P(kv_w), P(p_w), torch.nn.Parameter(torch.tensor(0.35, device=DEV))
    x = mot.functional.cross_attn(dev(toks), dev(pulled), pEt, pEb, q_w=pq, kv_w=pkv, proj_w=pp, lambda_factor=plam,
                                  ids_b=dev(padded) if dual else None,
                                  cos_q=rot[0].to(DEV), sin_q=rot[1].to(DEV), cos_k=rot[2].to(DEV), sin_k=rot[3].to(DEV),
                                  bpt=bpt, n_heads=H, norm_tok=norms, norm_byte=norms, head_layout=layout)
    (x * dev(g)).sum().backward()
    mot.check_status()
    for p, key in ((pEt, "tok_table"), (pEb, "byte_table"), (pq, "q_w"), (pkv, "kv_w"), (pp, "proj_w")):
        assert grel(host(p.grad), ref[key]) < GTOL, key
    # (one scalar: the sum of rows x HD products of table sums that arrive through fp32 atomics in arrival order -- it moves from run to
    #  run, 5.5e-5 in one of two runs of the 768-wide per_token case: twice the bar of the tensors)
    assert abs(float(plam.grad) - ref["lambda_factor"][0]) < 2 * GTOL * max(1.0, abs(ref["lambda_factor"][0]))


def test_cross_attn_locality_at_size(mot):
    """T = 16 384 (beyond what the oracle is run on): with head_layout="per_token" a byte of token t only influences output
    row t, and a token id only its own row; with the reference's "as_viewed" reshape (more than one head) a byte reaches
    the rows of other tokens -- and exactly the (head, token) pairs the reshape maps its (position, head) rows to."""
    from mixture_of_tokenizers_amd.modules import Rotary
    T, D, bpt, Vt = 16384, 256, 8, 1000
    H = D // 128
    gen = torch.Generator(device=DEV).manual_seed(3)
    Et, Eb = torch.randn((Vt, D), generator=gen, device=DEV), torch.randn((gi.BYTE_VOCAB, D), generator=gen, device=DEV)
    q_w, kv_w, p_w = (dev(f32(a)) for a in gi.cross_weights(77, D))
    toks = torch.randint(0, Vt, (1, T), generator=gen, device=DEV, dtype=torch.int32)
    ids = torch.randint(0, 256, (1, T * bpt), generator=gen, device=DEV, dtype=torch.int64)
    rq, rk = Rotary(128, T).to(DEV), Rotary(128, T * bpt).to(DEV)
    kw = dict(q_w=q_w, kv_w=kv_w, proj_w=p_w, lambda_factor=torch.tensor(0.5, device=DEV), cos_q=rq.cos, sin_q=rq.sin, cos_k=rk.cos, sin_k=rk.sin,
              bpt=bpt, n_heads=H)
    t_mod, c_mod = 9001, 3
    ids2 = ids.clone(); ids2[0, t_mod * bpt + c_mod] = (ids[0, t_mod * bpt + c_mod] + 1) % 256
    with torch.no_grad():
        for layout in ("per_token", "as_viewed"):
            a = mot.functional.cross_attn(toks, ids, Et, Eb, head_layout=layout, **kw)[0]
            b = mot.functional.cross_attn(toks, ids2, Et, Eb, head_layout=layout, **kw)[0]
            assert bool(torch.isfinite(a).all())
            changed = set(torch.nonzero((a != b).any(dim=1)).view(-1).tolist())
            if layout == "per_token":
                assert changed == {t_mod}
            else:   # kv position p, head hk sits in flat row r = p*H + hk, which query (h, t) = divmod(r // bpt, T) reads
                p = t_mod * bpt + c_mod
                expect = {((p * H + hk) // bpt) % T for hk in range(H)}
                assert changed == expect and changed != {t_mod}
    mot.check_status()


def test_cross_attn_kv_table_cache(mot):
    """No-grad calls through the module keep the per-byte-row K/V tables while byte table, kv_w and lambda are unchanged
    (tensor versions), and rebuild them after an in-place update."""
    from mixture_of_tokenizers_amd import modules as M
    name, Vt, D, bpt, T, seed = gi.CROSS_CASES[1]
    z = np.load(G / "cross_attn.npz")
    embed, mixin = build(M, Vt, D, bpt, T, seed, "pulled")
    args = dict(tokens=dev(z[f"{name}/tokens"]), byte_tensor=dev(z[f"{name}/padded"]), byte_tensor_pulled=dev(z[f"{name}/pulled"]))
    ca = mixin.mixin.mixin
    with torch.no_grad():
        a = mixin(*embed(**args))
        key1 = ca._kv_cache["key"]
        b = mixin(*embed(**args))                       # second call: tables reused
        assert ca._kv_cache["key"] == key1 and torch.equal(a, b)
        ca.kv_w.mul_(1.5)                               # in-place update bumps the version: tables rebuilt
        c = mixin(*embed(**args))
        assert ca._kv_cache["key"] != key1 and not torch.equal(a, c)
        fresh = mot.functional.cross_attn(args["tokens"], args["byte_tensor_pulled"], embed.embed_tokens.weight, embed.embed_bytes.weight,
                                          q_w=ca.q_w, kv_w=ca.kv_w, proj_w=ca.c_proj.weight, lambda_factor=ca.lambda_factor,
                                          cos_q=ca.rotary_q.cos, sin_q=ca.rotary_q.sin, cos_k=ca.rotary_k.cos, sin_k=ca.rotary_k.sin,
                                          bpt=bpt, n_heads=D // 128)
        assert torch.equal(c, fresh)
        ca.lambda_factor.fill_(0.9)
        d = mixin(*embed(**args))
        assert not torch.equal(c, d)


@pytest.mark.parametrize("T,matmul", [(200, None), (700, "bf16")])   # (bf16: narrowing kernels + the 256 x 256 product kernel in the graph)
def test_cross_attn_forward_is_capturable_in_a_hip_graph(mot, T, matmul):
    """The mixin forward is a fixed sequence of kernels on the given stream (seam gather, dense products, table kernels, attention):
    capture it, change tokens, ids and a weight in place, replay, compare with an eager call."""
    from mixture_of_tokenizers_amd.modules import Rotary
    D, bpt, Vt, H = 256, 8, 500, 2
    rs = np.random.RandomState(9981)
    toks = dev(rs.randint(0, Vt, (1, T)).astype(np.int32))
    ids = dev(rs.randint(0, gi.BYTE_VOCAB, (1, T * bpt)).astype(np.int64))
    Et, Eb = dev(f32(gi.normal_table(9982, Vt, D))), dev(f32(gi.normal_table(9983, gi.BYTE_VOCAB, D)))
    q_w, kv_w, p_w = (dev(f32(a)) for a in gi.cross_weights(9984, D))
    rq, rk = Rotary(128, T), Rotary(128, T * bpt)
    kw = dict(q_w=q_w, kv_w=kv_w, proj_w=p_w, lambda_factor=torch.tensor(0.6, device=DEV), cos_q=rq.cos.to(DEV), sin_q=rq.sin.to(DEV),
              cos_k=rk.cos.to(DEV), sin_k=rk.sin.to(DEV), bpt=bpt, n_heads=H, matmul=matmul)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        mot.functional.cross_attn(toks, ids, Et, Eb, **kw)          # warm-up: allocates the workspace
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph, stream=s):
        out = mot.functional.cross_attn(toks, ids, Et, Eb, **kw)
    toks.copy_(dev(rs.randint(0, Vt, (1, T)).astype(np.int32)))
    ids.copy_(dev(rs.randint(0, gi.BYTE_VOCAB, (1, T * bpt)).astype(np.int64)))
    q_w.mul_(1.25)
    graph.replay()
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = mot.functional.cross_attn(toks, ids, Et, Eb, **kw)
    assert torch.equal(out, ref)


def test_cross_attn_backward_is_capturable_in_a_hip_graph(mot):
    """mot_cross_attn_bwd: zeroing kernels, the counting sort of the kv positions, the attention backward, the table-row sums, the
    dense products and the embedding backward, all enqueued without a sync: capture, change the inputs in place, replay, compare."""
    from mixture_of_tokenizers_amd.modules import Rotary
    D, bpt, Vt, T, H = 256, 8, 500, 160, 2
    rs = np.random.RandomState(9991)
    toks = dev(rs.randint(0, Vt, (1, T)).astype(np.int32))
    ids = dev(rs.randint(0, gi.BYTE_VOCAB, (1, T * bpt)).astype(np.int64))
    Et, Eb = dev(f32(gi.normal_table(9992, Vt, D))), dev(f32(gi.normal_table(9993, gi.BYTE_VOCAB, D)))
    q_w, kv_w, p_w = (dev(f32(a)) for a in gi.cross_weights(9994, D))
    g = dev(f32(rs.standard_normal((1, T, D))))
    rq, rk = Rotary(128, T), Rotary(128, T * bpt)
    kw = dict(q_w=q_w, kv_w=kv_w, proj_w=p_w, lambda_factor=torch.tensor(0.6, device=DEV), cos_q=rq.cos.to(DEV), sin_q=rq.sin.to(DEV),
              cos_k=rk.cos.to(DEV), sin_k=rk.sin.to(DEV), bpt=bpt, n_heads=H)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        mot.functional.cross_attn_backward(g, toks, ids, Et, Eb, **kw)          # warm-up: allocates the workspace
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        got = mot.functional.cross_attn_backward(g, toks, ids, Et, Eb, **kw)
    toks.copy_(dev(rs.randint(0, Vt, (1, T)).astype(np.int32)))
    ids.copy_(dev(rs.randint(0, gi.BYTE_VOCAB, (1, T * bpt)).astype(np.int64)))
    g.copy_(dev(f32(rs.standard_normal((1, T, D)))))
    graph.replay()
    torch.cuda.synchronize()
    ref = mot.functional.cross_attn_backward(g, toks, ids, Et, Eb, **kw)
    for k in ("tok_table", "byte_table", "q_w", "kv_w", "proj_w", "lambda_factor"):
        assert grel(host(got[k]), host(ref[k])) < 2 * GTOL, k        # two GPU results (atomic order differs)


# ------------------------------------------------------------------------------------------------
# mathblations: DigitMixinCrossAttention (model.py:239-253 -> 89-154) through wte / dte / digit_mixin of DigitFrontEnd,
# against what the reference modules produced (tests/golden/digit_cross_attn.npz).  Same bars as above.
# ------------------------------------------------------------------------------------------------
def build_digit(case):
    from mixture_of_tokenizers_amd import modules as M
    name, lf, mtpn, D, H, B, seed = case
    z = np.load(G / "digit_cross_attn.npz")
    Vt = 10 ** lf + 3
    T = z[f"{name}/x_tokens"].shape[1]
    cfg = M.GPTConfig(vocab_size=Vt, n_layer=1, n_head=H, n_embd_tok=D, n_embd_digit=D, T=T + 1, length_factor=lf,
                      digit_mixin_method="cross_attn")
    net = M.DigitFrontEnd(cfg).to(DEV)
    ca = net.digit_mixin.cross_attn
    with torch.no_grad():
        net.wte.weight.copy_(dev(f32(gi.normal_table(seed + 1, Vt, D))))
        net.dte.weight.copy_(dev(f32(gi.normal_table(seed + 2, 14, D))))
        for lin, w in zip((ca.c_q, ca.c_k, ca.c_v, ca.c_proj), gi.digit_cross_weights(seed + 3, D)):
            lin.weight.copy_(dev(w))
    return z, net, T


@pytest.mark.parametrize("case", gi.DIGIT_CROSS_CASES, ids=lambda c: c[0])
def test_digit_cross_attn_modules_vs_reference(mot, case):
    name, lf, mtpn, D, H, B, seed = case
    z, net, T = build_digit(case)
    assert sorted(net.digit_mixin.state_dict()) == ["cross_attn.c_k.weight", "cross_attn.c_proj.weight", "cross_attn.c_q.weight",
                                                    "cross_attn.c_v.weight"]
    with torch.no_grad():
        x = net(dev(z[f"{name}/x_tokens"]), dev(z[f"{name}/x_digit_tokens"]))
    mot.check_status()
    assert x.shape == (B, T, D) and x.dtype == torch.float32
    r32, r64 = z[f"{name}/f32/x"], z[f"{name}/f64/x"]
    bar = 2 * max(np.abs(r32.astype(np.float64) - r64).max(), 1e-6 * np.abs(r64).max())
    assert np.abs(host(x).astype(np.float64) - r64).max() <= bar
    # one row alone gives the same values: rows of a batch do not see each other
    with torch.no_grad():
        x3 = net(dev(z[f"{name}/x_tokens"][3]), dev(z[f"{name}/x_digit_tokens"][3]))
    assert torch.equal(x3.reshape(T, D), x[3])


@pytest.mark.parametrize("case", gi.DIGIT_CROSS_CASES, ids=lambda c: c[0])
def test_digit_cross_attn_backward_vs_reference_autograd(mot, case):
    name, lf, mtpn, D, H, B, seed = case
    z, net, T = build_digit(case)
    x = net(dev(z[f"{name}/x_tokens"]), dev(z[f"{name}/x_digit_tokens"]))
    (x * dev(z[f"{name}/g"])).sum().backward()
    mot.check_status()
    ca = net.digit_mixin.cross_attn
    for key, p in (("d_tok", net.wte.weight), ("d_digit", net.dte.weight)):
        full, rows = host(p.grad), z[f"{name}/{key}_rows"]
        assert grel(full[rows], z[f"{name}/{key}_vals"]) <= GTOL, key
        mask = np.ones(len(full), bool); mask[rows] = False
        assert not full[mask].any(), key
    for key, lin in (("d_cq", ca.c_q), ("d_ck", ca.c_k), ("d_cv", ca.c_v), ("d_cproj", ca.c_proj)):
        assert grel(host(lin.weight.grad), z[f"{name}/{key}"]) <= GTOL, key


def test_digit_cross_attn_errors(mot):
    from mixture_of_tokenizers_amd import modules as M
    with pytest.raises(AssertionError):                                 # model.py:241
        M.make_digit_mixin(M.GPTConfig(n_embd_tok=256, n_embd_digit=128, n_head=2, digit_mixin_method="cross_attn"))
    with pytest.raises(NotImplementedError, match="head_dim 128"):
        M.make_digit_mixin(M.GPTConfig(n_embd_tok=256, n_embd_digit=256, n_head=4, digit_mixin_method="cross_attn"))
    z, net, T = build_digit(gi.DIGIT_CROSS_CASES[1])
    toks, digs = dev(z["runcfg/x_tokens"]), dev(z["runcfg/x_digit_tokens"])
    with torch.no_grad():
        with pytest.raises(AssertionError, match="KV length"):          # model.py:132
            net(toks, digs[:, :-4])
        with pytest.raises(AssertionError, match="Batch sizes"):        # model.py:129
            net(toks, digs[:-1])
        with pytest.raises(AssertionError, match="Digits must be provided"):   # model.py:321
            net(toks)


def test_cross_attn_bf16_tables_two_id_tensors(mot):
    """bf16 tables with the add_padded_and_pulled embedding (train_gpt.py:364-372): a key / value row per kv position, so the K / V
    projections, dW_kv and dxkv are products over T * bpt rows and run on the bf16 MFMA with the rest (xkv, dkv rounded to bf16).
    Float64 oracle on the bf16-valued operands, without those roundings: forward within four bf16 steps of the larger of the
    output and the outputs' rms, gradients within 1 % of each tensor's largest entry."""
    from mixture_of_tokenizers_amd.modules import Rotary
    D, bpt, Vt, T, seed = 256, 8, 512, 96, 9983
    H = D // 128
    tab = gi.synth_ttb(seed + 1, Vt, bpt, "left", mean_valid=4.0)
    toks = gi.fineweb_like_tokens(seed, 1, T, vocab=Vt, eot_p=0.01)
    Et, Eb = orc.bf16_round(gi.normal_table(seed + 2, Vt, D)), orc.bf16_round(gi.normal_table(seed + 3, gi.BYTE_VOCAB, D))
    q_w, kv_w, p_w = (f32(a) for a in gi.cross_weights(seed + 4, D))
    padded = orc.tokens_to_bytes(toks, tab.astype(np.float32))
    pulled = orc.pull_from_left(padded, bpt, gi.PAD, gi.EOT)
    g = orc.bf16_round(np.random.RandomState(seed + 5).standard_normal((1, T, D)))
    rq, rk = Rotary(128, T), Rotary(128, T * bpt)
    rot = [rq.cos, rq.sin, rk.cos, rk.sin]
    d64 = lambda a: np.asarray(a, dtype=np.float64)
    used = lambda w: d64(orc.bf16_round(w))
    lam = float(orc.bf16_round(np.array([0.35]))[0])
    args = (toks[0], padded[0], pulled[0], d64(Et), d64(Eb), used(q_w), used(kv_w), used(p_w), lam, *[r.numpy() for r in rot])
    orc.set_eps(2.0 ** -7)   # norm() on bf16 tensors: eps = finfo(bfloat16).eps
    try:
        ref = orc.cross_attn(*args, bpt=bpt, n_heads=H, dtype=np.float64, head_layout=0)
        refg = orc.cross_attn_bwd(*args, d64(g), bpt=bpt, n_heads=H, norm_tok=True, norm_byte=True, head_layout=0)
    finally:
        orc.set_eps(0.0)
    P = lambda a, dt=None: torch.nn.Parameter(dev(a) if dt is None else dev(a).to(dt))
    pEt, pEb = P(Et, torch.bfloat16), P(Eb, torch.bfloat16)
    pq, pkv, pp, plam = P(q_w), P(kv_w), P(p_w), torch.nn.Parameter(torch.tensor(0.35, device=DEV))
    x = mot.functional.cross_attn(dev(toks), dev(padded), pEt, pEb, q_w=pq, kv_w=pkv, proj_w=pp, lambda_factor=plam, ids_b=dev(pulled),
                                  cos_q=rot[0].to(DEV), sin_q=rot[1].to(DEV), cos_k=rot[2].to(DEV), sin_k=rot[3].to(DEV), bpt=bpt, n_heads=H)
    got, want = host(x.float())[0].astype(np.float64), d64(ref)
    assert (np.abs(got - want) <= 4 * 2.0 ** -7 * np.maximum(np.abs(want), np.sqrt((want ** 2).mean()))).all()
    (x.float() * dev(g)).sum().backward()
    mot.check_status()
    for p, key in ((pEt, "tok_table"), (pEb, "byte_table"), (pq, "q_w"), (pkv, "kv_w"), (pp, "proj_w")):
        assert grel(host(p.grad.float()), refg[key]) < 1e-2, key


@pytest.mark.parametrize("matmul,T", [(None, 120), ("fp32", 120), (None, 1100)])   # (1100 rows: the 256 x 256 bf16 product kernel, last block partial)
def test_cross_attn_bf16_tables(mot, matmul, T):
    """The production cast (train_gpt.py:1124-1126: nn.Embedding -> bfloat16; the attention weights stay fp32 masters and are cast
    where they are used, lines 277-278): bf16 tables go in, a bf16 result and bf16 table gradients come out.  The attention
    kernels are fp32 -- operands are widened once per call -- and the checker is the float64 oracle on the bf16-VALUED operands
    (the reference's own eager bf16 path rounds every intermediate and cannot run here on the CPU: flex_attention).
    matmul="fp32": every product on the fp32 MFMA; forward within one bf16 step of the oracle's result rounded once.
    matmul=None (what bf16 tables select): q, c_proj and their backward products on the bf16 MFMA, their row operands rounded to
    bf16 where the reference's are bf16 tensors (xq out of norm(), y out of the attention), and norm(k) / lambda v read from bf16
    copies of their tables (bf16 tensors there too).  The oracle is run with exactly those roundings (the normalised token rows
    rounded and handed in as a T-row table, `set_round_kv_bf16`, the attention output taken through an identity c_proj, rounded,
    and projected in float64; a key / value element on a rounding boundary that goes the other way in fp32 than in float64 moves
    the outputs of every token with that byte a little: 95.5 % equal instead of > 97 %): same bar against that; against the un-rounded evaluation: three steps
    of the larger of the output and the outputs' rms, rms error under one step.  Gradients: within 1 % of each tensor's largest entry either way (bf16 gradients of
    the tables carry 2^-8 of rounding, the bf16 products 2^-9 per operand)."""
    from mixture_of_tokenizers_amd.modules import Rotary
    D, bpt, Vt, seed = 256, 8, 512, 9981
    H = D // 128
    tab = gi.synth_ttb(seed + 1, Vt, bpt, "left", mean_valid=4.0)
    toks = gi.fineweb_like_tokens(seed, 1, T, vocab=Vt, eot_p=0.01)
    Et, Eb = orc.bf16_round(gi.normal_table(seed + 2, Vt, D)), orc.bf16_round(gi.normal_table(seed + 3, gi.BYTE_VOCAB, D))
    q_w, kv_w, p_w = (f32(a) for a in gi.cross_weights(seed + 4, D))
    pulled = orc.pull_from_left(orc.tokens_to_bytes(toks, tab.astype(np.float32)), bpt, gi.PAD, gi.EOT)
    g = orc.bf16_round(np.random.RandomState(seed + 5).standard_normal((1, T, D)))
    rq, rk = Rotary(128, T), Rotary(128, T * bpt)
    rot = [rq.cos, rq.sin, rk.cos, rk.sin]
    d64 = lambda a: np.asarray(a, dtype=np.float64)
    used = lambda w: d64(orc.bf16_round(w))                         # `.type_as(x)`: the weights as the bf16 matmuls see them
    lam = float(orc.bf16_round(np.array([0.35]))[0])
    rots = [r.numpy() for r in rot]
    args = (toks[0], pulled[0], None, d64(Et), d64(Eb), used(q_w), used(kv_w), used(p_w), lam, *rots)
    orc.set_eps(2.0 ** -7)   # norm() on bf16 tensors: eps = finfo(bfloat16).eps (what bf16 tables select in cross_attn)
    try:
        ref = orc.cross_attn(*args, bpt=bpt, n_heads=H, dtype=np.float64, head_layout=0)
        refg = orc.cross_attn_bwd(*args, d64(g), bpt=bpt, n_heads=H, norm_tok=True, norm_byte=True, head_layout=0)
    finally:
        orc.set_eps(0.0)
    P = lambda a, dt=None: torch.nn.Parameter(dev(a) if dt is None else dev(a).to(dt))
    pEt, pEb = P(Et, torch.bfloat16), P(Eb, torch.bfloat16)
    pq, pkv, pp, plam = P(q_w), P(kv_w), P(p_w), torch.nn.Parameter(torch.tensor(0.35, device=DEV))
    x = mot.functional.cross_attn(dev(toks), dev(pulled), pEt, pEb, q_w=pq, kv_w=pkv, proj_w=pp, lambda_factor=plam,
                                  cos_q=rot[0].to(DEV), sin_q=rot[1].to(DEV), cos_k=rot[2].to(DEV), sin_k=rot[3].to(DEV), bpt=bpt, n_heads=H,
                                  matmul=matmul)
    assert x.dtype == torch.bfloat16 and x.shape == (1, T, D)
    got = host(x.float())[0].astype(np.float64)
    steps = lambda want: np.abs(got - want) / (2.0 ** -7 * np.maximum(np.abs(want), 2.0 ** -6))
    want = orc.bf16_round(ref)
    if matmul == "fp32":
        assert (steps(want) <= 1).all() and (got == want).mean() > 0.97
    else:
        rows = d64(Et)[toks[0]]
        xq = d64(orc.bf16_round(rows / np.sqrt((rows * rows).mean(axis=1, keepdims=True) + 2.0 ** -7)))
        orc.set_eps(2.0 ** -7)
        orc.set_round_kv_bf16(True)   # norm(k) and lambda v: bf16 tensors in the reference, read as such by the attention kernels of the bf16 route
        try:
            y = orc.cross_attn(np.arange(T), pulled[0], None, xq, d64(Eb), used(q_w), used(kv_w), np.eye(D), lam, *rots, bpt=bpt, n_heads=H,
                               dtype=np.float64, head_layout=0, norm_tok=False)
        finally:
            orc.set_eps(0.0)
            orc.set_round_kv_bf16(False)
        emul = orc.bf16_round(d64(orc.bf16_round(y)) @ used(p_w).T)
        assert (steps(emul) <= 1).all() and (got == emul).mean() > 0.94, (steps(emul).max(), (got == emul).mean())
        # (without the two roundings: they move an output by ~2^-9 of the TYPICAL size of the outputs, whatever its own size)
        assert (np.abs(got - want) <= 3 * 2.0 ** -7 * np.maximum(np.abs(want), np.sqrt((want ** 2).mean()))).all() and np.sqrt((steps(want) ** 2).mean()) < 1.0
    (x.float() * dev(g)).sum().backward()
    mot.check_status()
    assert pEt.grad.dtype == torch.bfloat16 and pEb.grad.dtype == torch.bfloat16 and pq.grad.dtype == torch.float32
    for p, key in ((pEt, "tok_table"), (pEb, "byte_table"), (pq, "q_w"), (pkv, "kv_w"), (pp, "proj_w")):
        assert grel(host(p.grad.float()), refg[key]) < 1e-2, key
