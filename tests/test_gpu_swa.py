"""-m gpu: the sliding-window token <- character attention of the Llama character mixer (inference/inference.py:146-224) with
the RMSNorms in front and the residuals behind it (226-267), through functional.char_swa and through the module mirror.

PARITY UNPINNED BY THE REFERENCE: inference.py logs in to the HF hub at import (line 34) and its rotary embedding comes from
a package that is absent here, so neither a run nor a fixture exists.  Checker: oracle.char_swa, a float64 numpy restatement
line by line (the rotary step from the package's published algorithm).  Bar: 5e-6 of max|ref64|, as for the cross-attention
mixin (fp32 kernels: three dense contractions over `dim`, a softmax and two norms)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from util_gpu import DEV, dev, f32, host

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mot():
    import mixture_of_tokenizers_amd as m
    return m


def case(seed, B, T, c_v, d, H, hd, Vt, Vc):
    rs = np.random.RandomState(seed)
    w = lambda *s: rs.standard_normal(s) / np.sqrt(s[-1])
    return dict(toks=rs.randint(0, Vt, (B, T)).astype(np.int32), cid=rs.randint(0, Vc, (B, T, c_v)).astype(np.int64),
                Et=f32(rs.standard_normal((Vt, d))), Ec=f32(rs.standard_normal((Vc, d))),
                wa=f32(1 + 0.1 * rs.standard_normal(d)), wc=f32(1 + 0.1 * rs.standard_normal(d)),
                wq=f32(w(H * hd, d)), wk=f32(w(H * hd, d)), wv=f32(w(H * hd, d)), wo=f32(w(d, H * hd)))


@pytest.mark.parametrize("B,T,c_v,d,H,hd,window,version,seed", [
    (2, 40, 8, 256, 4, 64, 8, "two_residual", 1),        # the reference's geometry (8 x 8 keys), small dims
    (1, 7, 8, 256, 4, 64, 8, "two_residual", 2),         # a row shorter than the window: every query sees padding keys
    (3, 33, 8, 512, 4, 128, 8, "one_residual", 3),       # head_dim 128
    (2, 50, 5, 128, 2, 64, 6, "no_residual", 4),         # 30 keys: lanes beyond the keys idle
    (1, 300, 8, 2048, 32, 64, 8, "two_residual", 5),     # config-5 dims: Llama-3.2-1B hidden 2048, 32 heads x 64
])
def test_char_swa_vs_oracle(mot, B, T, c_v, d, H, hd, window, version, seed):
    Vt, Vc = 700, 132
    c = case(seed, B, T, c_v, d, H, hd, Vt, Vc)
    lt, lc = 0.8, 1.3
    ref = orc.char_swa(c["toks"], c["cid"], c["Et"], c["Ec"], c["wa"], c["wc"], c["wq"], c["wk"], c["wv"], c["wo"], n_heads=H, head_dim=hd,
                       window=window, norm_eps=1e-5, version=version, lambda_tok=lt, lambda_char=lc)
    g = {k: dev(v) for k, v in c.items()}
    two = version == "two_residual"
    x = mot.functional.char_swa(g["toks"], g["cid"], g["Et"], g["Ec"], attn_norm_w=g["wa"], char_norm_w=g["wc"], wq=g["wq"], wk=g["wk"],
                                wv=g["wv"], wo=g["wo"], n_heads=H, head_dim=hd, window=window, norm_eps=1e-5, version=version,
                                lambda_tok=torch.tensor([lt], device=DEV) if two else None, lambda_char=torch.tensor([lc], device=DEV) if two else None)
    mot.check_status()
    assert x.shape == (B, T, d) and x.dtype == torch.float32
    err = np.abs(host(x).astype(np.float64) - ref).max()
    assert err <= 5e-6 * np.abs(ref).max(), f"max abs err {err:.3e} vs max|ref| {np.abs(ref).max():.3e}"


def test_char_mixer_modules(mot):
    """The module mirror (ModelArgs, RMSNorm, FeedForward, TokenMixByCharBMM[Block], CharMixerFrontEnd): the reference's parameter
    names, h from the fused call, the feed-forward in torch on top; a window never leaves its batch row."""
    from mixture_of_tokenizers_amd import modules as M
    args = M.ModelArgs(version="two_residual", n_heads=4, dim=256, intermediate_dim=512, head_dim=64, norm_eps=1e-5)
    torch.manual_seed(7)
    fe = M.CharMixerFrontEnd(500, 132, args).to(DEV)
    keys = sorted(fe.state_dict())
    assert "char_token_mixer.tok_attention.wq.weight" in keys and "char_token_mixer.char_norm.weight" in keys
    assert "char_token_mixer.lambda_tok" in keys and "char_token_mixer.feed_forward.w3.weight" in keys and "char_token_mixer.current_step" in keys
    blk = fe.char_token_mixer
    with torch.no_grad():
        blk.lambda_tok.fill_(0.9); blk.lambda_char.fill_(0.4)
        blk.attention_norm.weight.uniform_(0.8, 1.2); blk.char_norm.weight.uniform_(0.8, 1.2)
    rs = np.random.RandomState(8)
    toks, cid = rs.randint(0, 500, (3, 21)).astype(np.int64), rs.randint(0, 132, (3, 21, 8)).astype(np.int64)
    with torch.no_grad():
        h = blk.mix(fe.embed_tokens(dev(toks)), fe.char_embeddings(dev(cid)))
        out = fe(dev(toks), dev(cid))
    p = {k: host(v).astype(np.float64) for k, v in fe.state_dict().items()}
    ta = "char_token_mixer.tok_attention."
    ref = orc.char_swa(toks, cid, p["embed_tokens.weight"], p["char_embeddings.weight"], p["char_token_mixer.attention_norm.weight"],
                       p["char_token_mixer.char_norm.weight"], p[ta + "wq.weight"], p[ta + "wk.weight"], p[ta + "wv.weight"], p[ta + "wo.weight"],
                       n_heads=4, head_dim=64, window=8, norm_eps=1e-5, version="two_residual", lambda_tok=0.9, lambda_char=0.4)
    assert np.abs(host(h).astype(np.float64) - ref).max() <= 5e-6 * np.abs(ref).max()
    with torch.no_grad():
        want = h + blk.feed_forward(blk.ffn_norm(h))
    assert torch.equal(out, want)
    # rows are independent: the same sequence alone gives the same rows
    with torch.no_grad():
        h1 = blk.mix(fe.embed_tokens(dev(toks[1:2])), fe.char_embeddings(dev(cid[1:2])))
    assert torch.allclose(h1[0], h[1], rtol=0, atol=0)
    with pytest.raises(RuntimeError, match="forward-only"):
        blk.mix(fe.embed_tokens(dev(toks)), fe.char_embeddings(dev(cid)))
    bad = cid.copy(); bad[0, 3, 2] = 132
    with torch.no_grad():
        blk.mix(fe.embed_tokens(dev(toks)), fe.char_embeddings(dev(bad)))
    with pytest.raises(IndexError):
        mot.check_status()


def test_char_swa_across_slabs(mot):
    """More tokens than one 65 536-token slab of the launcher: B x T = 3 x 30 000, so the second slab starts in the
    middle of batch row 2 and a window reads across the slab boundary.  The whole output is too big for the numpy oracle; the
    rows around every slab and batch-row boundary are checked instead: a stretch of a batch row, run as its own one-row input,
    must give the same values (a window never leaves its row and the kernel applies no position-dependent rotation), and that
    stretch is held to the float64 oracle."""
    B, T, c_v, d, H, hd, window = 3, 30000, 8, 128, 2, 64, 8
    Vt, Vc = 700, 132
    c = case(11, B, T, c_v, d, H, hd, Vt, Vc)
    g = {k: dev(v) for k, v in c.items()}
    lt, lc = torch.tensor([0.8], device=DEV), torch.tensor([1.3], device=DEV)
    kw = dict(attn_norm_w=g["wa"], char_norm_w=g["wc"], wq=g["wq"], wk=g["wk"], wv=g["wv"], wo=g["wo"], n_heads=H, head_dim=hd, window=window,
              norm_eps=1e-5, version="two_residual", lambda_tok=lt, lambda_char=lc)
    x = mot.functional.char_swa(g["toks"], g["cid"], g["Et"], g["Ec"], **kw)
    mot.check_status()
    assert x.shape == (B, T, d)
    xf = host(x)
    slab = 65536
    cut = slab - 2 * T                                   # position inside batch row 2 where the second slab starts
    assert 0 < cut < T
    for row, lo, hi in ((2, cut - 200, cut + 200), (0, 0, 300), (1, T - 300, T), (2, T - 300, T)):
        sub_t, sub_c = g["toks"][row:row + 1, lo:hi].contiguous(), g["cid"][row:row + 1, lo:hi].contiguous()
        xs = host(mot.functional.char_swa(sub_t, sub_c, g["Et"], g["Ec"], **kw))[0]
        skip = 0 if lo == 0 else window - 1              # the stretch's first window - 1 tokens lack their left context
        a, b_ = xf[row, lo + skip:hi], xs[skip:]
        assert np.abs(a - b_).max() <= 1e-6 * np.abs(b_).max(), (row, lo, hi, np.abs(a - b_).max())
        ref = orc.char_swa(c["toks"][row:row + 1, lo:hi], c["cid"][row:row + 1, lo:hi], c["Et"], c["Ec"], c["wa"], c["wc"], c["wq"], c["wk"],
                           c["wv"], c["wo"], n_heads=H, head_dim=hd, window=window, norm_eps=1e-5, version="two_residual",
                           lambda_tok=0.8, lambda_char=1.3)[0]
        err = np.abs(xs[skip:].astype(np.float64) - ref[skip:]).max()
        assert err <= 5e-6 * np.abs(ref).max(), (row, lo, hi, err)


@pytest.mark.parametrize("matmul,T", [(None, 60), ("fp32", 60), (None, 330)])   # (2 x 330 rows: the 256 x 256 bf16 product kernel)
def test_char_swa_bf16_tables(mot, matmul, T):
    """bf16 tables and weights: operands widened once, one rounding of the result -- held to the float64 oracle evaluated on the
    bf16 VALUES, within one bf16 step (parity unpinned, as everything of this file).  matmul="fp32": all of it on the fp32 kernels.
    matmul=None (what bf16 tables select): the two products over the tokens on the bf16 MFMA, their row operands rounded to bf16, and
    the projected keys / values kept in bf16 (as the reference's bf16 cast has them) -- the oracle rounds the same four tensors
    (round_token_products_bf16): > 99.5 % within one step, all within 1.5 steps of the larger of the output and the outputs' rms;
    against the oracle without those roundings: three such steps."""
    B, c_v, d, H, hd, window = 2, 8, 256, 4, 64, 8
    c = case(21, B, T, c_v, d, H, hd, 700, 132)
    c16 = {k: (orc.bf16_round(v) if v.dtype == np.float32 else v) for k, v in c.items()}
    lt, lc = (float(np.asarray(orc.bf16_round(np.float32(v))).reshape(-1)[0]) for v in (0.8, 1.3))
    oracle = lambda **kw: orc.char_swa(c16["toks"], c16["cid"], c16["Et"], c16["Ec"], c16["wa"], c16["wc"], c16["wq"], c16["wk"], c16["wv"], c16["wo"],
                                       n_heads=H, head_dim=hd, window=window, norm_eps=1e-5, version="two_residual", lambda_tok=lt, lambda_char=lc, **kw)
    b16 = lambda a: dev(a).bfloat16()
    x = mot.functional.char_swa(dev(c["toks"]), dev(c["cid"]), b16(c16["Et"]), b16(c16["Ec"]), attn_norm_w=b16(c16["wa"]), char_norm_w=b16(c16["wc"]),
                                wq=b16(c16["wq"]), wk=b16(c16["wk"]), wv=b16(c16["wv"]), wo=b16(c16["wo"]), n_heads=H, head_dim=hd, window=window,
                                norm_eps=1e-5, version="two_residual", lambda_tok=torch.tensor([0.8], device=DEV).bfloat16(),
                                lambda_char=torch.tensor([1.3], device=DEV).bfloat16(), matmul=matmul)
    assert x.dtype == torch.bfloat16 and x.shape == (B, T, d)
    got = host(x.float()).astype(np.float64)
    steps = lambda want: np.abs(got - want) / (2.0 ** -8 * np.maximum(np.abs(want), 2.0 ** -6))      # bf16 steps (2^-8 relative), floor near zero
    if matmul == "fp32":
        assert (steps(oracle()) <= 1).all()
    else:
        # (an element of xn or y that sits on a bf16 rounding boundary can round the other way in fp32 than in float64 and moves an
        #  output by |w| 2^-8 |y|: a second step for a handful of outputs; an element of a key or value row that does moves every output
        #  whose token attends to that character a little: measured 99.72 % within one step and 1.04 steps at most at 660 tokens)
        #  The output of wo is a bf16 tensor before the residuals are added (as in the reference's bf16 cast): where ITS rounding goes
        #  the other way (a few elements in 10^4), the result is off by one bf16 step of that intermediate, whatever the size of the sum.
        emul = oracle(round_token_products_bf16=True)
        attn = orc.char_swa(c16["toks"], c16["cid"], c16["Et"], c16["Ec"], c16["wa"], c16["wc"], c16["wq"], c16["wk"], c16["wv"], c16["wo"], n_heads=H,
                            head_dim=hd, window=window, norm_eps=1e-5, version="no_residual", round_token_products_bf16=True)
        em = steps(emul)
        allowed = 1.5 * 2.0 ** -8 * np.maximum(np.abs(emul), np.sqrt((emul ** 2).mean())) + 2.0 ** -7 * np.abs(attn)
        assert (np.abs(got - emul) <= allowed).all() and (em <= 1).mean() > 0.995, (em.max(), (em <= 1).mean())
        plain = oracle()   # without the two roundings: they move an output by ~2^-9 of the TYPICAL size of h, whatever its own size
        assert (np.abs(got - plain) <= 3 * 2.0 ** -8 * np.maximum(np.abs(plain), np.sqrt((plain ** 2).mean())) + 2.0 ** -7 * np.abs(attn)).all()


@pytest.mark.parametrize("tables", ["bf16", "fp32"])
def test_char_swa_residual_paths_agree(mot, tables):
    """two_residual: from 16 384 tokens on the residuals are added to the output of wo by the LDS-table MEAN kernel's read-modify-write
    forms (bf16 rows with bf16 tables; fp32 rows, after a plain product, with fp32 tables), below that by the per-token kernel / by the
    product's own C += after the MEAN kernel (what every other test of this file reaches).  One call over 2 x 8192 tokens against the
    same two batch rows in two calls: the same terms in another order -- bf16: equal but for roundings that sit on a boundary; fp32:
    within a few ulps of the sums' largest term."""
    B, T, c_v, d, H, hd, window = 2, 8192, 8, 256, 4, 64, 8
    c = case(31, B, T, c_v, d, H, hd, 900, 132)
    cast = (lambda a: dev(a).bfloat16()) if tables == "bf16" else dev
    t = {k: (cast(v) if v.dtype == np.float32 else dev(v)) for k, v in c.items()}
    lam = lambda x: torch.tensor([x], device=DEV, dtype=torch.bfloat16 if tables == "bf16" else torch.float32)
    kw = dict(attn_norm_w=t["wa"], char_norm_w=t["wc"], wq=t["wq"], wk=t["wk"], wv=t["wv"], wo=t["wo"], n_heads=H, head_dim=hd, window=window,
              version="two_residual", lambda_tok=lam(0.8), lambda_char=lam(1.3))
    whole = mot.functional.char_swa(t["toks"], t["cid"], t["Et"], t["Ec"], **kw)
    rows = torch.cat([mot.functional.char_swa(t["toks"][i:i + 1], t["cid"][i:i + 1], t["Et"], t["Ec"], **kw) for i in range(B)])
    mot.check_status()
    assert whole.shape == (B, T, d)
    a, b = host(whole.float()).astype(np.float64), host(rows.float()).astype(np.float64)
    if tables == "bf16":
        assert whole.dtype == torch.bfloat16
        assert (np.abs(a - b) <= 2.0 ** -7 * np.maximum(np.abs(b), 2.0 ** -6)).all() and (a == b).mean() > 0.999, (np.abs(a - b).max(), (a == b).mean())
    else:
        assert whole.dtype == torch.float32 and (np.abs(a - b) <= 4e-6 * np.maximum(np.abs(b), 1.0)).all(), np.abs(a - b).max()


@pytest.mark.parametrize("tables", ["fp32", "bf16"])
def test_char_swa_is_capturable_in_a_hip_graph(mot, tables):
    """mot_char_swa_fwd enqueues everything on the caller's stream without a sync -- the sliced few-row products and their ordered sums,
    the bf16 routes' narrowing / widening passes, the attention core, the residual kernels: capture, change the inputs in place,
    replay, compare with an eager call on the changed inputs (bit for bit: no atomics on this path)."""
    B, T, c_v, d, H, hd, window = 2, 90, 8, 256, 4, 64, 8
    c = case(41, B, T, c_v, d, H, hd, 600, 132)
    cast = (lambda a: dev(a).bfloat16()) if tables == "bf16" else dev
    t = {k: (cast(v) if v.dtype == np.float32 else dev(v)) for k, v in c.items()}
    lam = lambda x: torch.tensor([x], device=DEV, dtype=torch.bfloat16 if tables == "bf16" else torch.float32)
    kw = dict(attn_norm_w=t["wa"], char_norm_w=t["wc"], wq=t["wq"], wk=t["wk"], wv=t["wv"], wo=t["wo"], n_heads=H, head_dim=hd, window=window,
              version="two_residual", lambda_tok=lam(0.8), lambda_char=lam(1.3), matmul=None if tables == "bf16" else "bf16")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        mot.functional.char_swa(t["toks"], t["cid"], t["Et"], t["Ec"], **kw)      # warm-up: allocates the workspace
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph, stream=s):
        out = mot.functional.char_swa(t["toks"], t["cid"], t["Et"], t["Ec"], **kw)
    rs = np.random.RandomState(42)
    t["toks"].copy_(dev(rs.randint(0, 600, (B, T)).astype(np.int32)))
    t["cid"].copy_(dev(rs.randint(0, 132, (B, T, c_v)).astype(np.int64)))
    t["wk"].mul_(1.25)
    graph.replay()
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = mot.functional.char_swa(t["toks"], t["cid"], t["Et"], t["Ec"], **kw)
    mot.check_status()
    assert out.dtype == (torch.bfloat16 if tables == "bf16" else torch.float32) and torch.equal(out, ref)


def test_char_swa_kv_cache_reuses_and_refreshes(mot):
    """`kv_cache`: the per-character key / value tables are built once and reused while char_table, char_norm_w, wk, wv are unchanged
    (same result bit for bit, with the cached tables); an in-place change of any of them rebuilds the tables."""
    B, T, c_v, d, H, hd, window = 2, 40, 8, 256, 4, 64, 8
    c = case(23, B, T, c_v, d, H, hd, 500, 132)
    t = {k: dev(v) for k, v in c.items()}
    kw = dict(attn_norm_w=t["wa"], char_norm_w=t["wc"], wq=t["wq"], wk=t["wk"], wv=t["wv"], wo=t["wo"], n_heads=H, head_dim=hd, window=window,
              version="no_residual")
    cache = {}
    ref = mot.functional.char_swa(t["toks"], t["cid"], t["Et"], t["Ec"], **kw)
    a = mot.functional.char_swa(t["toks"], t["cid"], t["Et"], t["Ec"], kv_cache=cache, **kw)
    key = cache["key"]
    b = mot.functional.char_swa(t["toks"], t["cid"], t["Et"], t["Ec"], kv_cache=cache, **kw)      # tables reused
    assert torch.equal(a, ref) and torch.equal(b, ref) and cache["key"] == key
    t["wk"].mul_(1.5)                                                                              # in place: version changes
    ref2 = mot.functional.char_swa(t["toks"], t["cid"], t["Et"], t["Ec"], **kw)
    c2 = mot.functional.char_swa(t["toks"], t["cid"], t["Et"], t["Ec"], kv_cache=cache, **kw)
    assert cache["key"] != key and torch.equal(c2, ref2) and not torch.equal(c2, ref)
