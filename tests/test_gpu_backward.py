"""-m gpu: backward of the fused front-end (SUM / NOOP family) vs the gradients autograd left on the
reference modules (tests/golden/grads.npz) and vs the float64 oracle.

Tolerance: gradients are sums of up to thousands of fp32 contributions added with float atomics
(order-dependent, as in the reference's own GPU embedding backward), so the bar is relative to the
largest entry of each gradient tensor:  max|hip - ref64| <= 2e-5 * max|ref64|  (the reference's own
fp32 CPU gradients sit at ~1e-6..1e-5 of that scale from its float64 gradients)."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import oracle as orc
from util_gpu import DEV, dev, f32, host, rel

pytestmark = pytest.mark.gpu
G = gi.GOLDEN_DIR
TOL = 2e-5


@pytest.fixture(scope="module")
def mot():
    import mixture_of_tokenizers_amd as m
    return m


SUM_SMALL = ("small", 97, 64, 8, 8, 40, 501)


@pytest.mark.parametrize("variant", ["r71", "r71041", "r71081"])
@pytest.mark.parametrize("source", ["ttb", "given"])
def test_sum_backward_vs_reference_autograd(mot, variant, source):
    """loss.backward() through SumFrontEnd == what autograd gives for the reference's run-71 family."""
    from mixture_of_tokenizers_amd.modules import SumFrontEnd
    name, Vt, D, Db, bpt, T, seed = SUM_SMALL
    z = np.load(G / "grads.npz")
    toks = gi.edge_tokens(seed, 1, T, Vt, eot_p=0.08)
    tab = gi.synth_ttb(seed + 1000, Vt, bpt, "left")
    fe = SumFrontEnd(Vt, gi.BYTE_VOCAB, D, Db, bpt, variant=variant[1:], ttb=dev(tab)).to(DEV)
    with torch.no_grad():
        fe.embed_tokens.weight.copy_(dev(f32(gi.normal_table(seed + 1, Vt, D))))
        fe.embed_bytes.weight.copy_(dev(f32(gi.normal_table(seed + 2, gi.BYTE_VOCAB, Db))))
        if fe.scalars is not None:
            fe.scalars.copy_(torch.tensor([0.75, 1.25], device=DEV))
    if source == "ttb":
        x = fe(dev(toks)[0])
    else:
        pulled = orc.pull_from_left(orc.tokens_to_bytes(toks, tab.astype(np.float32)), bpt, gi.PAD, gi.EOT)
        x = fe(dev(toks)[0], dev(pulled))
    assert x.requires_grad
    (x * dev(f32(z["sum/g"]))).sum().backward()                  # train_gpt.py:1319 / main.py:304
    assert rel(host(fe.embed_tokens.weight.grad), z[f"sum/{variant}/f64/d_tok"]) < TOL
    assert rel(host(fe.embed_bytes.weight.grad), z[f"sum/{variant}/f64/d_byte"]) < TOL
    if fe.scalars is not None:
        assert rel(host(fe.scalars.grad), z[f"sum/{variant}/f64/d_scalars"]) < TOL
    # second backward accumulates into .grad like any torch parameter
    (fe(dev(toks)[0]) * dev(f32(z["sum/g"]))).sum().backward()
    assert rel(host(fe.embed_tokens.weight.grad), 2 * z[f"sum/{variant}/f64/d_tok"]) < TOL


def test_noop_backward_vs_reference_autograd(mot):
    """FlexibleEmbedding tokens-only mode + ByteMixinNoop (train_gpt.py:342-348, 421-427) under autograd."""
    from mixture_of_tokenizers_amd import modules as M
    name, Vt, Dt, Db, Dm, bpt, B, T, seed = ("small", 97, 32, 8, 64, 8, 2, 16, 401)
    z = np.load(G / "grads.npz")
    bp = M.ByteHyperparameters(bytes_per_token=bpt, vocab_size=gi.BYTE_VOCAB, byte_mixin_method="noop")
    dims = M.ModelDims(model_dim=Dt, byte_dim=Db, token_dim=Dt)
    embed, mixin = M.FlexibleEmbedding(dims, Vt, bp).to(DEV), M.ByteMixin(dims, T, bp).to(DEV)
    with torch.no_grad():
        embed.embed_tokens.weight.copy_(dev(f32(gi.normal_table(seed + 1, Vt, Dt))))
    toks = dev(gi.edge_tokens(seed, B, T, Vt, eot_p=0.08))
    x = mixin(*embed(tokens=toks, byte_tensor=None, byte_tensor_pulled=None))
    (x * dev(f32(z["scaled/g_noop"]))).sum().backward()
    assert rel(host(embed.embed_tokens.weight.grad), z["scaled/noop/f64/d_tok"]) < TOL


SCALED_SMALL = ("small", 97, 32, 8, 64, 8, 2, 16, 401)


@pytest.mark.parametrize("mode", ["padded", "pulled", "padded_and_pulled"])
def test_concat_backward_vs_reference_autograd(mot, mode):
    """loss.backward() through FlexibleEmbedding + ByteMixin(concat) (train_gpt.py:605-606, 1319) against the
    gradients autograd produced for the reference modules: embed tables, mixin weight."""
    from mixture_of_tokenizers_amd import modules as M
    name, Vt, Dt, Db, Dm, bpt, B, T, seed = SCALED_SMALL
    z, zf = np.load(G / "grads.npz"), np.load(G / "float_scaled.npz")
    bp = M.ByteHyperparameters(bytes_per_token=bpt, vocab_size=gi.BYTE_VOCAB, byte_mixin_method="concat", pull_in=mode != "padded",
                               add_padded_and_pulled=mode == "padded_and_pulled")
    dims = M.ModelDims(model_dim=Dm, byte_dim=Db, token_dim=Dt)
    embed, mixin = M.FlexibleEmbedding(dims, Vt, bp).to(DEV), M.ByteMixin(dims, T, bp).to(DEV)
    with torch.no_grad():
        embed.embed_tokens.weight.copy_(dev(f32(gi.normal_table(seed + 1, Vt, Dt))))
        embed.embed_bytes.weight.copy_(dev(f32(gi.normal_table(seed + 2, gi.BYTE_VOCAB, Db))))
        mixin.mixin.mixin.weight.copy_(dev(f32(gi.casted_linear_weight(seed + 3, Dm, Dt + bpt * Db))))
    x = mixin(*embed(tokens=dev(zf[f"{name}/tokens"]), byte_tensor=dev(zf[f"{name}/padded"]), byte_tensor_pulled=dev(zf[f"{name}/pulled"])))
    assert x.requires_grad
    (x * dev(f32(z["scaled/g"]))).sum().backward()
    assert rel(host(embed.embed_tokens.weight.grad), z[f"scaled/{mode}/f64/d_tok"]) < TOL
    assert rel(host(embed.embed_bytes.weight.grad), z[f"scaled/{mode}/f64/d_byte"]) < TOL
    assert rel(host(mixin.mixin.mixin.weight.grad), z[f"scaled/{mode}/f64/d_W"]) < TOL


def test_mathblations_backward_vs_reference_autograd(mot):
    """DigitFrontEnd (wte tied to an lm_head Parameter, dte, fc with bias, digits first) under autograd:
    the gradient lands on the shared Parameter exactly as in model.py:316-317."""
    from mixture_of_tokenizers_amd import modules as M
    z = np.load(G / "grads.npz")
    D = 32
    fe = M.DigitFrontEnd(M.GPTConfig(vocab_size=1003, n_embd_tok=D, n_embd_digit=D, length_factor=3, digit_mixin_method="concat")).to(DEV)
    lm_head = torch.nn.Linear(D, 1003, bias=False).to(DEV)
    fe.wte.weight = lm_head.weight
    Wf, bf = gi.linear_weight_bias(613, D, 4 * D)
    with torch.no_grad():
        lm_head.weight.copy_(dev(f32(gi.normal_table(611, 1003, D))))
        fe.dte.weight.copy_(dev(f32(gi.normal_table(612, 14, D))))
        fe.digit_mixin.fc.weight.copy_(dev(f32(Wf))); fe.digit_mixin.fc.bias.copy_(dev(f32(bf)))
    x = fe(dev(z["math/x_tokens"]), dev(z["math/x_digit_tokens"]))
    (x * dev(f32(z["math/g"]))).sum().backward()
    assert rel(host(lm_head.weight.grad), z["math/f64/d_tok"]) < TOL
    assert rel(host(fe.dte.weight.grad), z["math/f64/d_byte"]) < TOL
    assert rel(host(fe.digit_mixin.fc.weight.grad), z["math/f64/d_W"]) < TOL
    assert rel(host(fe.digit_mixin.fc.bias.grad), z["math/f64/d_bias"]) < TOL


@pytest.mark.parametrize("Dt,Db,bpt,Dm,Vt,B,T,kw,seed", [
    (256, 32, 16, 768, 4096, 2, 512, dict(norm_tok=True, norm_byte=True, norm_out=True), 9201),     # C2-CONCAT dims
    (256, 48, 16, 1024, 2048, 2, 160, dict(norm_tok=True, norm_byte=True, norm_out=True), 9202),    # production dims
    (128, 32, 8, 200, 512, 2, 77, dict(norm_tok=True, bias=True), 9203),                            # ragged Dm, bias, no out norm
    (256, 256, 3, 256, 1003, 8, 32, dict(bias=True, bytes_first=True), 9204),                       # mathblations dims
    (100, 20, 5, 384, 512, 2, 130, dict(norm_byte=True, norm_out=True, bytes_first=True, scaled=True), 9205),
    (256, 48, 16, 1024, 2048, 2, 160, dict(norm_tok=True, norm_byte=True, norm_out=True, dual=True), 9206),  # norm(emb(padded)+emb(pulled))
    (64, 24, 7, 96, 300, 3, 50, dict(norm_byte=True, dual=True, scaled=True, bias=True), 9207),               # ragged slots across lanes
    (128, 32, 8, 256, 512, 2, 90, dict(norm_tok=True, dual=True), 9208),                                    # two id tensors, no byte norm
    (768, 768, 3, 768, 1003, 4, 96, dict(bias=True, bytes_first=True), 9209),       # mathblations DEFAULT dims (model.py:21-24): K = 3072, slot-wise scatter
    (256, 256, 5, 512, 600, 2, 70, dict(norm_tok=True, norm_byte=True, norm_out=True), 9210),   # wide rows with every norm (K = 1536)
    # the reference's dimension sweeps (experiments100_000steps.sh: model 1024, bpt 16, token 768 / 896 / 1024 x byte 64 / 128)
    (768, 64, 16, 1024, 900, 2, 48, dict(norm_tok=True, norm_byte=True, norm_out=True), 9211),    # K = 1792: token part + one block of 16 slots
    (1024, 128, 16, 1024, 900, 2, 40, dict(norm_tok=True, norm_byte=True, norm_out=True), 9212),  # K = 3072: two blocks of 8 slots
    (896, 64, 16, 1024, 900, 2, 40, dict(norm_tok=True, norm_byte=True, norm_out=True), 9213),    # K = 1920: 896 is not 256 n -> the strided kernel
    (896, 128, 16, 1024, 900, 2, 40, dict(norm_tok=True, norm_byte=True, norm_out=True), 9214),   # K = 2944: token part on the general kernel, two slot blocks
])
def test_concat_backward_vs_oracle(mot, Dt, Db, bpt, Dm, Vt, B, T, kw, seed):
    kw = dict(kw)
    scaled, use_bias, dual = kw.pop("scaled", False), kw.pop("bias", False), kw.pop("dual", False)
    tab = gi.synth_ttb(seed + 1, Vt, bpt, "left", mean_valid=min(4.4, bpt / 2))
    toks = gi.fineweb_like_tokens(seed, B, T, vocab=Vt, eot_p=0.01)
    Et, Eb = f32(gi.normal_table(seed + 2, Vt, Dt)), f32(gi.normal_table(seed + 3, gi.BYTE_VOCAB, Db))
    K = Dt + bpt * Db
    W = f32(gi.casted_linear_weight(seed + 4, Dm, K))
    bias = f32(gi.linear_weight_bias(seed + 5, Dm, K)[1]) if use_bias else None
    g = f32(np.random.RandomState(seed + 6).standard_normal((B, T, Dm)))
    padded = orc.tokens_to_bytes(toks, tab.astype(np.float32))
    pulled = orc.pull_from_left(padded, bpt, gi.PAD, gi.EOT)
    okw = dict(kw)
    dEt, dEb, dW = (torch.nn.Parameter(dev(a)) for a in (Et, Eb, W))
    dbias = torch.nn.Parameter(dev(bias)) if use_bias else None
    gkw = dict(kw)
    if scaled:
        okw.update(scale_tok=1.3, scale_byte=0.6)
        st, sb = torch.nn.Parameter(torch.tensor([1.3], device=DEV)), torch.nn.Parameter(torch.tensor([0.6], device=DEV))
        gkw.update(scale_tok=st, scale_byte=sb)
    if dual:
        gkw.update(add_padded=True)
    ref = orc.embed_mix_bwd(toks, pulled, padded if dual else None, Et.astype(np.float64), Eb.astype(np.float64), g.astype(np.float64),
                            mode="concat_linear", bpt=bpt, weight=W.astype(np.float64),
                            bias=None if bias is None else bias.astype(np.float64), dtype=np.float64, **okw)
    x = mot.embed_mix(dev(toks), dEt, dEb, mode="concat_linear", bpt=bpt, ttb=dev(tab), pull="left", weight=dW, bias=dbias, **gkw)
    (x * dev(g)).sum().backward()
    assert rel(host(dEt.grad), ref["tok_table"]) < TOL
    assert rel(host(dEb.grad), ref["byte_table"]) < TOL
    assert rel(host(dW.grad), ref["weight"]) < TOL
    if use_bias:
        assert rel(host(dbias.grad), ref["bias"]) < TOL
    if scaled:
        assert abs(float(st.grad) - ref["scales"][0]) < TOL * abs(ref["scales"]).max()
        assert abs(float(sb.grad) - ref["scales"][1]) < TOL * abs(ref["scales"]).max()


def test_concat_backward_across_forward_slabs(mot):
    """More rows than one slab of the composed forward (65 536): the per-row factors the forward keeps for the backward
    (out_row_rnorm) and the saved output come from two slabs; gradients against the float64 oracle."""
    Dt, Db, bpt, Dm, Vt, B, T, seed = 64, 16, 4, 96, 700, 1, 70001, 9251
    rs = np.random.RandomState(seed)
    toks = rs.randint(0, Vt, (B, T)).astype(np.int32)
    ids = rs.randint(0, gi.BYTE_VOCAB, (B, T * bpt)).astype(np.int64)
    Et, Eb = f32(gi.normal_table(seed + 2, Vt, Dt)), f32(gi.normal_table(seed + 3, gi.BYTE_VOCAB, Db))
    K = Dt + bpt * Db
    W = f32(gi.casted_linear_weight(seed + 4, Dm, K))
    g = f32(rs.standard_normal((B, T, Dm)))
    kw = dict(norm_tok=True, norm_byte=True, norm_out=True)
    ref = orc.embed_mix_bwd(toks, ids, None, Et.astype(np.float64), Eb.astype(np.float64), g.astype(np.float64), mode="concat_linear", bpt=bpt,
                            weight=W.astype(np.float64), dtype=np.float64, **kw)
    dEt, dEb, dW = (torch.nn.Parameter(dev(a)) for a in (Et, Eb, W))
    x = mot.embed_mix(dev(toks), dEt, dEb, mode="concat_linear", bpt=bpt, ids_a=dev(ids), weight=dW, **kw)
    (x * dev(g)).sum().backward()
    mot.check_status()
    assert rel(host(dEt.grad), ref["tok_table"]) < TOL
    assert rel(host(dEb.grad), ref["byte_table"]) < TOL
    assert rel(host(dW.grad), ref["weight"]) < TOL


@pytest.mark.parametrize("D,Db,bpt,Vt,B,T,kw,seed", [
    (768, 48, 16, 4096, 4, 512, dict(norm_out=True), 9101),                                   # headline dims
    (768, 48, 16, 4096, 2, 300, dict(norm_tok=True, norm_byte=True, norm_out=True, scaled=True), 9102),
    (256, 32, 8, 512, 3, 200, dict(), 9103),                                                  # no norms at all
    (1024, 64, 16, 512, 2, 130, dict(norm_tok=True, norm_byte=True, scaled=True), 9104),      # 71081 shape
    (240, 12, 20, 300, 2, 100, dict(norm_out=True), 9105),                                    # D % 64 != 0
    (2048, 128, 16, 512, 1, 70, dict(norm_out=True), 9106),
])
def test_sum_backward_vs_oracle(mot, D, Db, bpt, Vt, B, T, kw, seed):
    kw = dict(kw)
    scaled = kw.pop("scaled", False)
    tab = gi.synth_ttb(seed + 1, Vt, bpt, "left")
    toks = gi.fineweb_like_tokens(seed, B, T, vocab=Vt, eot_p=0.01)
    Et, Eb = f32(gi.normal_table(seed + 2, Vt, D)), f32(gi.normal_table(seed + 3, gi.BYTE_VOCAB, Db))
    g = f32(np.random.RandomState(seed + 4).standard_normal((B, T, D)))
    padded = orc.tokens_to_bytes(toks, tab.astype(np.float32))
    pulled = orc.pull_from_left(padded, bpt, gi.PAD, gi.EOT)
    okw, gkw = dict(kw), dict(kw)
    if scaled:
        okw.update(scale_tok=1.3, scale_byte=0.6)
        gkw.update(scale_tok=torch.tensor([1.3], device=DEV), scale_byte=torch.tensor([0.6], device=DEV))
    ref = orc.embed_mix_bwd(toks, pulled, None, Et.astype(np.float64), Eb.astype(np.float64), g.astype(np.float64),
                            mode="sum", bpt=bpt, dtype=np.float64, **okw)
    got = mot.functional.embed_mix_backward(dev(g), dev(toks), dev(Et), dev(Eb), mode="sum", bpt=bpt, ids_a=dev(pulled), **gkw)
    assert rel(host(got["tok_table"]), ref["tok_table"]) < TOL
    assert rel(host(got["byte_table"]), ref["byte_table"]) < TOL
    if scaled:
        assert abs(float(got["scale_tok"]) - ref["scales"][0]) < TOL * abs(ref["scales"]).max()
        assert abs(float(got["scale_byte"]) - ref["scales"][1]) < TOL * abs(ref["scales"]).max()
    # rows no token touched get exactly zero gradient
    untouched = np.setdiff1d(np.arange(Vt), np.unique(toks))
    assert not host(got["tok_table"])[untouched].any()
    if not kw.get("norm_byte"):      # two id tensors: emb(padded) + emb(pulled)
        ref2 = orc.embed_mix_bwd(toks, pulled, padded, Et.astype(np.float64), Eb.astype(np.float64), g.astype(np.float64),
                                 mode="sum", bpt=bpt, dtype=np.float64, **okw)
        got2 = mot.functional.embed_mix_backward(dev(g), dev(toks), dev(Et), dev(Eb), mode="sum", bpt=bpt, ids_a=dev(pulled),
                                                 ids_b=dev(padded), **gkw)
        assert rel(host(got2["byte_table"]), ref2["byte_table"]) < TOL


def _scalar_scale(ref_scales, n_tokens, D):
    """Scale for the gradients of the learned scalars, d s = sum over n_tokens * D products of O(1) factors with random signs.
    With the output norm the two sums are equal and opposite and can cancel to (almost) nothing, so a bar relative to the
    result alone is ill-conditioned: fp32 summation error is relative to the terms, whose natural scale is sqrt(n_tokens * D)."""
    return max(float(np.abs(ref_scales).max()), float(np.sqrt(n_tokens * D)))


def _bucket_case(seed, Vt=512, D=128, Db=16, bpt=8, B=4, T=96):
    """SumFrontEnd 71041 (both pre-norms, learned scalars, output norm) with SEEDED weights, its gradients bound to a
    GradBucket; returns the module, the bucket, the inputs and the float64 oracle gradients of the whole batch."""
    from mixture_of_tokenizers_amd.grad_sync import GradBucket
    from mixture_of_tokenizers_amd.modules import SumFrontEnd
    tab = gi.synth_ttb(seed + 1, Vt, bpt, "left")
    toks = gi.fineweb_like_tokens(seed, B, T, vocab=Vt, eot_p=0.02)
    g = f32(np.random.RandomState(seed + 2).standard_normal((B, T, D)))
    Et, Eb = f32(gi.normal_table(seed + 3, Vt, D)), f32(gi.normal_table(seed + 4, gi.BYTE_VOCAB, Db))
    sc = f32(np.random.RandomState(seed + 5).uniform(0.5, 1.5, 2))       # [-2] bytes, [-1] tokens
    fe = SumFrontEnd(Vt, gi.BYTE_VOCAB, D, Db, bpt, variant="71041", ttb=dev(tab)).to(DEV)
    with torch.no_grad():    # nn.Embedding's own init draws from torch's global generator: never leave a parity test on that
        fe.embed_tokens.weight.copy_(dev(Et)); fe.embed_bytes.weight.copy_(dev(Eb)); fe.scalars.copy_(dev(sc))
    bucket = GradBucket(list(fe.parameters()), in_place=True)
    pulled = orc.pull_from_left(orc.tokens_to_bytes(toks, tab.astype(np.float32)), bpt, gi.PAD, gi.EOT)
    ref = orc.embed_mix_bwd(toks, pulled, None, Et.astype(np.float64), Eb.astype(np.float64), g.astype(np.float64), mode="sum", bpt=bpt,
                            dtype=np.float64, norm_tok=True, norm_byte=True, norm_out=True, scale_tok=float(sc[1]), scale_byte=float(sc[0]))
    return fe, bucket, toks, g, ref


def _check_bucket_grads(fe, ref, n_tokens, D, tol, what):
    assert rel(host(fe.embed_tokens.weight.grad), ref["tok_table"]) < tol, what
    assert rel(host(fe.embed_bytes.weight.grad), ref["byte_table"]) < tol, what
    got = host(fe.scalars.grad).astype(np.float64)          # [d scale_byte, d scale_tok]
    want = np.array([ref["scales"][1], ref["scales"][0]])
    bar = tol * _scalar_scale(want, n_tokens, D)
    assert np.abs(got - want).max() < bar, f"{what}: scalars.grad {got} vs float64 {want}: |difference| {np.abs(got - want).max():.3e} >= {bar:.3e}"


def test_grad_bucket_accumulates_in_place(mot):
    """GradBucket (grad_sync.py): the front-end's .grad tensors are views of one flat buffer, the backward kernel adds into
    them in place (no temporary table-sized gradient), and two micro-batches accumulate to the gradient of their union
    (train_gpt.py:1319-1321).  Every gradient is held to the float64 oracle -- after the two micro-batches and after the
    whole batch -- so a failure names the tensor, the element and the pass."""
    fe, bucket, toks, g, ref = _bucket_case(9300)
    B, T, D = g.shape
    ptrs = [p.grad.data_ptr() for p in bucket.params]
    flat_ptr = bucket.flat.data_ptr()
    for rows in (slice(0, 2), slice(2, 4)):
        (fe(dev(toks[rows])) * dev(g[rows])).sum().backward()
    assert [p.grad.data_ptr() for p in bucket.params] == ptrs and bucket.flat.data_ptr() == flat_ptr
    assert bucket.all_reduce() is None                      # no process group: nothing to exchange
    _check_bucket_grads(fe, ref, B * T, D, TOL, "two micro-batches accumulated")
    acc = [p.grad.clone() for p in bucket.params]
    bucket.zero_()
    assert all(float(p.grad.abs().max()) == 0.0 for p in bucket.params)
    (fe(dev(toks)) * dev(g)).sum().backward()
    _check_bucket_grads(fe, ref, B * T, D, TOL, "whole batch")
    for a, p, name in zip(acc, bucket.params, ("embed_tokens.weight", "embed_bytes.weight", "scalars")):
        if name == "scalars":
            assert float((a - p.grad).abs().max()) < 2 * TOL * _scalar_scale(host(p.grad), B * T, D), name
        else:
            assert rel(host(a), host(p.grad)) < 2 * TOL, name     # two GPU results, each within TOL of the exact gradient


def test_grad_bucket_seed_sweep(mot):
    """The same accumulation over 48 independently seeded weight / token / gradient draws in ONE process (the workspace, the
    allocator's free blocks and the sort scratch carry whatever the previous draw left): every gradient of every draw against
    float64.  Also records how often the scalars' gradient nearly cancels (|result| < 1 % of sqrt(N D)), the case in which a
    bar relative to the result alone -- what this file used in round 1 -- cannot be met by any fp32 summation."""
    near_cancel = 0
    for k in range(48):
        fe, bucket, toks, g, ref = _bucket_case(977000 + 13 * k, B=4, T=96 + 8 * (k % 5))
        B, T, D = g.shape
        for rows in (slice(0, 1), slice(1, 4)):             # uneven micro-batches
            (fe(dev(toks[rows])) * dev(g[rows])).sum().backward()
        _check_bucket_grads(fe, ref, B * T, D, TOL, f"draw {k}")
        near_cancel += float(np.abs(ref["scales"]).max()) < 0.01 * np.sqrt(B * T * D)
    mot.check_status()
    print(f"draws whose scalar gradients nearly cancel: {near_cancel} of 48")


@pytest.mark.parametrize("B,T,Vt,same", [(1, 1, 5, False), (1, 67, 3, True), (5, 413, 1, True), (3, 1000, 50000, False)])
def test_sum_backward_degenerate_groupings(mot, B, T, Vt, same):
    """The counting sort's corner cases: a single position, every position the same token (one run that crosses every
    wave and workgroup boundary), a one-row table, and a vocabulary far larger than the batch (almost no reuse)."""
    D, Db, bpt = 128, 16, 8
    rs = np.random.RandomState(9470 + T)
    toks = (np.full((B, T), Vt - 1) if same else rs.randint(0, Vt, (B, T))).astype(np.int32)
    ids = rs.randint(0, gi.BYTE_VOCAB, (B, T * bpt)).astype(np.int64)
    Et, Eb = f32(gi.normal_table(9471, Vt, D)), f32(gi.normal_table(9472, gi.BYTE_VOCAB, Db))
    g = f32(rs.standard_normal((B, T, D)))
    kw = dict(mode="sum", bpt=bpt, norm_out=True, norm_tok=True)
    ref = orc.embed_mix_bwd(toks, ids, None, Et.astype(np.float64), Eb.astype(np.float64), g.astype(np.float64), dtype=np.float64, **kw)
    got = mot.functional.embed_mix_backward(dev(g), dev(toks), dev(Et), dev(Eb), ids_a=dev(ids), **kw)
    mot.check_status()
    assert rel(host(got["tok_table"]), ref["tok_table"]) < TOL
    assert rel(host(got["byte_table"]), ref["byte_table"]) < TOL


def test_backward_is_capturable_in_a_hip_graph(mot):
    """The backward enqueues a memset, the three sort kernels and the scatter kernel on the given stream, with no host
    sync and no allocation once the workspace exists: capture it, change the batch in place, replay, compare with eager."""
    D, Db, bpt, Vt, B, T = 256, 16, 16, 1024, 4, 300
    Et, Eb = dev(f32(gi.normal_table(9451, Vt, D))), dev(f32(gi.normal_table(9452, gi.BYTE_VOCAB, Db)))
    rs = np.random.RandomState(9453)
    toks = dev(gi.fineweb_like_tokens(9454, B, T, vocab=Vt, eot_p=0.01))
    ids = dev(rs.randint(0, gi.BYTE_VOCAB, (B, T * bpt)).astype(np.int64))
    g = dev(f32(rs.standard_normal((B, T, D))))
    into = {"tok_table": torch.zeros_like(Et), "byte_table": torch.zeros_like(Eb)}
    kw = dict(mode="sum", bpt=bpt, ids_a=ids, norm_out=True, norm_byte=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        mot.functional.embed_mix_backward(g, toks, Et, Eb, into=into, **kw)     # warm-up: allocates the workspace
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        into["tok_table"].zero_(); into["byte_table"].zero_()
        mot.functional.embed_mix_backward(g, toks, Et, Eb, into=into, **kw)
    toks.copy_(dev(gi.fineweb_like_tokens(9455, B, T, vocab=Vt, eot_p=0.01)))
    ids.copy_(dev(rs.randint(0, gi.BYTE_VOCAB, (B, T * bpt)).astype(np.int64)))
    g.copy_(dev(f32(rs.standard_normal((B, T, D)))))
    graph.replay()
    torch.cuda.synchronize()
    ref = mot.functional.embed_mix_backward(g, toks, Et, Eb, **kw)
    mot.check_status()
    assert rel(host(into["tok_table"]), host(ref["tok_table"])) < 2 * TOL      # two GPU results (atomic order differs)
    assert rel(host(into["byte_table"]), host(ref["byte_table"])) < 2 * TOL


def test_sum_backward_many_chunks(mot):
    """More positions than 256 workgroups x 2048, N = 3 x 200 001 not a multiple of anything convenient (partial last
    sort chunk, uneven shares per workgroup and wave), with a tiny vocabulary: 37 long runs that cross many wave and
    workgroup boundaries of the sorted order."""
    D, Db, bpt, Vt, B, T = 64, 8, 8, 37, 3, 200001
    rs = np.random.RandomState(9401)
    toks = rs.randint(0, Vt, (B, T)).astype(np.int32)
    ids = rs.randint(0, gi.BYTE_VOCAB, (B, T * bpt)).astype(np.int64)
    Et, Eb = f32(gi.normal_table(9402, Vt, D)), f32(gi.normal_table(9403, gi.BYTE_VOCAB, Db))
    g = f32(rs.standard_normal((B, T, D)))
    ref = orc.embed_mix_bwd(toks, ids, None, Et.astype(np.float64), Eb.astype(np.float64), g.astype(np.float64),
                            mode="sum", bpt=bpt, dtype=np.float64, norm_out=True)
    got = mot.functional.embed_mix_backward(dev(g), dev(toks), dev(Et), dev(Eb), mode="sum", bpt=bpt, ids_a=dev(ids), norm_out=True)
    mot.check_status()
    assert rel(host(got["tok_table"]), ref["tok_table"]) < TOL
    assert rel(host(got["byte_table"]), ref["byte_table"]) < TOL


def test_backward_flags_out_of_range_ids(mot):
    D, Db, bpt, Vt, T = 64, 8, 8, 50, 300
    toks = torch.randint(0, Vt, (1, T), dtype=torch.int32, device=DEV)
    ids = torch.randint(0, gi.BYTE_VOCAB, (1, T * bpt), dtype=torch.int64, device=DEV)
    toks[0, 7] = Vt + 5
    ids[0, 11] = 9999
    Et, Eb = torch.randn(Vt, D, device=DEV), torch.randn(gi.BYTE_VOCAB, Db, device=DEV)
    g = torch.randn(1, T, D, device=DEV)
    got = mot.functional.embed_mix_backward(g, toks, Et, Eb, mode="sum", bpt=bpt, ids_a=ids)
    assert bool(torch.isfinite(got["tok_table"]).all())
    with pytest.raises(IndexError):
        mot.check_status()


@pytest.mark.parametrize("D,Vt,B,T,norm_tok,scaled,seed", [
    (256, 512, 3, 700, True, False, 9501),     # lean kernel, NOOP (tokens-only embedding, train_gpt.py:342-348)
    (768, 4096, 2, 300, True, True, 9502),
    (64, 50, 1, 129, False, False, 9503),
    (96, 50, 2, 77, True, False, 9504),        # D % 64 != 0: general kernel
])
def test_noop_backward_vs_oracle(mot, D, Vt, B, T, norm_tok, scaled, seed):
    toks = gi.fineweb_like_tokens(seed, B, T, vocab=Vt, eot_p=0.01)
    Et = f32(gi.normal_table(seed + 1, Vt, D))
    g = f32(np.random.RandomState(seed + 2).standard_normal((B, T, D)))
    okw, gkw = dict(norm_tok=norm_tok), dict(norm_tok=norm_tok)
    if scaled:
        okw.update(scale_tok=1.7)
        gkw.update(scale_tok=torch.tensor([1.7], device=DEV))
    ref = orc.embed_mix_bwd(toks, None, None, Et.astype(np.float64), None, g.astype(np.float64), mode="noop", bpt=0, dtype=np.float64, **okw)
    got = mot.functional.embed_mix_backward(dev(g), dev(toks), dev(Et), None, mode="noop", **gkw)
    assert rel(host(got["tok_table"]), ref["tok_table"]) < TOL
    if scaled:
        assert abs(float(got["scale_tok"]) - ref["scales"][0]) < TOL * max(abs(ref["scales"][0]), 1.0)
    touched = np.zeros(Vt, bool); touched[toks.reshape(-1)] = True
    assert not host(got["tok_table"])[~touched].any()          # rows no token touched get exactly zero gradient


def test_full_size_backward_properties(mot):
    """Config-4 size (256 x 2048 tokens, GPT-2 vocab, d 768, bpt 16), where the oracle is too slow to run whole:
    size-independent properties of the backward -- it is linear in the upstream gradient, rows of tokens that do not
    occur stay exactly zero, the byte-table gradient's column sums equal the column sums of the per-position byte
    gradients (= of dx, folded over the 16 slots), and 16 sampled rows of the token-table gradient match the oracle
    evaluated on the positions of those tokens alone."""
    B, T, Vt, D, Db, bpt = 256, 2048, 50257, 768, 48, 16
    toks_np = gi.fineweb_like_tokens(12345, B, T, vocab=Vt)
    tab = gi.widen_left_pad(gi.load_real_ttb8(), bpt)
    toks = dev(toks_np)
    ids = mot.data_creation.pull_from_left(mot.data_creation.tokens_to_bytes(toks, dev(tab)), bpt, gi.PAD, gi.EOT)
    gen = torch.Generator(device=DEV).manual_seed(7)
    Et = torch.randn((Vt, D), generator=gen, device=DEV)
    Eb = torch.randn((gi.BYTE_VOCAB, Db), generator=gen, device=DEV)
    g1 = torch.randn((B, T, D), generator=gen, device=DEV)
    g2 = torch.randn((B, T, D), generator=gen, device=DEV)
    bw = lambda g: mot.functional.embed_mix_backward(g, toks, Et, Eb, mode="sum", bpt=bpt, ids_a=ids, norm_out=True)
    r1, r2, r12 = bw(g1), bw(g2), bw(g1 + g2)
    mot.check_status()
    for k in ("tok_table", "byte_table"):
        a, b = r12[k], r1[k] + r2[k]
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), k               # linearity
    counts = torch.bincount(toks.view(-1).long(), minlength=Vt)
    assert float(r1["tok_table"][counts == 0].abs().max()) == 0.0                            # untouched rows
    # column sums: sum_rows d_byte[:, wi] == sum over positions and slots of dx[n, k*Db + wi]; dx itself is what a backward
    # with a single-row "token table" view returns summed per token, i.e. the column sums of d_tok folded over the slots
    col_tok = r1["tok_table"].double().sum(0).view(bpt, Db).sum(0)
    col_byte = r1["byte_table"].double().sum(0)
    assert float((col_tok - col_byte).abs().max()) <= 2e-5 * float(col_byte.abs().max())
    # a few token rows against the oracle on just their positions (the gradient of a row only involves its own positions)
    flat = toks_np.reshape(-1)
    ids_np, g_np = host(ids).reshape(-1, bpt), host(g1).reshape(-1, D)
    Et_np, Eb_np = host(Et).astype(np.float64), host(Eb).astype(np.float64)
    for tok in (0, 1, 17, 262, 1000, 5000, 20000, 50255):
        pos = np.flatnonzero(flat == tok)[:64]
        if len(pos) == 0:
            continue
        sub = orc.embed_mix_bwd(flat[pos][None], ids_np[pos].reshape(1, -1), None, Et_np, Eb_np, g_np[pos][None].astype(np.float64),
                                mode="sum", bpt=bpt, dtype=np.float64, norm_out=True)
        if len(np.flatnonzero(flat == tok)) <= 64:      # all of the token's positions were included: its row must match
            assert rel(host(r1["tok_table"][tok]), sub["tok_table"][tok]) < TOL, tok


@pytest.mark.parametrize("D,bpt,Vt,Vc,B,T,kw,seed", [
    (256, 8, 500, 132, 3, 100, dict(scaled=True), 9601),                          # the two_residual residual (inference.py:267): both lambdas
    (128, 8, 300, 132, 2, 77, dict(norm_byte=True, scaled=True), 9602),           # per-character rms norm: its backward needs S = G V^T
    (64, 5, 100, 40, 1, 70001, dict(norm_tok=True, norm_byte=True), 9603),        # more tokens than one slab of the dense products
    (2048, 8, 700, 132, 1, 300, dict(scaled=True), 9604),                         # config-5 width
])
def test_mean_backward_vs_oracle(mot, D, bpt, Vt, Vc, B, T, kw, seed):
    """Backward of the MEAN mix (x = s_t a + s_c mean_k v_k): token-table scatter + the character table's gradient as dense
    products over the token axis.  float64 oracle (parity unpinned by the reference, whose inference file never trains), through
    autograd: embed_mix records the node, loss.backward() fills .grad."""
    kw = dict(kw)
    scaled = kw.pop("scaled", False)
    rs = np.random.RandomState(seed)
    toks = rs.randint(0, Vt, (B, T)).astype(np.int32)
    ids = rs.randint(0, Vc, (B, T * bpt)).astype(np.int64)
    Et, Ec = f32(gi.normal_table(seed + 1, Vt, D)), f32(gi.normal_table(seed + 2, Vc, D))
    g = f32(rs.standard_normal((B, T, D)))
    okw, gkw = dict(kw), dict(kw)
    pt, pc = torch.nn.Parameter(dev(Et)), torch.nn.Parameter(dev(Ec))
    if scaled:
        okw.update(scale_tok=0.8, scale_byte=1.3)
        st, sb = torch.nn.Parameter(torch.tensor([0.8], device=DEV)), torch.nn.Parameter(torch.tensor([1.3], device=DEV))
        gkw.update(scale_tok=st, scale_byte=sb)
    ref = orc.embed_mix_bwd(toks, ids, None, Et.astype(np.float64), Ec.astype(np.float64), g.astype(np.float64), mode="mean", bpt=bpt,
                            dtype=np.float64, **okw)
    x = mot.embed_mix(dev(toks), pt, pc, mode="mean", bpt=bpt, ids_a=dev(ids), **gkw)
    (x * dev(g)).sum().backward()
    mot.check_status()
    assert rel(host(pt.grad), ref["tok_table"]) < TOL
    assert rel(host(pc.grad), ref["byte_table"]) < TOL
    if scaled:
        scale = _scalar_scale(ref["scales"], B * T, D)
        assert abs(float(st.grad) - ref["scales"][0]) < TOL * scale and abs(float(sb.grad) - ref["scales"][1]) < TOL * scale


# ---- the fixed-point scale of the privatised byte-table sums comes from ONE sampled gradient row per wave:
#      the cases below make the sample unrepresentative in both directions, on the two lane-contiguous kernels
#      (norm_out only -> embed_mix_bwd_plain_kernel; with the per-embedding norms -> embed_mix_bwd_lc_kernel).
_FX_KW = [dict(norm_out=True), dict(norm_tok=True, norm_byte=True, norm_out=True)]


@pytest.mark.parametrize("kw", _FX_KW, ids=["plain", "lc"])
@pytest.mark.parametrize("keep", [0.002, 0.03])
def test_sum_backward_sparse_gradient_rows(mot, kw, keep):
    """grad_out is exactly zero on almost every position (masked / padded positions of a real loss): most workgroups sample only
    zero rows.  A scale of 2^0 there would round every |term| < 0.5 of the non-zero positions to nothing."""
    D, Db, bpt, Vt, B, T, seed = 768, 48, 16, 2048, 8, 1024, 9301
    tab = gi.synth_ttb(seed + 1, Vt, bpt, "left")
    toks = gi.fineweb_like_tokens(seed, B, T, vocab=Vt, eot_p=0.01)
    Et, Eb = f32(gi.normal_table(seed + 2, Vt, D)), f32(gi.normal_table(seed + 3, gi.BYTE_VOCAB, Db))
    rs = np.random.RandomState(seed + 4)
    g = f32(rs.standard_normal((B, T, D)) * 1e-3)              # small on purpose: |term| << 0.5 without a scale
    g *= (rs.random_sample((B, T, 1)) < keep)
    assert 0 < np.count_nonzero(g.any(-1)) < 0.05 * B * T
    pulled = orc.pull_from_left(orc.tokens_to_bytes(toks, tab.astype(np.float32)), bpt, gi.PAD, gi.EOT)
    ref = orc.embed_mix_bwd(toks, pulled, None, Et.astype(np.float64), Eb.astype(np.float64), g.astype(np.float64), mode="sum", bpt=bpt,
                            dtype=np.float64, **kw)
    got = mot.functional.embed_mix_backward(dev(g), dev(toks), dev(Et), dev(Eb), mode="sum", bpt=bpt, ids_a=dev(pulled), **kw)
    assert np.abs(ref["byte_table"]).max() > 0
    assert rel(host(got["byte_table"]), ref["byte_table"]) < TOL
    assert rel(host(got["tok_table"]), ref["tok_table"]) < TOL


@pytest.mark.parametrize("kw", _FX_KW, ids=["plain", "lc"])
def test_sum_backward_rows_far_below_the_sample(mot, kw):
    """Batch rows 1.. carry gradients 1e-9 x those of batch row 0 and use byte ids no position of row 0 uses: the gradient rows
    of those ids are sums of tiny terms only and must come out with their own relative precision (bar relative to THEIR maximum),
    not rounded against a scale taken from the large rows."""
    D, Db, bpt, Vt, B, T, seed = 768, 48, 16, 1024, 6, 700, 9311
    rs = np.random.RandomState(seed)
    toks = gi.fineweb_like_tokens(seed, B, T, vocab=Vt, eot_p=0.0)
    ids = rs.randint(0, 200, size=(B, T * bpt)).astype(np.int64)
    ids[1:] += 220                                              # rows 1..: byte ids 220..419 only
    Et, Eb = f32(gi.normal_table(seed + 2, Vt, D)), f32(gi.normal_table(seed + 3, gi.BYTE_VOCAB, Db))
    g = f32(rs.standard_normal((B, T, D)))
    g[1:] *= 1e-9
    ref = orc.embed_mix_bwd(toks, ids, None, Et.astype(np.float64), Eb.astype(np.float64), g.astype(np.float64), mode="sum", bpt=bpt,
                            dtype=np.float64, **kw)
    got = mot.functional.embed_mix_backward(dev(g), dev(toks), dev(Et), dev(Eb), mode="sum", bpt=bpt, ids_a=dev(ids), **kw)
    gb, rb = host(got["byte_table"]), ref["byte_table"]
    assert rel(gb[:200], rb[:200]) < TOL
    assert np.abs(rb[220:420]).max() < 1e-6 * np.abs(rb[:200]).max()        # the small rows really are far below the large ones
    assert rel(gb[220:420], rb[220:420]) < TOL                              # ... and are held to their own scale
    assert rel(host(got["tok_table"]), ref["tok_table"]) < TOL


@pytest.mark.parametrize("kw", _FX_KW, ids=["plain", "lc"])
def test_sum_backward_non_finite_gradient_reaches_the_byte_table(mot, kw):
    """A NaN in grad_out must show in the byte-table gradient of the ids that position uses (the reference's autograd propagates
    it); v_max_f32 drops NaNs, so the range check of the fixed-point path is made on bit patterns / on the wave sums."""
    D, Db, bpt, Vt, B, T, seed = 768, 48, 16, 512, 2, 256, 9321
    rs = np.random.RandomState(seed)
    toks = gi.fineweb_like_tokens(seed, B, T, vocab=Vt, eot_p=0.0)
    ids = rs.randint(0, 256, size=(B, T * bpt)).astype(np.int64)
    ids[0, :bpt] = 300 + np.arange(bpt)                         # the poisoned position uses ids nobody else uses
    Et, Eb = f32(gi.normal_table(seed + 2, Vt, D)), f32(gi.normal_table(seed + 3, gi.BYTE_VOCAB, Db))
    g = f32(rs.standard_normal((B, T, D)))
    g[0, 0, 5] = np.nan
    got = mot.functional.embed_mix_backward(dev(g), dev(toks), dev(Et), dev(Eb), mode="sum", bpt=bpt, ids_a=dev(ids), **kw)
    gb = host(got["byte_table"])
    assert np.isnan(gb[300 + 5 // Db]).any()                    # with the output norm every element of the position is NaN; without, element 5
    assert np.isnan(host(got["tok_table"])[toks[0, 0]]).any()
    clean = np.setdiff1d(np.arange(256), [])                    # rows 0..255 are shared with healthy positions only
    assert np.isfinite(gb[clean]).all()


def test_token_order_given_or_not(mot):
    """mot_token_order once per batch + backward calls that are handed the order (what the autograd node does) against backward
    calls that group the positions themselves and against the float64 oracle; the order buffer's sorted arrays are a permutation
    of the positions with non-decreasing token ids, every position exactly once."""
    D, Db, bpt, Vt, B, T, seed = 768, 48, 16, 3000, 6, 700, 9401
    tab = gi.synth_ttb(seed + 1, Vt, bpt, "left")
    toks = gi.fineweb_like_tokens(seed, B, T, vocab=Vt, eot_p=0.01)
    Et, Eb = f32(gi.normal_table(seed + 2, Vt, D)), f32(gi.normal_table(seed + 3, gi.BYTE_VOCAB, Db))
    g = f32(np.random.RandomState(seed + 4).standard_normal((B, T, D)))
    pulled = orc.pull_from_left(orc.tokens_to_bytes(toks, tab.astype(np.float32)), bpt, gi.PAD, gi.EOT)
    F = mot.functional
    dtoks = dev(toks)
    order = F.token_order(dtoks, Vt)
    n = B * T
    o = host(order)
    assert o.size == 2 * Vt + 3 * n
    pos_sorted, tok_sorted = o[2 * Vt + n:2 * Vt + 2 * n], o[2 * Vt + 2 * n:]
    assert np.array_equal(np.sort(pos_sorted), np.arange(n)) and (np.diff(tok_sorted) >= 0).all()
    assert np.array_equal(toks.reshape(-1)[pos_sorted], tok_sorted)
    for kw in (dict(norm_out=True), dict(norm_tok=True, norm_byte=True, norm_out=True)):
        ref = orc.embed_mix_bwd(toks, pulled, None, Et.astype(np.float64), Eb.astype(np.float64), g.astype(np.float64), mode="sum", bpt=bpt,
                                dtype=np.float64, **kw)
        a = F.embed_mix_backward(dev(g), dtoks, dev(Et), dev(Eb), mode="sum", bpt=bpt, ids_a=dev(pulled), token_order=order, **kw)
        b = F.embed_mix_backward(dev(g), dtoks, dev(Et), dev(Eb), mode="sum", bpt=bpt, ids_a=dev(pulled), **kw)
        for k in ("tok_table", "byte_table"):
            assert rel(host(a[k]), ref[k]) < TOL and rel(host(b[k]), ref[k]) < TOL
    ref = orc.embed_mix_bwd(toks, None, None, Et.astype(np.float64), None, g.astype(np.float64), mode="noop", bpt=0, dtype=np.float64, norm_tok=True)
    a = F.embed_mix_backward(dev(g), dtoks, dev(Et), mode="noop", norm_tok=True, token_order=order)
    assert rel(host(a["tok_table"]), ref["tok_table"]) < TOL
    with pytest.raises(ValueError, match="token_order"):
        F.embed_mix_backward(dev(g), dtoks, dev(Et), mode="noop", norm_tok=True, token_order=order[:-1])


def test_autograd_reuses_the_token_order_per_token_tensor(mot):
    """The autograd node asks for the token order beside the forward (side stream) and keeps it per token TENSOR and version: a
    second forward + backward over the same tensor reuses it, an in-place change of the tokens makes a new one; gradients match
    the oracle every time (a stale order would scatter rows to the wrong tokens)."""
    D, Db, bpt, Vt, B, T, seed = 256, 32, 8, 700, 4, 300, 9411
    F = mot.functional
    tab = gi.synth_ttb(seed + 1, Vt, bpt, "left")
    Et, Eb = f32(gi.normal_table(seed + 2, Vt, D)), f32(gi.normal_table(seed + 3, gi.BYTE_VOCAB, Db))
    g = f32(np.random.RandomState(seed + 4).standard_normal((B, T, D)))
    pt, pb = torch.nn.Parameter(dev(Et)), torch.nn.Parameter(dev(Eb))
    toks_a = gi.fineweb_like_tokens(seed, B, T, vocab=Vt, eot_p=0.01)
    toks_b = gi.fineweb_like_tokens(seed + 9, B, T, vocab=Vt, eot_p=0.01)
    dtoks = dev(toks_a)
    F._token_orders.clear()

    def run(toks_np):
        pt.grad = pb.grad = None
        x = mot.embed_mix(dtoks, pt, pb, mode="sum", bpt=bpt, ttb=dev(tab), pull="left", norm_out=True)
        x.backward(dev(g))
        torch.cuda.synchronize()
        pulled = orc.pull_from_left(orc.tokens_to_bytes(toks_np, tab.astype(np.float32)), bpt, gi.PAD, gi.EOT)
        ref = orc.embed_mix_bwd(toks_np, pulled, None, Et.astype(np.float64), Eb.astype(np.float64), g.astype(np.float64), mode="sum", bpt=bpt,
                                dtype=np.float64, norm_out=True)
        assert rel(host(pt.grad), ref["tok_table"]) < TOL and rel(host(pb.grad), ref["byte_table"]) < TOL

    run(toks_a)
    assert len(F._token_orders.entries) == 1
    first = F._token_orders.entries[0][3]
    run(toks_a)
    assert len(F._token_orders.entries) == 1 and F._token_orders.entries[0][3] is first      # reused
    dtoks.copy_(dev(toks_b))                                                                  # same tensor, new version
    run(toks_b)
    assert F._token_orders.entries[0][3] is not first
    mot.check_status()


# ---- embed_mix_bwd_plain_kernel (round 3): its own matrix -- every built row width, both modes, with and without the output norm,
#      fp32 and bf16, the corner cases of its segment / run / stretch logic, and its slow path (byte ids without an LDS row)
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("D,Db,bpt,mode,norm_out,B,T,Vt,seed", [
    (256, 32, 8, "sum", True, 3, 333, 700, 9601),
    (512, 32, 16, "sum", False, 2, 500, 300, 9602),
    (768, 48, 16, "sum", True, 1, 4097, 50, 9603),        # few distinct tokens: long runs across wave stretches and 128-place segments
    (768, 128, 6, "sum", True, 2, 300, 600, 9604),        # byte rows of 128 columns: only a third of the table has an LDS row -> the slow path
    (256, 4, 64, "sum", True, 2, 130, 100, 9605),         # one 16-byte chunk per byte slot, 64 slots
    (768, 48, 16, "sum", True, 1, 5, 9, 9606),            # fewer positions than waves
    (512, 0, 0, "noop", True, 3, 400, 500, 9607),
    (768, 0, 0, "noop", False, 2, 777, 64, 9608),
    (256, 0, 0, "noop", False, 1, 1, 3, 9609),            # one position
])
def test_plain_backward_kernel_matrix(mot, dtype, D, Db, bpt, mode, norm_out, B, T, Vt, seed):
    rs = np.random.RandomState(seed)
    bf = dtype == "bf16"
    if bf and Db % 8:
        pytest.skip("bf16 table rows are multiples of 16 bytes (validated at the boundary)")
    rnd = (lambda a: orc.bf16_round(a)) if bf else f32
    toks = gi.fineweb_like_tokens(seed, B, T, vocab=Vt, eot_p=0.0)
    Et = rnd(gi.normal_table(seed + 2, Vt, D))
    g = rnd(rs.standard_normal((B, T, D)))
    cast = (lambda a: dev(a).bfloat16()) if bf else dev
    kw = dict(norm_out=norm_out)
    if bf:
        orc.set_eps(2.0 ** -7)
    try:
        if mode == "sum":
            Eb = rnd(gi.normal_table(seed + 3, gi.BYTE_VOCAB, Db))
            ids = rs.randint(0, gi.BYTE_VOCAB, (B, T * bpt)).astype(np.int64)
            ref = orc.embed_mix_bwd(toks, ids, None, Et.astype(np.float64), Eb.astype(np.float64), g.astype(np.float64), mode="sum", bpt=bpt,
                                    dtype=np.float64, **kw)
            got = mot.functional.embed_mix_backward(cast(g), dev(toks), cast(Et), cast(Eb), mode="sum", bpt=bpt, ids_a=dev(ids), **kw)
            assert rel(host(got["byte_table"]), ref["byte_table"]) < TOL
        else:
            ref = orc.embed_mix_bwd(toks, None, None, Et.astype(np.float64), None, g.astype(np.float64), mode="noop", bpt=0, dtype=np.float64, **kw)
            got = mot.functional.embed_mix_backward(cast(g), dev(toks), cast(Et), mode="noop", **kw)
    finally:
        orc.set_eps(0.0)
    mot.check_status()
    assert got["tok_table"].dtype == torch.float32
    assert rel(host(got["tok_table"]), ref["tok_table"]) < TOL
    untouched = np.setdiff1d(np.arange(Vt), np.unique(toks))
    assert not host(got["tok_table"])[untouched].any()
    # accumulate semantics: a second call into the same buffers doubles them
    into = {k: v.clone() for k, v in got.items() if v is not None and k in ("tok_table", "byte_table")}
    if mode == "sum":
        mot.functional.embed_mix_backward(cast(g), dev(toks), cast(Et), cast(Eb), mode="sum", bpt=bpt, ids_a=dev(ids), into=into, **kw)
    else:
        mot.functional.embed_mix_backward(cast(g), dev(toks), cast(Et), mode="noop", into=into, **kw)
    assert rel(host(into["tok_table"]), 2 * ref["tok_table"]) < TOL


def test_plain_backward_kernel_flags_bad_ids(mot):
    """Out-of-range byte ids in the plain kernel's id path (one 8-byte load per lane + ds_bpermute): flagged in the status word, read
    as row 0, never a fault; the other positions' gradients are unaffected."""
    D, Db, bpt, Vt, B, T, seed = 768, 48, 16, 300, 2, 200, 9621
    rs = np.random.RandomState(seed)
    toks = gi.fineweb_like_tokens(seed, B, T, vocab=Vt, eot_p=0.0)
    ids = rs.randint(0, gi.BYTE_VOCAB, (B, T * bpt)).astype(np.int64)
    bad = ids.copy()
    bad[1, 5 * bpt + 3] = gi.BYTE_VOCAB + 7
    bad[0, 17 * bpt] = -2
    Et, Eb = f32(gi.normal_table(seed + 2, Vt, D)), f32(gi.normal_table(seed + 3, gi.BYTE_VOCAB, Db))
    g = f32(rs.standard_normal((B, T, D)))
    clean = ids.copy(); clean[1, 5 * bpt + 3] = 0; clean[0, 17 * bpt] = 0
    ref = orc.embed_mix_bwd(toks, clean, None, Et.astype(np.float64), Eb.astype(np.float64), g.astype(np.float64), mode="sum", bpt=bpt, dtype=np.float64,
                            norm_out=True)
    got = mot.functional.embed_mix_backward(dev(g), dev(toks), dev(Et), dev(Eb), mode="sum", bpt=bpt, ids_a=dev(bad), norm_out=True)
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        mot.check_status()
    assert rel(host(got["tok_table"]), ref["tok_table"]) < TOL and rel(host(got["byte_table"]), ref["byte_table"]) < TOL


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("D,Db,bpt,mode,B,T,Vt,seed", [
    (256, 16, 16, "sum", 48, 2048, 3000, 9651),          # 98 304 positions: every workgroup and wave has a full stretch
    (768, 0, 0, "noop", 96, 1024, 20000, 9652),          # tokens-only, many short runs: a flush every few places
    (512, 32, 16, "sum", 64, 2048, 7, 9653),             # seven tokens: runs far longer than a wave's stretch (rows shared by every wave)
])
def test_plain_backward_kernel_large_batches(mot, dtype, D, Db, bpt, mode, B, T, Vt, seed):
    """embed_mix_bwd_plain_kernel on batches that fill the chip (all 256 workgroups, 128-place segments, runs cut by stretch
    boundaries, runs of length one): same bar against the float64 oracle.  (Written for the flusher-wave variant of round 3 -- one wave
    issuing the atomic row-adds of the other eleven through LDS mailboxes; correct, 20 % slower, not kept: DESIGN.md section 3.)"""
    rs = np.random.RandomState(seed)
    bf = dtype == "bf16"
    rnd = (lambda a: orc.bf16_round(a)) if bf else f32
    cast = (lambda a: dev(a).bfloat16()) if bf else dev
    toks = gi.fineweb_like_tokens(seed, B, T, vocab=Vt, eot_p=0.0)
    Et = rnd(gi.normal_table(seed + 2, Vt, D))
    g = rnd(rs.standard_normal((B, T, D)))
    if bf:
        orc.set_eps(2.0 ** -7)
    try:
        if mode == "sum":
            Eb = rnd(gi.normal_table(seed + 3, gi.BYTE_VOCAB, Db))
            ids = rs.randint(0, 256, (B, T * bpt)).astype(np.int64)
            ref = orc.embed_mix_bwd(toks, ids, None, Et.astype(np.float64), Eb.astype(np.float64), g.astype(np.float64), mode="sum", bpt=bpt,
                                    dtype=np.float64, norm_out=True)
            got = mot.functional.embed_mix_backward(cast(g), dev(toks), cast(Et), cast(Eb), mode="sum", bpt=bpt, ids_a=dev(ids), norm_out=True)
            assert rel(host(got["byte_table"]), ref["byte_table"]) < TOL
        else:
            ref = orc.embed_mix_bwd(toks, None, None, Et.astype(np.float64), None, g.astype(np.float64), mode="noop", bpt=0, dtype=np.float64, norm_out=True)
            got = mot.functional.embed_mix_backward(cast(g), dev(toks), cast(Et), mode="noop", norm_out=True)
    finally:
        orc.set_eps(0.0)
    torch.cuda.synchronize()
    mot.check_status()
    assert rel(host(got["tok_table"]), ref["tok_table"]) < TOL
    untouched = np.setdiff1d(np.arange(Vt), np.unique(toks))
    assert not host(got["tok_table"])[untouched].any()
