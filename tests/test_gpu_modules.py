"""-m gpu: the nn.Module / loader boundary (the reference's own class and function names) driven
the way the reference's training loops drive it, against the fixtures the reference produced."""
import json
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import oracle as orc
from test_gpu_embed import assert_gemm_close
from util_gpu import DEV, assert_close, dev, f32, host

pytestmark = pytest.mark.gpu
G = gi.GOLDEN_DIR


@pytest.fixture(scope="module")
def M():
    from mixture_of_tokenizers_amd import modules
    return modules


SCALED = [("small", 97, 32, 8, 64, 8, 2, 16, 401), ("c2dims", 512, 256, 32, 768, 16, 1, 48, 402)]


class Host(torch.nn.Module):
    """The two attributes of the reference GPT that form the path (train_gpt.py:549-606)."""

    def __init__(self, M, dims, vocab, bp, fused=True):
        super().__init__()
        self.embed = M.FlexibleEmbedding(dims, vocab, bp, fused=fused)
        self.byte_mixin = M.ByteMixin(dims, 64, bp)

    @torch.no_grad()
    def forward(self, toks_in, bytes_padded_in, bytes_pulled_in):
        xt, xb = self.embed(tokens=toks_in, byte_tensor=bytes_padded_in, byte_tensor_pulled=bytes_pulled_in)  # :605
        return self.byte_mixin(xt, xb)                                                                         # :606


@pytest.mark.parametrize("case", SCALED, ids=lambda c: c[0])
@pytest.mark.parametrize("fused", [True, False])
def test_scaled_pretrain_modules(M, case, fused):
    name, Vt, Dt, Db, Dm, bpt, B, T, seed = case
    z = np.load(G / "float_scaled.npz")
    toks, padded, pulled = dev(z[f"{name}/tokens"]), dev(z[f"{name}/padded"]), dev(z[f"{name}/pulled"])
    Et, Eb = f32(gi.normal_table(seed + 1, Vt, Dt)), f32(gi.normal_table(seed + 2, gi.BYTE_VOCAB, Db))
    W = f32(gi.casted_linear_weight(seed + 3, Dm, Dt + bpt * Db))
    for mode, kw in (("padded", dict(pull_in=False)), ("pulled", dict(pull_in=True)),
                     ("padded_and_pulled", dict(pull_in=True, add_padded_and_pulled=True))):
        if f"{name}/{mode}/f64/x" not in z:
            continue
        bp = M.ByteHyperparameters(bytes_per_token=bpt, vocab_size=gi.BYTE_VOCAB, byte_mixin_method="concat", **kw)
        net = Host(M, M.ModelDims(model_dim=Dm, byte_dim=Db, token_dim=Dt), Vt, bp, fused=fused).to(DEV)
        net.load_state_dict({"embed.embed_tokens.weight": dev(Et), "embed.embed_bytes.weight": dev(Eb),
                             "byte_mixin.mixin.mixin.weight": dev(W)})
        x = net(toks, padded, pulled)
        assert x.shape == (B, T, Dm) and x.dtype == torch.float32
        assert_gemm_close(host(x), z[f"{name}/{mode}/f32/x"], z[f"{name}/{mode}/f64/x"])
        if not fused and name == "small":
            with torch.no_grad():
                te, be = net.embed(toks, padded, pulled)
            assert_close(host(te), z[f"{name}/{mode}/f32/tok_embs"])
            assert_close(host(be), z[f"{name}/{mode}/f32/byte_embs"])
    bp = M.ByteHyperparameters(bytes_per_token=bpt, vocab_size=gi.BYTE_VOCAB, byte_mixin_method="noop")
    net = Host(M, M.ModelDims(model_dim=Dt, byte_dim=Db, token_dim=Dt), Vt, bp, fused=fused).to(DEV)
    net.embed.embed_tokens.weight.data.copy_(dev(Et))
    assert_close(host(net(toks, None, None)), z[f"{name}/noop/f32/x"])


def test_fused_front_end(M):
    name, Vt, Dt, Db, Dm, bpt, B, T, seed = SCALED[1]
    z = np.load(G / "float_scaled.npz")
    bp = M.ByteHyperparameters(bytes_per_token=bpt, vocab_size=gi.BYTE_VOCAB, byte_mixin_method="concat", pull_in=True)
    fe = M.FusedFrontEnd(M.ModelDims(model_dim=Dm, byte_dim=Db, token_dim=Dt), Vt, bp, dev(gi.synth_ttb(seed + 1000, Vt, bpt, "left"))).to(DEV)
    assert "ttb" not in fe.state_dict() and sorted(fe.state_dict()) == sorted(
        ["embed.embed_tokens.weight", "embed.embed_bytes.weight", "byte_mixin.mixin.mixin.weight"])
    fe.embed.embed_tokens.weight.data.copy_(dev(f32(gi.normal_table(seed + 1, Vt, Dt))))
    fe.embed.embed_bytes.weight.data.copy_(dev(f32(gi.normal_table(seed + 2, gi.BYTE_VOCAB, Db))))
    fe.byte_mixin.mixin.mixin.weight.data.copy_(dev(f32(gi.casted_linear_weight(seed + 3, Dm, Dt + bpt * Db))))
    y = fe(dev(z[f"{name}/tokens"]))                       # autograd on: records one backward node
    assert y.requires_grad and y.grad_fn is not None
    with torch.no_grad():
        r = fe(dev(z[f"{name}/tokens"]), return_ids=True)
    np.testing.assert_array_equal(host(r.ids_pulled), z[f"{name}/pulled"])
    assert_gemm_close(host(r.x), z[f"{name}/pulled/f32/x"], z[f"{name}/pulled/f64/x"])


def test_mathblations_front_end_and_weight_tying(M):
    """BASELINE config 1 through the wte / dte / digit_mixin attributes (model.py:304-327), with
    wte.weight tied to an lm_head Parameter as model.py:316-317 does."""
    z = np.load(G / "mathblations_c1.npz")
    D = 256
    cfg = M.GPTConfig(vocab_size=1003, n_embd_tok=D, n_embd_digit=D, length_factor=3, digit_mixin_method="concat")
    fe = M.DigitFrontEnd(cfg).to(DEV)
    lm_head = torch.nn.Linear(D, 1003, bias=False).to(DEV)
    fe.wte.weight = lm_head.weight                                        # tied: one Parameter
    Wf, bf = gi.linear_weight_bias(603, D, 4 * D)
    with torch.no_grad():
        lm_head.weight.copy_(dev(f32(gi.normal_table(601, 1003, D))))     # written through the OTHER owner
        fe.dte.weight.copy_(dev(f32(gi.normal_table(602, 14, D))))
        fe.digit_mixin.fc.weight.copy_(dev(f32(Wf))); fe.digit_mixin.fc.bias.copy_(dev(f32(bf)))
        x = fe(dev(z["x_tokens"]), dev(z["x_digit_tokens"]))
    assert_gemm_close(host(x), z["concat/f32/x"], z["concat/f64/x"])
    with torch.no_grad():                                                 # dense seam inputs give the same result
        we = torch.nn.functional.embedding(dev(z["x_tokens"]), lm_head.weight)
        de = torch.nn.functional.embedding(dev(z["x_digit_tokens"]), fe.dte.weight)
        x2 = fe.digit_mixin(we, de)
    assert_gemm_close(host(x2), z["concat/f32/x"], z["concat/f64/x"])
    noop = M.DigitFrontEnd(M.GPTConfig(vocab_size=1003, n_embd_tok=D, digit_mixin_method="noop")).to(DEV)
    with torch.no_grad():
        y = noop(dev(z["x_tokens"]))
        assert torch.equal(y, torch.nn.functional.embedding(dev(z["x_tokens"]), noop.wte.weight))   # a pure gather is exact


@pytest.mark.parametrize("variant", ["71", "71041", "71081"])
def test_sum_front_end(M, variant):
    name, Vt, D, Db, bpt, T, seed = ("c2dims", 512, 768, 48, 16, 48, 502)
    z = np.load(G / "sum_modes.npz")
    fe = M.SumFrontEnd(Vt, gi.BYTE_VOCAB, D, Db, bpt, variant=variant, ttb=dev(gi.synth_ttb(seed + 1000, Vt, bpt, "left"))).to(DEV)
    with torch.no_grad():
        fe.embed_tokens.weight.copy_(dev(f32(gi.normal_table(seed + 1, Vt, D))))
        fe.embed_bytes.weight.copy_(dev(f32(gi.normal_table(seed + 2, gi.BYTE_VOCAB, Db))))
        if fe.scalars is not None:
            fe.scalars.copy_(dev(f32(z[f"{name}/scales"][::-1].copy())))      # [-1] tokens, [-2] bytes
        toks = dev(z[f"{name}/tokens"])[0]                                    # 1-D, as modded-nanogpt feeds it
        x = fe(toks)
        x2 = fe(toks, dev(z[f"{name}/pulled"]).to(torch.int32))               # loader-provided ids (int32 there)
    assert x.shape == (1, T, D)
    assert_close(host(x), z[f"{name}/r{variant}/f32"])
    assert torch.equal(x, x2)


def test_loader_create_data_matches_reference_fixture():
    from mixture_of_tokenizers_amd import loader
    from mixture_of_tokenizers_amd.modules import ByteHyperparameters
    z = np.load(G / "loader.npz")
    bpt, vocab = 16, 512
    tab = dev(gi.synth_ttb(3001, vocab, bpt, "left"))
    # main() forces pull_out=False when the mixout is noop (train_gpt.py:1013-1014); any other combination is
    # a KeyError in the reference's dispatch table too (train_gpt.py:766-783)
    bp = ByteHyperparameters(bytes_per_token=bpt, byte_mixin_method="concat", pull_in=True, byte_mixout_method="noop", pull_out=False)
    create = loader.make_create_data_from_toks(bp, tab, tab)          # _create_data_from_toks_TT_FF
    data = torch.from_numpy(z["data"])
    pos, batch, seq = int(z["pos"]), int(z["batch"]), int(z["seq"])
    for world in (1, 2, 4):
        for rank in range(world):
            toks = loader.rank_slice(data, pos, batch, seq, rank, world).to(DEV)
            toks_in, bytes_padded_in, bytes_pulled_in, targets = create(toks)
            p = f"w{world}r{rank}"
            for got, key in ((toks_in, "toks_in"), (bytes_padded_in, "bytes_padded_in"), (bytes_pulled_in, "bytes_pulled_in"), (targets, "targets")):
                assert got.is_contiguous()
                np.testing.assert_array_equal(host(got), z[f"{p}/{key}"])
            assert toks_in.dtype == torch.int32 and bytes_pulled_in.dtype == torch.int64


_VARIANTS = {   # dispatch key (byte_in, pull_in, byte_out, pull_out) of train_gpt.py:766-783 -> the reference function's suffix
    (True, True, True, True): "TT_TT", (True, False, True, True): "TF_TT", (True, True, True, False): "TT_TF", (True, True, False, False): "TT_FF",
    (False, False, True, True): "FF_TT", (False, False, True, False): "FF_TF", (True, False, False, False): "TF_FF", (False, False, False, False): "FF_FF"}


@pytest.mark.parametrize("key", sorted(_VARIANTS), ids=lambda k: _VARIANTS[k])
def test_loader_all_eight_create_data_variants_match_the_reference(key):
    """Every `_create_data_from_toks_*` of train_gpt.py:686-764, run as it stands (AST-extracted, oracle/gen_golden.py) on the same
    rank slices: inputs from the left-padded table pulled from the left, targets from the right-padded table pulled from the right,
    shifted by one token / bpt byte slots; a None of the reference is a None here (bit-exact otherwise)."""
    from mixture_of_tokenizers_amd import loader
    from mixture_of_tokenizers_amd.modules import ByteHyperparameters
    z = np.load(G / "loader.npz")
    bpt, vocab = 16, 512
    tabl, tabr = dev(gi.synth_ttb(3001, vocab, bpt, "left")), dev(gi.synth_ttb(3001, vocab, bpt, "right"))
    byte_in, pull_in, byte_out, pull_out = key
    bp = ByteHyperparameters(bytes_per_token=bpt, byte_mixin_method="concat" if byte_in else "noop", pull_in=pull_in,
                             byte_mixout_method="copy" if byte_out else "noop", pull_out=pull_out, padding_in="left", padding_out="right")
    create = loader.make_create_data_from_toks(bp, tabl, tabr)
    data = torch.from_numpy(z["data"])
    pos, batch, seq = int(z["pos"]), int(z["batch"]), int(z["seq"])
    name = _VARIANTS[key]
    assert name in set(z["variants/names"])
    for world, rank in ((1, 0), (2, 1)):
        toks = loader.rank_slice(data, pos, batch, seq, rank, world).to(DEV)
        got = dict(zip(("toks_in", "bytes_padded_in", "bytes_pulled_in", "targets"), create(toks)))
        for what, val in got.items():
            k = f"variants/w{world}r{rank}/{name}/{what}"
            if k not in z.files:
                assert val is None, (name, what)
            else:
                assert val is not None and val.is_contiguous(), (name, what)
                np.testing.assert_array_equal(host(val), z[k])
                assert val.dtype == (torch.int32 if what == "toks_in" or (what == "targets" and not byte_out) else torch.int64)
    with pytest.raises(KeyError):      # a combination outside the reference's dispatch table is a KeyError there too (766-783)
        loader.make_create_data_from_toks(ByteHyperparameters(bytes_per_token=bpt, byte_mixin_method="noop", pull_in=True, byte_mixout_method="noop",
                                                              pull_out=True), tabl, tabr)


def test_distributed_data_generator_end_to_end(tmp_path, monkeypatch):
    """Shards on disk -> generator (train_gpt.py:651-806) -> tensors on the GPU, both ranks of a
    2-way batch shard, checked against the oracle run on the same token stream."""
    from mixture_of_tokenizers_amd import loader
    from mixture_of_tokenizers_amd.modules import ByteHyperparameters
    vocab, bpt, seq, batch = 300, 8, 31, 4
    tabl, tabr = gi.synth_ttb(11, vocab, bpt, "left"), gi.synth_ttb(11, vocab, bpt, "right")
    (tmp_path / "embeddings").mkdir(); (tmp_path / "data").mkdir()
    for side, tab in (("left", tabl), ("right", tabr)):
        (tmp_path / "embeddings" / f"ttb_{bpt}_{side}_pad.json").write_text(json.dumps({str(i): [int(v) for v in r] for i, r in enumerate(tab)}))
    stream = gi.edge_tokens(12, 1, 3000, vocab).reshape(-1)
    loader.write_data_shard(tmp_path / "data" / "train_000001.bin", stream)
    monkeypatch.chdir(tmp_path)
    bp = ByteHyperparameters(bytes_per_token=bpt, byte_mixin_method="concat", byte_mixout_method="copy", padding_in="left",
                             padding_out="right", pull_in=True, pull_out=True)
    for rank in range(2):
        gen = loader.distributed_data_generator("data/train_*.bin", seq, batch, rank, 2, bp, vocab_size=vocab, device=DEV)
        for step in range(3):
            toks_in, padded_in, pulled_in, targets = next(gen)
            L = batch * (seq + 1) // 2
            ref = stream[step * batch * (seq + 1) + rank * L:][:L].reshape(-1, seq + 1)
            np.testing.assert_array_equal(host(toks_in), ref[:, :-1])
            pi = orc.tokens_to_bytes(ref, tabl.astype(np.float32))
            np.testing.assert_array_equal(host(padded_in), pi[:, :-bpt])
            np.testing.assert_array_equal(host(pulled_in), orc.pull_from_left(pi, bpt, gi.PAD, gi.EOT)[:, :-bpt])
            po = orc.pull_from_right(orc.tokens_to_bytes(ref, tabr.astype(np.float32)), bpt, gi.PAD, gi.EOT)
            np.testing.assert_array_equal(host(targets), po[:, bpt:])


def test_make_embedding_reproduces_reference_quirk(tmp_path, monkeypatch):
    """make_embedding on a table file without the EOT row (the only one the reference ships): the
    missing row keeps torch's random init, drawn in the same order as the reference draws it
    (data_creation.py:51-58), so the same seed yields the same garbage ids (SURVEY section 7 quirk i)."""
    from mixture_of_tokenizers_amd import data_creation as dc
    z = np.load(G / "make_embedding.npz")
    rows = np.load(G / "ttb_8_left_pad.npz")["rows"]
    (tmp_path / "embeddings").mkdir()
    (tmp_path / "embeddings" / "ttb_8_left_pad.json").write_text(json.dumps({str(i): [int(v) for v in r] for i, r in enumerate(rows)}))
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(int(z["seed"]))
    emb = dc.make_embedding("ttb_8_left_pad.json", gi.GPT2_VOCAB)
    assert isinstance(emb, torch.nn.Embedding) and emb.weight.dtype == torch.float32 and not emb.weight.requires_grad
    np.testing.assert_array_equal(emb.weight[50256].numpy(), z["eot_row_f32"])
    emb = emb.to(DEV)                                                        # train_gpt.py:666
    np.testing.assert_array_equal(host(dc.tokens_to_bytes(dev(z["tokens"]), emb)), z["padded"])
    with torch.no_grad():
        emb.weight[50256] = 457.0                                            # the table cache follows in-place edits
    assert host(dc.tokens_to_bytes(dev(z["tokens"]), emb)).reshape(4, 8)[2].tolist() == [457] * 8


def test_training_steps_reduce_the_loss(M):
    """The point of a drop-in: a few optimizer steps through the fused modules (forward + backward kernels,
    torch.optim on the same Parameters the reference groups by name, train_gpt.py:1154-1157) lower a loss."""
    torch.manual_seed(0)
    Vt, Dt, Db, Dm, bpt, B, T = 512, 64, 16, 128, 8, 4, 64
    bp = M.ByteHyperparameters(bytes_per_token=bpt, vocab_size=gi.BYTE_VOCAB, byte_mixin_method="concat", pull_in=True)
    net = Host(M, M.ModelDims(model_dim=Dm, byte_dim=Db, token_dim=Dt), Vt, bp).to(DEV)
    from mixture_of_tokenizers_amd import data_creation as dc
    tab = dev(gi.synth_ttb(5, Vt, bpt, "left"))
    toks = dev(gi.fineweb_like_tokens(6, B, T, vocab=Vt, eot_p=0.02))
    padded = dc.tokens_to_bytes(toks, tab)
    pulled = dc.pull_from_left(padded, bpt, gi.PAD, gi.EOT)
    target = torch.randn(B, T, Dm, device=DEV)
    opt = torch.optim.Adam([p for n, p in net.named_parameters()], lr=0.05)
    losses = []
    for _ in range(12):
        opt.zero_grad(set_to_none=True)
        xt, xb = net.embed(tokens=toks, byte_tensor=padded, byte_tensor_pulled=pulled)
        x = net.byte_mixin(xt, xb)
        loss = ((x - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())
    assert losses[-1] < 0.8 * losses[0], losses
