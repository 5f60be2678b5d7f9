#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

TEST INFRASTRUCTURE.  This script imports the reference from /root/reference on CPU
(eager: TORCHDYNAMO_DISABLE=1), feeds it the seeded inputs of tests/golden_inputs.py
and stores inputs + the reference's outputs as small .npz fixtures.  The reference
itself never enters the repo and never travels to the GPU box; the fixtures do.

How each part of the reference is loaded:
  * scaled-pre-train/data_creation.py  -- imported whole, with an empty stand-in module
    for `tiktoken` (only used by its offline tokeniser, not by the path).
  * scaled-pre-train/train_gpt.py      -- cannot be imported (allocates on cuda at import);
    the class/function definitions on the path are taken from its AST and exec'd.
  * modded-nanogpt/runs/71_*.py        -- a training script; `norm` and `mixin_bytes`
    are taken from its AST (minus the @torch.compile decorator).
  * mathblations/{model,data}.py       -- imported whole.

Run:  python oracle/gen_golden.py          (about a minute)
"""
from __future__ import annotations

import ast
import json
import os
import random
import sys
import types
from pathlib import Path

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
os.environ.setdefault("HF_HUB_OFFLINE", "1")

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(REPO / "tests"))
import golden_inputs as gi  # noqa: E402

OUT = REPO / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)
torch.set_num_threads(1)  # one fixed reduction order for the fp32 goldens


# ------------------------------------------------------------------ reference loaders
def load_data_creation():
    sys.modules.setdefault("tiktoken", types.ModuleType("tiktoken"))
    sys.path.insert(0, str(REF / "scaled-pre-train"))
    import data_creation  # type: ignore
    return data_creation


def _exec_nodes(path: Path, names: set[str], ns: dict, strip_decorators=True) -> dict:
    tree = ast.parse(path.read_text())
    picked = []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            if strip_decorators and isinstance(node, ast.FunctionDef):
                node.decorator_list = []
            picked.append(node)
    missing = names - {n.name for n in picked}
    assert not missing, f"{path}: not found: {missing}"
    mod = ast.Module(body=picked, type_ignores=[])
    exec(compile(mod, str(path), "exec"), ns)
    return ns


def load_train_gpt_defs() -> dict:
    import einops
    from dataclasses import dataclass
    from typing import Literal
    import torch.nn.functional as F
    from torch import Tensor, nn
    ns = dict(torch=torch, nn=nn, F=F, Tensor=Tensor, einops=einops, dataclass=dataclass, Literal=Literal)
    names = {"ByteHyperparameters", "ModelDims", "norm", "CastedLinear", "FlexibleEmbedding",
             "ByteMixinNoop", "ByteMixinConcat", "ByteMixin", "Rotary", "CrossAttention", "ByteMixinCrossAttn"}
    # the dataclasses keep their decorators
    return _exec_nodes(REF / "scaled-pre-train" / "train_gpt.py", names, ns, strip_decorators=False)


def load_create_data_TT_FF(dc, ttb_in, pull_in, bpt):
    """_create_data_from_toks_TT_FF (train_gpt.py:720-728) is nested inside the loader
    generator; take its FunctionDef from the AST and bind the closure names."""
    from torch import Tensor
    tree = ast.parse((REF / "scaled-pre-train" / "train_gpt.py").read_text())
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "_create_data_from_toks_TT_FF")
    ns = dict(tokens_to_bytes=dc.tokens_to_bytes, ttb_in=ttb_in, pull_in=pull_in, bpt=bpt, Tensor=Tensor)
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "train_gpt.py", "exec"), ns)
    return ns["_create_data_from_toks_TT_FF"]


def load_create_data_variants(dc, ttb_in, ttb_out, pull_in, pull_out, bpt) -> dict:
    """All eight `_create_data_from_toks_*` variants (train_gpt.py:686-764), nested inside the loader generator: every nested
    FunctionDef of that name is taken from the AST and executed with the closure names bound, exactly as it stands."""
    from torch import Tensor
    tree = ast.parse((REF / "scaled-pre-train" / "train_gpt.py").read_text())
    fns = [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name.startswith("_create_data_from_toks_")]
    assert len(fns) == 8, [f.name for f in fns]
    ns = dict(tokens_to_bytes=dc.tokens_to_bytes, ttb_in=ttb_in, ttb_out=ttb_out, pull_in=pull_in, pull_out=pull_out, bpt=bpt, Tensor=Tensor)
    exec(compile(ast.Module(body=fns, type_ignores=[]), "train_gpt.py", "exec"), ns)
    return {f.name: ns[f.name] for f in fns}


def load_run71_defs() -> dict:
    import torch.nn.functional as F
    from torch import Tensor, nn
    ns = dict(torch=torch, nn=nn, F=F, Tensor=Tensor)
    return _exec_nodes(REF / "modded-nanogpt" / "runs" / "71_mot-in_toks-valemb.py", {"norm", "mixin_bytes"}, ns)


def load_mathblations():
    sys.path.insert(0, str(REF / "mathblations"))
    import data as mdata  # type: ignore
    import model as mmodel  # type: ignore
    return mmodel, mdata


# ------------------------------------------------------------------ helpers
def ttb_embedding(table_i16: np.ndarray) -> torch.nn.Embedding:
    """What make_embedding (data_creation.py:51-58) yields for a complete table."""
    emb = torch.nn.Embedding(*table_i16.shape)
    emb.weight.data = torch.from_numpy(table_i16.astype(np.float32))
    emb.weight.requires_grad = False
    return emb


def t2n(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ G0: data fixture
def gen_ttb_fixture():
    src = REF / "modded-nanogpt" / "embeddings" / "ttb_8_left_pad.json"
    d = json.loads(src.read_text())
    rows = np.full((len(d), 8), gi.PAD, dtype=np.int16)
    for k, v in d.items():
        rows[int(k)] = v
    assert sorted(int(k) for k in d) == list(range(len(d)))
    np.savez_compressed(OUT / "ttb_8_left_pad.npz", rows=rows)
    print("ttb fixture:", rows.shape)


# ------------------------------------------------------------------ G1/G2: index path
def gen_index(dc):
    out = {}
    # G1: real GPT-2 table, bpt 8
    tl = gi.load_real_ttb8()
    tr = gi.to_right_pad(tl)
    toks = gi.fineweb_like_tokens(11, 4, 64, eot_p=0.03)
    e = gi.GPT2_VOCAB - 1
    toks[0, 0] = e; toks[1, 63] = e; toks[2, 30] = e; toks[2, 31] = e; toks[3][toks[3] == e] = 11
    out["real/tokens"] = toks
    for side, tab, fn in (("left", tl, dc.pull_from_left), ("right", tr, dc.pull_from_right)):
        bt = dc.tokens_to_bytes(torch.from_numpy(toks), ttb_embedding(tab))
        out[f"real/{side}/padded"] = t2n(bt)
        out[f"real/{side}/pulled"] = t2n(fn(bt, 8, gi.PAD, gi.EOT))
    # 1-D token input -> (1, T*bpt)  (data_creation.py:66-67)
    out["real/left/padded_1d"] = t2n(dc.tokens_to_bytes(torch.from_numpy(toks[0]), ttb_embedding(tl)))
    # bpt 16 table derived from the real one, both pulls on the left-padded bytes
    t16 = gi.widen_left_pad(tl, 16)
    bt = dc.tokens_to_bytes(torch.from_numpy(toks), ttb_embedding(t16))
    out["real16/left/padded"] = t2n(bt)
    out["real16/left/pulled"] = t2n(dc.pull_from_left(bt, 16, gi.PAD, gi.EOT))

    # G2: synthetic tables
    for name, bpt, B, T, vocab, seed in gi.SYNTH_INDEX_CASES:
        toks = gi.edge_tokens(seed, B, T, vocab)
        out[f"{name}/tokens"] = toks
        for side, fn in (("left", dc.pull_from_left), ("right", dc.pull_from_right)):
            tab = gi.synth_ttb(seed + 1000, vocab, bpt, side)
            bt = dc.tokens_to_bytes(torch.from_numpy(toks), ttb_embedding(tab))
            out[f"{name}/{side}/padded"] = t2n(bt)
            out[f"{name}/{side}/pulled"] = t2n(fn(bt, bpt, gi.PAD, gi.EOT))
            # the "wrong-side" pull too: pull_from_right on left-padded bytes etc. is legal
            other = dc.pull_from_right if side == "left" else dc.pull_from_left
            out[f"{name}/{side}/pulled_other"] = t2n(other(bt, bpt, gi.PAD, gi.EOT))
    # raw int64 byte tensors
    for name, bpt, B, Tr, seed in gi.RAW_INDEX_CASES:
        x = gi.raw_byte_tensor(seed, B, Tr, bpt)
        out[f"{name}/in"] = x
        out[f"{name}/left"] = t2n(dc.pull_from_left(torch.from_numpy(x), bpt, gi.PAD, gi.EOT))
        out[f"{name}/right"] = t2n(dc.pull_from_right(torch.from_numpy(x), bpt, gi.PAD, gi.EOT))
    # T == 0 (data_creation.py:82-83, 190)
    z = torch.zeros((2, 0), dtype=torch.int64)
    out["empty/left"] = t2n(dc.pull_from_left(z, 8, gi.PAD, gi.EOT))
    out["empty/right"] = t2n(dc.pull_from_right(z, 8, gi.PAD, gi.EOT))
    np.savez_compressed(OUT / "index.npz", **out)
    print("index:", len(out), "arrays")


def gen_make_embedding_quirk(dc):
    """make_embedding on the only table file present: row 50256 is absent from the JSON and
    keeps its random-normal init (SURVEY section 7 quirk i).  Seeded, so a drop-in that draws
    from torch's RNG in the same order reproduces the same garbage row."""
    cwd = os.getcwd()
    os.chdir(REF / "modded-nanogpt")
    try:
        torch.manual_seed(1234)
        emb = dc.make_embedding("ttb_8_left_pad.json", gi.GPT2_VOCAB)
    finally:
        os.chdir(cwd)
    toks = torch.tensor([[0, 50255, 50256, 1234]], dtype=torch.int32)
    bt = dc.tokens_to_bytes(toks, emb)
    np.savez_compressed(OUT / "make_embedding.npz", seed=np.int64(1234), tokens=t2n(toks), padded=t2n(bt),
                        eot_row_f32=t2n(emb.weight[50256]), weight_dtype=str(emb.weight.dtype),
                        requires_grad=np.bool_(emb.weight.requires_grad))
    print("make_embedding quirk row:", t2n(bt).reshape(4, 8)[2])


# ------------------------------------------------------------------ loader slice/shift + create_batch
def gen_loader(dc):
    out = {}
    bpt, vocab, seq, batch = 16, 512, 24, 4
    tab = gi.synth_ttb(3001, vocab, bpt, "left")
    tabr = gi.synth_ttb(3001, vocab, bpt, "right")
    data = gi.edge_tokens(301, 1, 3 * batch * (seq + 1) + 7, vocab).reshape(-1)
    out["data"] = data
    import functools
    ttb_in = ttb_embedding(tab)
    pull_in = functools.partial(dc.pull_from_left, bytes_per_token=bpt, pad_byte=456, eot_byte=457)
    create = load_create_data_TT_FF(dc, ttb_in, pull_in, bpt)
    d = torch.from_numpy(data)
    pos = batch * (seq + 1)  # second step of the generator
    for world in (1, 2, 4):
        for rank in range(world):
            # train_gpt.py:796-797, 804 (expression evaluated verbatim on CPU)
            local_seq_len = seq + 1
            local_batch_size = (batch * local_seq_len) // world
            tokens = d[pos + rank * local_batch_size:][:local_batch_size].view(-1, local_seq_len)
            toks_in, bytes_padded_in, bytes_pulled_in, targets = create(tokens)
            p = f"w{world}r{rank}"
            out[f"{p}/toks_in"] = t2n(toks_in)
            out[f"{p}/bytes_padded_in"] = t2n(bytes_padded_in)
            out[f"{p}/bytes_pulled_in"] = t2n(bytes_pulled_in)
            out[f"{p}/targets"] = t2n(targets)
    out["pos"] = np.int64(pos); out["batch"] = np.int64(batch); out["seq"] = np.int64(seq)
    # every variant of the dispatch table (train_gpt.py:686-783): inputs left-padded + pulled from the left, targets right-padded +
    # pulled from the right (so that a mixed-up table or direction cannot pass); rank 1 of 2 and the whole batch
    ttb_out = ttb_embedding(tabr)
    pull_out = functools.partial(dc.pull_from_right, bytes_per_token=bpt, pad_byte=456, eot_byte=457)
    variants = load_create_data_variants(dc, ttb_in, ttb_out, pull_in, pull_out, bpt)
    names = []
    for world, rank in ((1, 0), (2, 1)):
        local_batch_size = (batch * (seq + 1)) // world
        tokens = d[pos + rank * local_batch_size:][:local_batch_size].view(-1, seq + 1)
        for name, fn in sorted(variants.items()):
            res = fn(tokens)
            key = name[len("_create_data_from_toks_"):]
            names.append(key)
            for what, val in zip(("toks_in", "bytes_padded_in", "bytes_pulled_in", "targets"), res):
                if val is not None:                      # a missing key IS the reference's None
                    out[f"variants/w{world}r{rank}/{key}/{what}"] = t2n(val)
    out["variants/names"] = np.array(sorted(set(names)))
    # create_batch (data_creation.py:308-330)
    toks = torch.from_numpy(gi.edge_tokens(302, 3, 40, vocab))
    full = dc.create_batch(toks, bpt, 456, 457, ttb_embedding(tabr), ttb_embedding(tab))
    out["create_batch/tokens"] = t2n(toks)
    out["create_batch/full"] = t2n(full)
    np.savez_compressed(OUT / "loader.npz", **out)
    print("loader:", len(out), "arrays")


# ------------------------------------------------------------------ G3: scaled-pre-train float path
SCALED_CASES = [
    # name, Vt, Dt, Db, Dm, bpt, B, T, seed
    ("small", 97, 32, 8, 64, 8, 2, 16, 401),
    ("c2dims", 512, 256, 32, 768, 16, 1, 48, 402),
]


def scaled_inputs(dc_or_none, name, Vt, Dt, Db, Dm, bpt, B, T, seed):
    """ids through the ORACLE-independent path: table + edge tokens; pulled ids are
    produced by whoever calls (reference here, oracle/HIP in tests)."""
    tab = gi.synth_ttb(seed + 1000, Vt, bpt, "left")
    toks = gi.edge_tokens(seed, B, T, Vt, eot_p=0.08)
    return tab, toks


def gen_scaled(dc, tg):
    out = {}
    for (name, Vt, Dt, Db, Dm, bpt, B, T, seed) in SCALED_CASES:
        tab, toks = scaled_inputs(dc, name, Vt, Dt, Db, Dm, bpt, B, T, seed)
        padded = dc.tokens_to_bytes(torch.from_numpy(toks), ttb_embedding(tab))
        pulled = dc.pull_from_left(padded, bpt, gi.PAD, gi.EOT)
        Et = gi.normal_table(seed + 1, Vt, Dt)
        Eb = gi.normal_table(seed + 2, gi.BYTE_VOCAB, Db)
        W = gi.casted_linear_weight(seed + 3, Dm, Dt + bpt * Db)
        out[f"{name}/tokens"] = toks
        out[f"{name}/padded"] = t2n(padded)
        out[f"{name}/pulled"] = t2n(pulled)
        for dt_name, tdt in (("f32", torch.float32), ("f64", torch.float64)):
            for mode, bp_kw in (
                ("padded", dict(pull_in=False)),
                ("pulled", dict(pull_in=True, add_padded_and_pulled=False)),
                ("padded_and_pulled", dict(pull_in=True, add_padded_and_pulled=True)),
            ):
                bp = tg["ByteHyperparameters"](bytes_per_token=bpt, vocab_size=gi.BYTE_VOCAB,
                                               byte_mixin_method="concat", **bp_kw)
                dims = tg["ModelDims"](model_dim=Dm, byte_dim=Db, token_dim=Dt)
                emb = tg["FlexibleEmbedding"](dims, Vt, bp)
                mix = tg["ByteMixin"](dims, T, bp)
                emb.embed_tokens.weight.data = torch.from_numpy(Et).to(tdt)
                emb.embed_bytes.weight.data = torch.from_numpy(Eb).to(tdt)
                mix.mixin.mixin.weight.data = torch.from_numpy(W).to(tdt)
                assert list(dict(emb.state_dict())) == ["embed_tokens.weight", "embed_bytes.weight"]
                assert list(dict(mix.state_dict())) == ["mixin.mixin.weight"]
                with torch.no_grad():
                    xt, xb = emb(tokens=torch.from_numpy(toks), byte_tensor=padded, byte_tensor_pulled=pulled)
                    x = mix(xt, xb)
                if name == "small":
                    out[f"{name}/{mode}/{dt_name}/tok_embs"] = t2n(xt)
                    out[f"{name}/{mode}/{dt_name}/byte_embs"] = t2n(xb)
                if name == "small" or mode == "pulled":
                    out[f"{name}/{mode}/{dt_name}/x"] = t2n(x)
            # noop: token table has model_dim columns (train_gpt.py:330)
            bp = tg["ByteHyperparameters"](bytes_per_token=bpt, vocab_size=gi.BYTE_VOCAB, byte_mixin_method="noop")
            dims = tg["ModelDims"](model_dim=Dt, byte_dim=Db, token_dim=Dt)
            emb = tg["FlexibleEmbedding"](dims, Vt, bp)
            mix = tg["ByteMixin"](dims, T, bp)
            emb.embed_tokens.weight.data = torch.from_numpy(Et).to(tdt)
            with torch.no_grad():
                xt, xb = emb(tokens=torch.from_numpy(toks), byte_tensor=None, byte_tensor_pulled=None)
                assert xb is None
                out[f"{name}/noop/{dt_name}/x"] = t2n(mix(xt, xb))
    np.savez_compressed(OUT / "float_scaled.npz", **out)
    print("float_scaled:", len(out), "arrays")


# ------------------------------------------------------------------ G5: SUM modes (modded-nanogpt 71 family)
SUM_CASES = [
    # name, Vt, D, Db, bpt, T, seed
    ("small", 97, 64, 8, 8, 40, 501),
    ("c2dims", 512, 768, 48, 16, 48, 502),
]


def gen_sum(dc, r71):
    out = {}
    norm, mixin_bytes = r71["norm"], r71["mixin_bytes"]
    for (name, Vt, D, Db, bpt, T, seed) in SUM_CASES:
        tab = gi.synth_ttb(seed + 1000, Vt, bpt, "left")
        toks = gi.edge_tokens(seed, 1, T, Vt, eot_p=0.08)
        padded = dc.tokens_to_bytes(torch.from_numpy(toks), ttb_embedding(tab))
        pulled = dc.pull_from_left(padded, bpt, gi.PAD, gi.EOT)        # (1, T*bpt)
        # PER-TOKEN byte layout (bpt, T): slot k of every token -- the semantics of
        # train_gpt.py:442 / runs/7*.py:227-231, not the .view(16,-1) reshape of runs/71*.py:479
        byte_inputs = pulled.view(T, bpt).t().contiguous()
        Et = gi.normal_table(seed + 1, Vt, D)
        Eb = gi.normal_table(seed + 2, gi.BYTE_VOCAB, Db)
        s_tok, s_byte = 1.25, 0.75
        out[f"{name}/tokens"] = toks
        out[f"{name}/pulled"] = t2n(pulled)
        out[f"{name}/scales"] = np.array([s_tok, s_byte])
        for dt_name, tdt in (("f32", torch.float32), ("f64", torch.float64)):
            et, eb = torch.from_numpy(Et).to(tdt), torch.from_numpy(Eb).to(tdt)
            tok1d = torch.from_numpy(toks[0]).long()
            with torch.no_grad():
                # runs/71_mot-in_toks-valemb.py:312-314
                x_toks = et[tok1d][None]
                x_bytes = eb[byte_inputs].squeeze()
                out[f"{name}/r71/{dt_name}"] = t2n(mixin_bytes(x_toks, x_bytes))
                # runs/71041_mot-in_toks-valemb.py:311-313 (scalars are learned; fixed here)
                x_toks = norm(et[tok1d][None]) * s_tok
                x_bytes = norm(eb[byte_inputs].squeeze()) * s_byte
                out[f"{name}/r71041/{dt_name}"] = t2n(mixin_bytes(x_toks, x_bytes))
                # runs/71081_mot-in_toks-valemb.py:302-304,315
                x0t = norm(et[tok1d][None])
                x0b = norm(eb[byte_inputs].squeeze())
                x0b = torch.cat([b for b in x0b], dim=-1)[None]
                out[f"{name}/r71081/{dt_name}"] = t2n(x0t * s_tok + x0b * s_byte)
    np.savez_compressed(OUT / "sum_modes.npz", **out)
    print("sum_modes:", len(out), "arrays")


# ------------------------------------------------------------------ G4: mathblations C1
def gen_mathblations(mmodel, mdata):
    out = {}
    random.seed(0)
    gen = mdata.GenerateEquations()
    assert gen.vocab_size == 1003 and gen.max_possible_num_tokens == 33
    xs, xd, alltok, alldig = [], [], [], []
    for _ in range(8):
        x_tokens, x_digit_tokens, y_tokens, y_digit_tokens, _, _ = gen()
        xs.append(x_tokens); xd.append(x_digit_tokens)
        full = torch.cat([x_tokens, y_tokens[-1:]])
        alltok.append(full); alldig.append(gen.tokens_to_digits(full))
    x_tokens = torch.stack(xs); x_digits = torch.stack(xd)
    out["x_tokens"] = t2n(x_tokens); out["x_digit_tokens"] = t2n(x_digits)
    out["all_tokens"] = t2n(torch.stack(alltok)); out["all_digits"] = t2n(torch.stack(alldig))
    # every token id through tokens_to_digits (the full 1003 x 3 table)
    out["digit_table"] = t2n(gen.tokens_to_digits(torch.arange(gen.vocab_size))).reshape(gen.vocab_size, 3)
    D = 256
    for tied, mixout in ((True, "noop"),):
        cfg = mmodel.GPTConfig(vocab_size=gen.vocab_size, n_layer=1, n_head=2, n_embd_tok=D, n_embd_digit=D,
                               T=gen.max_possible_num_tokens - 1, length_factor=3,
                               digit_mixin_method="concat", digit_mixout_method=mixout)
        net = mmodel.GPT(cfg)
        assert net.wte.weight is net.lm_head.weight  # model.py:316-317
        Wt = gi.normal_table(601, gen.vocab_size, D)
        Wd = gi.normal_table(602, 14, D)
        Wf, bf = gi.linear_weight_bias(603, D, D + 3 * D)
        for dt_name, tdt in (("f32", torch.float32), ("f64", torch.float64)):
            net = net.to(tdt)
            net.lm_head.weight.data = torch.from_numpy(Wt).to(tdt)   # tied: wte sees it
            net.dte.weight.data = torch.from_numpy(Wd).to(tdt)
            net.digit_mixin.fc.weight.data = torch.from_numpy(Wf).to(tdt)
            net.digit_mixin.fc.bias.data = torch.from_numpy(bf).to(tdt)
            with torch.no_grad():
                we = net.wte(x_tokens)            # model.py:323
                de = net.dte(x_digits)            # model.py:326
                x = net.digit_mixin(we, de)       # model.py:327
            out[f"concat/{dt_name}/x"] = t2n(x)
    np.savez_compressed(OUT / "mathblations_c1.npz", **out)
    print("mathblations:", len(out), "arrays", "x", out["concat/f32/x"].shape)


# ------------------------------------------------------------------ gradients (autograd of the reference)
def gen_grads(dc, tg, r71, mmodel):
    """dL/dparams for L = sum(x * g), g seeded: what loss.backward() (train_gpt.py:1319, main.py:304)
    leaves in .grad for the parameters of the path.  Small shapes; fp32 and float64."""
    out = {}
    norm, mixin_bytes = r71["norm"], r71["mixin_bytes"]
    # SUM family (runs/71, 71041, 71081)
    name, Vt, D, Db, bpt, T, seed = SUM_CASES[0]
    tab = gi.synth_ttb(seed + 1000, Vt, bpt, "left")
    toks = gi.edge_tokens(seed, 1, T, Vt, eot_p=0.08)
    pulled = dc.pull_from_left(dc.tokens_to_bytes(torch.from_numpy(toks), ttb_embedding(tab)), bpt, gi.PAD, gi.EOT)
    byte_inputs = pulled.view(T, bpt).t().contiguous()
    g = np.random.RandomState(seed + 77).standard_normal((1, T, D))
    out["sum/g"] = g
    for dt_name, tdt in (("f32", torch.float32), ("f64", torch.float64)):
        for variant in ("r71", "r71041", "r71081"):
            et = torch.from_numpy(gi.normal_table(seed + 1, Vt, D)).to(tdt).requires_grad_()
            eb = torch.from_numpy(gi.normal_table(seed + 2, gi.BYTE_VOCAB, Db)).to(tdt).requires_grad_()
            sc = torch.tensor([0.75, 1.25], dtype=tdt, requires_grad=True)       # [-2] bytes, [-1] tokens
            tok1d = torch.from_numpy(toks[0]).long()
            if variant == "r71":
                x = mixin_bytes(et[tok1d][None], eb[byte_inputs].squeeze())
            elif variant == "r71041":
                x = mixin_bytes(norm(et[tok1d][None]) * sc[-1], norm(eb[byte_inputs].squeeze()) * sc[-2])
            else:
                x0b = norm(eb[byte_inputs].squeeze())
                x = norm(et[tok1d][None]) * sc[-1] + torch.cat([b for b in x0b], dim=-1)[None] * sc[-2]
            (x * torch.from_numpy(g).to(tdt)).sum().backward()
            out[f"sum/{variant}/{dt_name}/d_tok"] = t2n(et.grad)
            out[f"sum/{variant}/{dt_name}/d_byte"] = t2n(eb.grad)
            if variant != "r71":
                out[f"sum/{variant}/{dt_name}/d_scalars"] = t2n(sc.grad)
    # scaled-pre-train concat (FlexibleEmbedding + ByteMixinConcat), three byte modes + noop
    name, Vt, Dt, Db, Dm, bpt, B, T, seed = SCALED_CASES[0]
    tab, toks = scaled_inputs(dc, name, Vt, Dt, Db, Dm, bpt, B, T, seed)
    padded = dc.tokens_to_bytes(torch.from_numpy(toks), ttb_embedding(tab))
    pulled = dc.pull_from_left(padded, bpt, gi.PAD, gi.EOT)
    g = np.random.RandomState(seed + 77).standard_normal((B, T, Dm))
    out["scaled/g"] = g
    for dt_name, tdt in (("f32", torch.float32), ("f64", torch.float64)):
        for mode, bp_kw in (("padded", dict(pull_in=False)), ("pulled", dict(pull_in=True)),
                            ("padded_and_pulled", dict(pull_in=True, add_padded_and_pulled=True))):
            bp = tg["ByteHyperparameters"](bytes_per_token=bpt, vocab_size=gi.BYTE_VOCAB, byte_mixin_method="concat", **bp_kw)
            dims = tg["ModelDims"](model_dim=Dm, byte_dim=Db, token_dim=Dt)
            emb, mix = tg["FlexibleEmbedding"](dims, Vt, bp), tg["ByteMixin"](dims, T, bp)
            emb.embed_tokens.weight.data = torch.from_numpy(gi.normal_table(seed + 1, Vt, Dt)).to(tdt)
            emb.embed_bytes.weight.data = torch.from_numpy(gi.normal_table(seed + 2, gi.BYTE_VOCAB, Db)).to(tdt)
            mix.mixin.mixin.weight.data = torch.from_numpy(gi.casted_linear_weight(seed + 3, Dm, Dt + bpt * Db)).to(tdt)
            xt, xb = emb(tokens=torch.from_numpy(toks), byte_tensor=padded, byte_tensor_pulled=pulled)
            (mix(xt, xb) * torch.from_numpy(g).to(tdt)).sum().backward()
            out[f"scaled/{mode}/{dt_name}/d_tok"] = t2n(emb.embed_tokens.weight.grad)
            out[f"scaled/{mode}/{dt_name}/d_byte"] = t2n(emb.embed_bytes.weight.grad)
            out[f"scaled/{mode}/{dt_name}/d_W"] = t2n(mix.mixin.mixin.weight.grad)
        bp = tg["ByteHyperparameters"](bytes_per_token=bpt, vocab_size=gi.BYTE_VOCAB, byte_mixin_method="noop")
        emb = tg["FlexibleEmbedding"](tg["ModelDims"](model_dim=Dt, byte_dim=Db, token_dim=Dt), Vt, bp)
        emb.embed_tokens.weight.data = torch.from_numpy(gi.normal_table(seed + 1, Vt, Dt)).to(tdt)
        xt, _ = emb(tokens=torch.from_numpy(toks), byte_tensor=None, byte_tensor_pulled=None)
        gn = np.random.RandomState(seed + 78).standard_normal((B, T, Dt))
        out["scaled/g_noop"] = gn
        (xt * torch.from_numpy(gn).to(tdt)).sum().backward()
        out[f"scaled/noop/{dt_name}/d_tok"] = t2n(emb.embed_tokens.weight.grad)
    # mathblations DigitMixinConcat (bias, digits first), small dims, tied wte/lm_head
    D = 32
    random.seed(1)
    import data as mdata  # type: ignore
    gen = mdata.GenerateEquations()
    xs, xd = zip(*[(lambda r: (r[0], r[1]))(gen()) for _ in range(3)])
    x_tokens, x_digits = torch.stack(xs), torch.stack(xd)
    out["math/x_tokens"], out["math/x_digit_tokens"] = t2n(x_tokens), t2n(x_digits)
    g = np.random.RandomState(991).standard_normal((3, 32, D))
    out["math/g"] = g
    cfg = mmodel.GPTConfig(vocab_size=gen.vocab_size, n_layer=1, n_head=2, n_embd_tok=D, n_embd_digit=D, T=32,
                           length_factor=3, digit_mixin_method="concat")
    for dt_name, tdt in (("f32", torch.float32), ("f64", torch.float64)):
        net = mmodel.GPT(cfg).to(tdt)
        net.lm_head.weight.data = torch.from_numpy(gi.normal_table(611, gen.vocab_size, D)).to(tdt)
        net.dte.weight.data = torch.from_numpy(gi.normal_table(612, 14, D)).to(tdt)
        Wf, bf = gi.linear_weight_bias(613, D, 4 * D)
        net.digit_mixin.fc.weight.data = torch.from_numpy(Wf).to(tdt)
        net.digit_mixin.fc.bias.data = torch.from_numpy(bf).to(tdt)
        x = net.digit_mixin(net.wte(x_tokens), net.dte(x_digits))
        (x * torch.from_numpy(g).to(tdt)).sum().backward()
        out[f"math/{dt_name}/d_tok"] = t2n(net.wte.weight.grad)
        out[f"math/{dt_name}/d_byte"] = t2n(net.dte.weight.grad)
        out[f"math/{dt_name}/d_W"] = t2n(net.digit_mixin.fc.weight.grad)
        out[f"math/{dt_name}/d_bias"] = t2n(net.digit_mixin.fc.bias.grad)
    np.savez_compressed(OUT / "grads.npz", **out)
    print("grads:", len(out), "arrays")


# ------------------------------------------------------------------ bf16 (the dtype the training loop runs)
def gen_bf16(dc, tg, r71):  # noqa: C901
    """The same modules with bf16 tables, eager on CPU (train_gpt.py:1124-1126 casts nn.Embedding to bf16;
    runs/7*.py do the same).  Outputs are bf16, stored widened to float32 (exact)."""
    out = {}
    norm, mixin_bytes = r71["norm"], r71["mixin_bytes"]
    name, Vt, D, Db, bpt, T, seed = SUM_CASES[1]
    tab = gi.synth_ttb(seed + 1000, Vt, bpt, "left")
    toks = gi.edge_tokens(seed, 1, T, Vt, eot_p=0.08)
    pulled = dc.pull_from_left(dc.tokens_to_bytes(torch.from_numpy(toks), ttb_embedding(tab)), bpt, gi.PAD, gi.EOT)
    byte_inputs = pulled.view(T, bpt).t().contiguous()
    et = torch.from_numpy(gi.normal_table(seed + 1, Vt, D)).bfloat16()
    eb = torch.from_numpy(gi.normal_table(seed + 2, gi.BYTE_VOCAB, Db)).bfloat16()
    tok1d = torch.from_numpy(toks[0]).long()
    with torch.no_grad():
        out["sum/r71"] = t2n(mixin_bytes(et[tok1d][None], eb[byte_inputs].squeeze()).float())
        sc = torch.tensor([0.75, 1.25])
        out["sum/r71041"] = t2n(mixin_bytes(norm(et[tok1d][None]) * sc[-1], norm(eb[byte_inputs].squeeze()) * sc[-2]).float())
        out["sum/noop"] = t2n(norm(et[tok1d][None]).float())
        out["sum/byte_embs"] = t2n(norm(eb[pulled.view(-1)]).float())
    # concat mixin in the production dtypes: bf16 nn.Embedding tables (train_gpt.py:1124-1126), fp32 CastedLinear master weight
    # cast to the activations' dtype per call (185-186), eager on CPU.  x is bf16, stored widened (exact).
    for (cname, Vt2, Dt2, Db2, Dm2, bpt2, B2, T2, seed2) in SCALED_CASES[:1] + [("c2row", 512, 256, 32, 768, 16, 1, 40, 403)]:
        tab2 = gi.synth_ttb(seed2 + 1000, Vt2, bpt2, "left")
        toks2 = gi.edge_tokens(seed2, B2, T2, Vt2, eot_p=0.08)
        padded2 = dc.tokens_to_bytes(torch.from_numpy(toks2), ttb_embedding(tab2))
        pulled2 = dc.pull_from_left(padded2, bpt2, gi.PAD, gi.EOT)
        bp = tg["ByteHyperparameters"](bytes_per_token=bpt2, vocab_size=gi.BYTE_VOCAB, byte_mixin_method="concat", pull_in=True)
        dims = tg["ModelDims"](model_dim=Dm2, byte_dim=Db2, token_dim=Dt2)
        emb, mix = tg["FlexibleEmbedding"](dims, Vt2, bp), tg["ByteMixin"](dims, T2, bp)
        emb.embed_tokens.weight.data = torch.from_numpy(gi.normal_table(seed2 + 1, Vt2, Dt2))
        emb.embed_bytes.weight.data = torch.from_numpy(gi.normal_table(seed2 + 2, gi.BYTE_VOCAB, Db2))
        mix.mixin.mixin.weight.data = torch.from_numpy(gi.casted_linear_weight(seed2 + 3, Dm2, Dt2 + bpt2 * Db2)).float()
        for m in emb.modules():
            if isinstance(m, torch.nn.Embedding):
                m.bfloat16()
        with torch.no_grad():
            xt, xb = emb(tokens=torch.from_numpy(toks2), byte_tensor=padded2, byte_tensor_pulled=pulled2)
            x = mix(xt, xb)
        assert x.dtype == torch.bfloat16
        out[f"concat/{cname}/tokens"], out[f"concat/{cname}/pulled"] = toks2, t2n(pulled2)
        out[f"concat/{cname}/x"] = t2n(x.float())
    np.savez_compressed(OUT / "bf16.npz", **out)
    print("bf16:", len(out), "arrays")


# ------------------------------------------------------------------ G6: cross-attention mixin (train_gpt.py:243-300, 446-464)
def gen_cross_attn(dc, tg):
    out = {}
    for (name, Vt, D, bpt, T, seed) in gi.CROSS_CASES:
        tab = gi.synth_ttb(seed + 1000, Vt, bpt, "left")
        toks = gi.edge_tokens(seed, 1, T, Vt, eot_p=0.08)
        padded = dc.tokens_to_bytes(torch.from_numpy(toks), ttb_embedding(tab))
        pulled = dc.pull_from_left(padded, bpt, gi.PAD, gi.EOT)
        Et, Eb = gi.normal_table(seed + 1, Vt, D), gi.normal_table(seed + 2, gi.BYTE_VOCAB, D)
        q_w, kv_w, p_w = gi.cross_weights(seed + 3, D)
        out[f"{name}/tokens"], out[f"{name}/padded"], out[f"{name}/pulled"] = toks, t2n(padded), t2n(pulled)
        for mode, bp_kw in (("pulled", dict(pull_in=True)), ("padded_and_pulled", dict(pull_in=True, add_padded_and_pulled=True))):
            for dt_name, tdt in (("f32", torch.float32), ("f64", torch.float64)):
                bp = tg["ByteHyperparameters"](bytes_per_token=bpt, vocab_size=gi.BYTE_VOCAB, byte_mixin_method="cross_attn", **bp_kw)
                dims = tg["ModelDims"](model_dim=D, byte_dim=D, token_dim=D)
                emb = tg["FlexibleEmbedding"](dims, Vt, bp)
                mix = tg["ByteMixin"](dims, T, bp)
                assert sorted(dict(mix.state_dict())) == ["mixin.mixin.c_proj.weight", "mixin.mixin.kv_w", "mixin.mixin.lambda_factor",
                                                          "mixin.mixin.q_w"]
                if dt_name == "f32" and mode == "pulled":   # the module's own Rotary buffers (non-persistent)
                    out[f"{name}/cos_q"], out[f"{name}/sin_q"] = t2n(mix.mixin.mixin.rotary_q.cos), t2n(mix.mixin.mixin.rotary_q.sin)
                    out[f"{name}/cos_k"], out[f"{name}/sin_k"] = t2n(mix.mixin.mixin.rotary_k.cos), t2n(mix.mixin.mixin.rotary_k.sin)
                emb, mix = emb.to(tdt), mix.to(tdt)
                emb.embed_tokens.weight.data = torch.from_numpy(Et).to(tdt)
                emb.embed_bytes.weight.data = torch.from_numpy(Eb).to(tdt)
                ca = mix.mixin.mixin
                ca.q_w.data, ca.kv_w.data = torch.from_numpy(q_w).to(tdt), torch.from_numpy(kv_w).to(tdt)
                ca.c_proj.weight.data = torch.from_numpy(p_w).to(tdt)
                ca.lambda_factor.data = torch.tensor(0.7, dtype=tdt)
                with torch.no_grad():
                    xt, xb = emb(tokens=torch.from_numpy(toks), byte_tensor=padded, byte_tensor_pulled=pulled)
                    x = mix(xt, xb)
                assert x.shape == (1, T, D)
                out[f"{name}/{mode}/{dt_name}/x"] = t2n(x)
    np.savez_compressed(OUT / "cross_attn.npz", **out)
    print("cross_attn:", len(out), "arrays")
    gen_cross_attn_grads(dc, tg)


def gen_cross_attn_grads(dc, tg):
    """loss.backward() through FlexibleEmbedding + ByteMixin(cross_attn) in float64 (train_gpt.py:1319): gradients of
    every parameter for a fixed upstream gradient.  Table gradients are stored for the rows that were touched only."""
    out = {}
    for (name, Vt, D, bpt, T, seed), modes in ((gi.CROSS_CASES[0], ("pulled", "padded_and_pulled")), (gi.CROSS_CASES[1], ("pulled",))):
        tab = gi.synth_ttb(seed + 1000, Vt, bpt, "left")
        toks = gi.edge_tokens(seed, 1, T, Vt, eot_p=0.08)
        padded = dc.tokens_to_bytes(torch.from_numpy(toks), ttb_embedding(tab))
        pulled = dc.pull_from_left(padded, bpt, gi.PAD, gi.EOT)
        Et, Eb = gi.normal_table(seed + 1, Vt, D), gi.normal_table(seed + 2, gi.BYTE_VOCAB, D)
        q_w, kv_w, p_w = gi.cross_weights(seed + 3, D)
        g = np.random.RandomState(seed + 9).standard_normal((1, T, D))
        out[f"{name}/g"] = g.astype(np.float32)
        for mode in modes:
            bp = tg["ByteHyperparameters"](bytes_per_token=bpt, vocab_size=gi.BYTE_VOCAB, byte_mixin_method="cross_attn", pull_in=True,
                                           add_padded_and_pulled=mode == "padded_and_pulled")
            dims = tg["ModelDims"](model_dim=D, byte_dim=D, token_dim=D)
            emb, mix = tg["FlexibleEmbedding"](dims, Vt, bp).double(), tg["ByteMixin"](dims, T, bp).double()
            emb.embed_tokens.weight.data = torch.from_numpy(Et).double()
            emb.embed_bytes.weight.data = torch.from_numpy(Eb).double()
            ca = mix.mixin.mixin
            ca.q_w.data, ca.kv_w.data = torch.from_numpy(q_w).double(), torch.from_numpy(kv_w).double()
            ca.c_proj.weight.data = torch.from_numpy(p_w).double()
            ca.lambda_factor.data = torch.tensor(0.7, dtype=torch.float64)
            xt, xb = emb(tokens=torch.from_numpy(toks), byte_tensor=padded, byte_tensor_pulled=pulled)
            x = mix(xt, xb)
            (x * torch.from_numpy(g.astype(np.float32)).double()).sum().backward()
            for key, p_ in (("d_tok", emb.embed_tokens.weight), ("d_byte", emb.embed_bytes.weight)):
                gr = t2n(p_.grad)
                rows = np.flatnonzero(np.abs(gr).sum(1))
                out[f"{name}/{mode}/{key}_rows"] = rows.astype(np.int32)
                out[f"{name}/{mode}/{key}_vals"] = gr[rows].astype(np.float32)
            out[f"{name}/{mode}/d_qw"] = t2n(ca.q_w.grad).astype(np.float32)
            out[f"{name}/{mode}/d_kvw"] = t2n(ca.kv_w.grad).astype(np.float32)
            out[f"{name}/{mode}/d_pw"] = t2n(ca.c_proj.weight.grad).astype(np.float32)
            out[f"{name}/{mode}/d_lambda"] = np.array([float(ca.lambda_factor.grad)])
    np.savez_compressed(OUT / "cross_attn_grads.npz", **out)
    print("cross_attn_grads:", len(out), "arrays")


def gen_digit_cross_attn(mmodel, mdata):
    """mathblations DigitMixinCrossAttention (model.py:239-253 -> CrossAttention 89-154), use_digit_self_attn=False:
    forward in fp32 / float64 and the float64 autograd gradients.  CrossAttention.__init__ builds its block mask with
    create_block_mask's default device ("cuda", model.py:119-121), which does not exist on this host: the name is rebound
    to the same torch function with device="cpu" for the construction; flex_attention then runs its eager CPU path."""
    import functools
    from torch import nn
    mmodel.create_block_mask = functools.partial(mmodel.create_block_mask, device="cpu")
    out = {}
    for (name, mdpt, mtpn, D, H, B, seed) in gi.DIGIT_CROSS_CASES:
        random.seed(seed)
        gen = mdata.GenerateEquations(max_digits_per_token=mdpt, max_tokens_per_num=mtpn)
        xs, xd = [], []
        for _ in range(B):
            x_tokens, x_digit_tokens, *_ = gen()
            xs.append(x_tokens); xd.append(x_digit_tokens)
        x_tokens, x_digits = torch.stack(xs), torch.stack(xd)
        T = x_tokens.shape[1]
        assert x_digits.shape == (B, T * mdpt) and T == gen.max_possible_num_tokens - 1
        out[f"{name}/x_tokens"], out[f"{name}/x_digit_tokens"] = t2n(x_tokens), t2n(x_digits)
        Wt, Wd = gi.normal_table(seed + 1, gen.vocab_size, D), gi.normal_table(seed + 2, 14, D)
        ws = gi.digit_cross_weights(seed + 3, D)
        g = np.random.RandomState(seed + 9).standard_normal((B, T, D)).astype(np.float32)
        out[f"{name}/g"] = g
        for dt_name, tdt in (("f32", torch.float32), ("f64", torch.float64)):
            cfg = mmodel.GPTConfig(vocab_size=gen.vocab_size, n_layer=1, n_head=H, n_embd_tok=D, n_embd_digit=D, T=T + 1,
                                   length_factor=mdpt, digit_mixin_method="cross_attn", digit_mixout_method="noop")
            wte, dte = nn.Embedding(gen.vocab_size, D).to(tdt), nn.Embedding(14, D).to(tdt)   # model.py:304-305
            mix = mmodel.make_digit_mixin(cfg).to(tdt)
            assert sorted(dict(mix.state_dict())) == ["cross_attn.c_k.weight", "cross_attn.c_proj.weight", "cross_attn.c_q.weight",
                                                      "cross_attn.c_v.weight"]
            wte.weight.data, dte.weight.data = torch.from_numpy(Wt).to(tdt), torch.from_numpy(Wd).to(tdt)
            ca = mix.cross_attn
            for lin, w in zip((ca.c_q, ca.c_k, ca.c_v, ca.c_proj), ws):
                lin.weight.data = torch.from_numpy(w).to(tdt)
            we, de = wte(x_tokens), dte(x_digits)          # model.py:323, 326
            x = mix(we, de)                                # model.py:327
            assert x.shape == (B, T, D) and x.dtype == tdt
            out[f"{name}/{dt_name}/x"] = t2n(x)
            if dt_name == "f32":   # the Rotary module's cached tables after the call (model.py:40-49): bf16 values
                (cq, sq), (ck, sk) = ca.rotary(torch.empty(1, T, 1, 1)), ca.rotary(torch.empty(1, T * mdpt, 1, 1))
                out[f"{name}/cos_q"], out[f"{name}/sin_q"] = t2n(cq[0, :, 0].float()), t2n(sq[0, :, 0].float())
                out[f"{name}/cos_k"], out[f"{name}/sin_k"] = t2n(ck[0, :, 0].float()), t2n(sk[0, :, 0].float())
                continue
            (x * torch.from_numpy(g).double()).sum().backward()
            for key, p_ in (("d_tok", wte.weight), ("d_digit", dte.weight)):
                gr = t2n(p_.grad)
                rows = np.flatnonzero(np.abs(gr).sum(1))
                out[f"{name}/{key}_rows"] = rows.astype(np.int32)
                out[f"{name}/{key}_vals"] = gr[rows].astype(np.float32)
            for key, lin in (("d_cq", ca.c_q), ("d_ck", ca.c_k), ("d_cv", ca.c_v), ("d_cproj", ca.c_proj)):
                out[f"{name}/{key}"] = t2n(lin.weight.grad).astype(np.float32)
    np.savez_compressed(OUT / "digit_cross_attn.npz", **out)
    print("digit_cross_attn:", len(out), "arrays")


def main():
    if sys.argv[1:] == ["cross_attn"]:          # regenerate one fixture without touching the others
        gen_cross_attn(load_data_creation(), load_train_gpt_defs())
        return
    if sys.argv[1:] == ["digit_cross_attn"]:
        gen_digit_cross_attn(*load_mathblations())
        return
    if sys.argv[1:] == ["loader"]:
        gen_loader(load_data_creation())
        return
    if sys.argv[1:] == ["bf16"]:
        gen_bf16(load_data_creation(), load_train_gpt_defs(), load_run71_defs())
        return
    dc = load_data_creation()
    gen_ttb_fixture()
    gen_index(dc)
    gen_make_embedding_quirk(dc)
    gen_loader(dc)
    gen_scaled(dc, load_train_gpt_defs())
    gen_sum(dc, load_run71_defs())
    mm = load_mathblations()
    gen_mathblations(*mm)
    gen_grads(dc, load_train_gpt_defs(), load_run71_defs(), mm[0])
    gen_bf16(dc, load_train_gpt_defs(), load_run71_defs())
    gen_cross_attn(dc, load_train_gpt_defs())
    gen_digit_cross_attn(*mm)
    meta = dict(torch=torch.__version__, numpy=np.__version__, python=sys.version.split()[0],
                threads=torch.get_num_threads(), reference="snimu/mixture-of-tokenizers @ 2025-08-24",
                generator="oracle/gen_golden.py")
    (OUT / "META.json").write_text(json.dumps(meta, indent=1) + "\n")
    total = sum(p.stat().st_size for p in OUT.glob("*"))
    print(f"fixtures: {total/1e6:.2f} MB in {OUT}")


if __name__ == "__main__":
    main()
