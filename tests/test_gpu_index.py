"""-m gpu: integer path of the HIP library vs the golden fixtures and the CPU oracle (bit-exact)."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import oracle as orc
from util_gpu import DEV, dev, host

pytestmark = pytest.mark.gpu
G = gi.GOLDEN_DIR


@pytest.fixture(scope="module")
def dc():
    from mixture_of_tokenizers_amd import data_creation
    return data_creation


@pytest.fixture(scope="module")
def index():
    return np.load(G / "index.npz")


def test_real_vocab_golden(dc, index):
    tl = gi.load_real_ttb8()
    tr = gi.to_right_pad(tl)
    toks = dev(index["real/tokens"])
    for side, tab, pull in (("left", tl, dc.pull_from_left), ("right", tr, dc.pull_from_right)):
        padded = dc.tokens_to_bytes(toks, dev(tab))
        assert padded.dtype == torch.int64 and padded.shape == (4, 64 * 8)
        np.testing.assert_array_equal(host(padded), index[f"real/{side}/padded"])
        np.testing.assert_array_equal(host(pull(padded, 8, gi.PAD, gi.EOT)), index[f"real/{side}/pulled"])
    np.testing.assert_array_equal(host(dc.tokens_to_bytes(toks[0], dev(tl))), index["real/left/padded_1d"])
    t16 = gi.widen_left_pad(tl, 16)
    padded = dc.tokens_to_bytes(toks, dev(t16))
    np.testing.assert_array_equal(host(dc.pull_from_left(padded, 16, gi.PAD, gi.EOT)), index["real16/left/pulled"])


@pytest.mark.parametrize("case", gi.SYNTH_INDEX_CASES, ids=lambda c: c[0])
def test_synth_golden(dc, index, case):
    name, bpt, B, T, vocab, seed = case
    toks = dev(gi.edge_tokens(seed, B, T, vocab))
    for side in ("left", "right"):
        tab = dev(gi.synth_ttb(seed + 1000, vocab, bpt, side))
        padded = dc.tokens_to_bytes(toks, tab)
        np.testing.assert_array_equal(host(padded), index[f"{name}/{side}/padded"])
        own = dc.pull_from_left if side == "left" else dc.pull_from_right
        other = dc.pull_from_right if side == "left" else dc.pull_from_left
        np.testing.assert_array_equal(host(own(padded, bpt, gi.PAD, gi.EOT)), index[f"{name}/{side}/pulled"])
        np.testing.assert_array_equal(host(other(padded, bpt, gi.PAD, gi.EOT)), index[f"{name}/{side}/pulled_other"])


@pytest.mark.parametrize("case", gi.RAW_INDEX_CASES, ids=lambda c: c[0])
def test_raw_golden(dc, index, case):
    name, bpt, B, Tr, seed = case
    x = dev(index[f"{name}/in"])
    np.testing.assert_array_equal(host(dc.pull_from_left(x, bpt, gi.PAD, gi.EOT)), index[f"{name}/left"])
    np.testing.assert_array_equal(host(dc.pull_from_right(x, bpt, gi.PAD, gi.EOT)), index[f"{name}/right"])


def test_edge_shapes_and_errors(dc):
    z = torch.zeros((2, 0), dtype=torch.int64, device=DEV)
    assert dc.pull_from_left(z, 8, gi.PAD, gi.EOT).shape == (2, 0)  # data_creation.py:190
    assert dc.pull_from_right(z, 8, gi.PAD, gi.EOT).shape == (2, 0)
    with pytest.raises(AssertionError):  # data_creation.py:85,192
        dc.pull_from_left(torch.zeros((1, 12), dtype=torch.int64, device=DEV), 8, gi.PAD, gi.EOT)
    # out-of-range token: flagged on the device, raised on check (nn.Embedding raises IndexError)
    import mixture_of_tokenizers_amd as mot
    tab = dev(gi.synth_ttb(1, 16, 8, "left"))
    dc.tokens_to_bytes(torch.tensor([[3, 99]], dtype=torch.int32, device=DEV), tab)
    with pytest.raises(IndexError):
        mot.check_status()
    mot.check_status()  # cleared


def test_loader_and_create_batch_golden(dc):
    z = np.load(G / "loader.npz")
    bpt, vocab = 16, 512
    tab, tabr = dev(gi.synth_ttb(3001, vocab, bpt, "left")), dev(gi.synth_ttb(3001, vocab, bpt, "right"))
    full = dc.create_batch(dev(z["create_batch/tokens"]), bpt, gi.PAD, gi.EOT, tabr, tab)
    assert full.dtype == torch.int64
    np.testing.assert_array_equal(host(full), z["create_batch/full"])


# ---- vs the oracle on seeded inputs, including tile-boundary and long-lookback cases
def test_tokens_to_digits_golden(dc):
    """mathblations/data.py:92-109, 160-175 on the device: the reference's digit ids for its own equations, bit for bit."""
    z = np.load(G / "mathblations_c1.npz")
    tab = dc.make_digit_table(3).to(DEV)
    got = dc.tokens_to_digits(dev(z["all_tokens"]), tab)
    assert got.dtype == torch.int64
    np.testing.assert_array_equal(host(got), z["all_digits"])
    np.testing.assert_array_equal(host(got)[:, :-3], z["x_digit_tokens"])          # inputs drop the last token's digits
    np.testing.assert_array_equal(host(dc.tokens_to_digits(dev(z["all_tokens"][0]), tab)), z["all_digits"][0])


@pytest.mark.parametrize("bpt,B,T,vocab,seed,eot_p", [
    (16, 8, 2048, 512, 7001, 1 / 700),     # C4-shaped rows (T=2048): 8 tiles per row
    (16, 3, 1000, 512, 7002, 0.02),        # ragged last tile
    (8, 5, 777, 97, 7003, 0.0),            # no EOT anywhere
    (32, 2, 513, 512, 7004, 0.3),          # EOT-dense
    (3, 8, 33, 64, 7005, 0.1),             # mathblations-sized slots
    (64, 2, 130, 512, 7006, 0.05),         # MOT_MAX_BPT
    (20, 4, 300, 512, 7007, 0.01),
])
def test_vs_oracle(dc, bpt, B, T, vocab, seed, eot_p):
    toks = gi.edge_tokens(seed, B, T, vocab, eot_p=eot_p)
    if eot_p == 0.0:
        toks[toks == vocab - 1] = 2
    for side in ("left", "right"):
        tab = gi.synth_ttb(seed + 1, vocab, bpt, side, mean_valid=min(4.4, bpt / 2))
        padded_ref = orc.tokens_to_bytes(toks, tab.astype(np.float32))
        padded = dc.tokens_to_bytes(dev(toks), dev(tab))
        np.testing.assert_array_equal(host(padded), padded_ref)
        np.testing.assert_array_equal(host(dc.pull_from_left(padded, bpt, gi.PAD, gi.EOT)),
                                      orc.pull_from_left(padded_ref, bpt, gi.PAD, gi.EOT))
        np.testing.assert_array_equal(host(dc.pull_from_right(padded, bpt, gi.PAD, gi.EOT)),
                                      orc.pull_from_right(padded_ref, bpt, gi.PAD, gi.EOT))


def test_long_lookback_across_tiles(dc):
    """Hundreds of consecutive empty tokens: the halo walk must cross several 64-token steps
    and several tiles to find the bytes a window needs (unbounded in tokens)."""
    bpt, vocab, T = 16, 64, 1500
    tab = gi.synth_ttb(42, vocab, bpt, "left")
    tabr = gi.to_right_pad(tab)
    toks = np.zeros((3, T), dtype=np.int32)            # id 0 = no valid byte
    toks[0, 5] = 7; toks[0, 700] = 9; toks[0, 1499] = 11
    toks[1, 0] = 5; toks[1, 1] = vocab - 1; toks[1, 1400] = 6
    toks[2, :] = 0
    for t_, p in ((tab, "left"), (tabr, "right")):
        ref = orc.tokens_to_bytes(toks, t_.astype(np.float32))
        got = dc.tokens_to_bytes(dev(toks), dev(t_))
        np.testing.assert_array_equal(host(dc.pull_from_left(got, bpt, gi.PAD, gi.EOT)), orc.pull_from_left(ref, bpt, gi.PAD, gi.EOT))
        np.testing.assert_array_equal(host(dc.pull_from_right(got, bpt, gi.PAD, gi.EOT)), orc.pull_from_right(ref, bpt, gi.PAD, gi.EOT))


def test_full_size_c4(dc):
    """BASELINE config 4 (B x T = 256 x 2048, bpt 16, GPT-2 vocab, FineWeb-shaped ids): whole-tensor
    equality with the oracle plus the size-independent properties of the pull."""
    B, T, bpt = 256, 2048, 16
    tab = gi.widen_left_pad(gi.load_real_ttb8(), bpt)
    toks = gi.fineweb_like_tokens(12345, B, T)
    padded = dc.tokens_to_bytes(dev(toks), dev(tab))
    pulled = dc.pull_from_left(padded, bpt, gi.PAD, gi.EOT)
    ref_padded = orc.tokens_to_bytes(toks, tab.astype(np.float32))
    ref_pulled = orc.pull_from_left(ref_padded, bpt, gi.PAD, gi.EOT)
    np.testing.assert_array_equal(host(padded), ref_padded)
    np.testing.assert_array_equal(host(pulled), ref_pulled)
    p3, q3 = padded.view(B, T, bpt), pulled.view(B, T, bpt)
    is_eot = (p3 == gi.EOT).all(-1)
    assert torch.equal(q3[is_eot], p3[is_eot])                       # EOT tokens keep their bytes
    nv_p, nv_q = (p3 != gi.PAD).sum(-1), (q3 != gi.PAD).sum(-1)
    assert bool((nv_q >= nv_p).all())                                # a pull never loses a token's own bytes
    own = p3[:, :, -1][~is_eot & (nv_p > 0)]
    assert torch.equal(q3[:, :, -1][~is_eot & (nv_p > 0)], own)      # window ends at the token's last byte
    # idempotence does not hold for pulls; sortedness of pads does: pads are a prefix of every token
    pad_mask = (q3 == gi.PAD)
    assert bool((pad_mask[:, :, 1:] <= pad_mask[:, :, :-1]).all())
    stats = orc.byte_stats(ref_padded, ref_pulled, gi.PAD)
    assert int((padded == gi.PAD).sum()) == stats[1] and int((pulled == gi.PAD).sum()) == stats[2]
    mean_valid = float(nv_p.float().mean())
    assert 3.4 <= mean_valid <= 5.4, mean_valid                      # 4.4 +- 1 chars/token


# ------------------------------------------------------------------------------------------------
# The byte-index work INSIDE the fused forward (mot_wave.hpp, round 2): every wave produces the ids of its own unit of 16 or 32
# tokens from a 64-token window, with no workgroup barrier.  Same cases as the standalone kernels above, through
# embed_mix(..., return_ids=True): padded and pulled ids bit-exact against the oracle, for both unit sizes (the launcher takes
# 32-token units from 131 072 tokens on), both pull directions and no pull, int16 and int32 tables, rows that cannot be read
# with 16-byte vectors (bpt 3, 20), windows that must walk outwards for their halo.
# ------------------------------------------------------------------------------------------------
def _fused_ids(toks, tab, bpt, pull, add_padded=False):
    import mixture_of_tokenizers_amd as mot
    Db = 4
    Et = torch.zeros((int(tab.shape[0]), bpt * Db), device=DEV)
    Eb = torch.zeros((gi.BYTE_VOCAB, Db), device=DEV)
    r = mot.embed_mix(dev(toks), Et, Eb, mode="sum", bpt=bpt, ttb=dev(tab), pull=pull, add_padded=add_padded, return_ids=True)
    mot.check_status()
    return host(r.ids_padded), host(r.ids_pulled)


@pytest.mark.parametrize("bpt,B,T,vocab,seed,eot_p", [
    (16, 8, 2048, 512, 7101, 1 / 700),     # 16 384 tokens: 16-token units
    (16, 64, 2048, 512, 7102, 1 / 700),    # 131 072 tokens: 32-token units
    (16, 3, 1000, 512, 7103, 0.02),        # ragged last unit (1000 = 62 * 16 + 8)
    (8, 5, 777, 97, 7104, 0.0),            # no EOT anywhere; rows of one 16-byte vector
    (32, 2, 513, 512, 7105, 0.3),          # EOT-dense; rows of four vectors (two cached, two re-read)
    (3, 8, 33, 64, 7106, 0.1),             # mathblations-sized slots: element loads; rows shorter than a window
    (64, 2, 130, 512, 7107, 0.05),         # MOT_MAX_BPT
    (20, 4, 300, 512, 7108, 0.01),         # 40-byte rows: element loads
    (16, 1, 1, 64, 7109, 0.0),             # a single token
    (16, 2, 17, 64, 7110, 0.2),
])
def test_fused_ids_vs_oracle(bpt, B, T, vocab, seed, eot_p):
    toks = gi.edge_tokens(seed, B, T, vocab, eot_p=eot_p)
    if eot_p == 0.0:
        toks[toks == vocab - 1] = 2
    for side, pull in (("left", "left"), ("right", "right"), ("left", None)):
        tab = gi.synth_ttb(seed + 1, vocab, bpt, side, mean_valid=min(4.4, bpt / 2))
        padded_ref = orc.tokens_to_bytes(toks, tab.astype(np.float32))
        pulled_ref = {"left": orc.pull_from_left, "right": orc.pull_from_right}[pull](padded_ref, bpt, gi.PAD, gi.EOT) if pull else padded_ref
        for t_ in (tab, tab.astype(np.int32)):
            padded, pulled = _fused_ids(toks, t_, bpt, pull, add_padded=(seed % 2 == 0))
            np.testing.assert_array_equal(padded, padded_ref)
            np.testing.assert_array_equal(pulled, pulled_ref)


@pytest.mark.parametrize("B,T", [(3, 1500), (90, 1500)])   # 16-token and 32-token units
def test_fused_ids_long_lookback(B, T):
    """Hundreds of consecutive empty tokens: a window's built-in halo (48 or 32 tokens) holds no byte, so the wave walks outwards,
    64 tokens per step, across many steps -- in both directions -- and stops at an EOT token or the row's end."""
    bpt, vocab = 16, 64
    tab = gi.synth_ttb(42, vocab, bpt, "left")
    tabr = gi.to_right_pad(tab)
    toks = np.zeros((B, T), dtype=np.int32)            # id 0 = no valid byte
    toks[0, 5] = 7; toks[0, 700] = 9; toks[0, 1499] = 11
    toks[1, 0] = 5; toks[1, 1] = vocab - 1; toks[1, 1400] = 6
    rs = np.random.RandomState(4)
    for b in range(3, B):                               # sparse rows: a byte-carrying token every ~200 positions, an EOT now and then
        at = rs.choice(T, 8, replace=False)
        toks[b, at] = rs.randint(1, vocab, 8)
    for t_, pull, fn in ((tab, "left", orc.pull_from_left), (tabr, "right", orc.pull_from_right)):
        ref = orc.tokens_to_bytes(toks, t_.astype(np.float32))
        padded, pulled = _fused_ids(toks, t_, bpt, pull)
        np.testing.assert_array_equal(padded, ref)
        np.testing.assert_array_equal(pulled, fn(ref, bpt, gi.PAD, gi.EOT))
