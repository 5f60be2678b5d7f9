"""Deterministic inputs shared by oracle/gen_golden.py (which runs the reference on
them, in the build container) and the tests (which run the oracle / the HIP path on
them, anywhere).  Only numpy's legacy RandomState is used: its streams are frozen
across numpy versions, so a seed names the same arrays here and on the GPU box.

Nothing in this file comes from the reference; constants (pad 456, eot 457, byte vocab
458, GPT-2 vocab 50257, EOT token = vocab-1) are the values SURVEY.md section 8 records.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

GOLDEN_DIR = Path(__file__).resolve().parent / "golden"

PAD, EOT, BYTE_VOCAB = 456, 457, 458
GPT2_VOCAB = 50257


# --------------------------------------------------------------------------- tables
def synth_ttb(seed: int, vocab: int, bpt: int, side: str, mean_valid: float = 4.4,
              pad: int = PAD, eot: int = EOT, byte_vocab: int = 456) -> np.ndarray:
    """Synthetic token->byte table (vocab, bpt) int16.

    Row lengths follow a clipped geometric-ish histogram with the FineWeb mean of
    ~4.4 valid chars/token and always include empty rows (0 valid) and full rows.
    The last row is the EOT token: [eot]*bpt (create_ttb.py:20-22 semantics).
    side = "left": valid bytes right-aligned (left-padded); "right": left-aligned.
    """
    rs = np.random.RandomState(seed)
    lens = np.clip(np.round(rs.gamma(2.0, mean_valid / 2.0, size=vocab)), 0, bpt).astype(np.int64)
    lens[0] = 0                      # a token with no valid byte at all
    lens[1 % vocab] = bpt            # a full token
    tab = np.full((vocab, bpt), pad, dtype=np.int16)
    chars = rs.randint(0, byte_vocab, size=(vocab, bpt)).astype(np.int16)
    for v in range(vocab):
        n = int(lens[v])
        if n == 0:
            continue
        if side == "left":
            tab[v, bpt - n:] = chars[v, :n]
        else:
            tab[v, :n] = chars[v, :n]
    tab[vocab - 1, :] = eot
    return tab


def load_real_ttb8() -> np.ndarray:
    """GPT-2 token->byte table, bpt 8, left-padded: (50257, 8) int16.

    Data fixture tests/golden/ttb_8_left_pad.npz (the reference's
    modded-nanogpt/embeddings/ttb_8_left_pad.json re-encoded as int16; rows 0..50255).
    Row 50256 (EOT) is absent from that file and set to [457]*8 as create_ttb.py:20-22
    would have written it (SURVEY section 7, quirk i).
    """
    z = np.load(GOLDEN_DIR / "ttb_8_left_pad.npz")
    tab = np.full((GPT2_VOCAB, 8), PAD, dtype=np.int16)
    tab[: z["rows"].shape[0]] = z["rows"]
    tab[GPT2_VOCAB - 1] = EOT
    return tab


def to_right_pad(tab_left: np.ndarray, pad: int = PAD) -> np.ndarray:
    """Right-padded variant of a left-padded table (valid bytes moved to the front)."""
    out = np.full_like(tab_left, pad)
    for v in range(tab_left.shape[0]):
        row = tab_left[v]
        valid = row[row != pad]
        out[v, : valid.size] = valid
    return out


def widen_left_pad(tab_left: np.ndarray, bpt: int, pad: int = PAD, eot: int = EOT) -> np.ndarray:
    """bpt-wide left-padded table derived from a narrower one (extra pads on the left);
    all-EOT rows stay all-EOT so the EOT token is still recognised."""
    v, b0 = tab_left.shape
    out = np.full((v, bpt), pad, dtype=tab_left.dtype)
    out[:, bpt - b0:] = tab_left
    is_eot = (tab_left == eot).all(axis=1)
    out[is_eot] = eot
    return out


# --------------------------------------------------------------------------- tokens
def fineweb_like_tokens(seed: int, B: int, T: int, vocab: int = GPT2_VOCAB, eot_p: float = 1.0 / 700,
                        uniform: bool = False) -> np.ndarray:
    """id = min(floor(Veff * u^3), vocab-2), then EOT (vocab-1) with p=1/700."""
    rs = np.random.RandomState(seed)
    u = rs.random_sample((B, T))
    if uniform:
        ids = np.floor(u * (vocab - 1))
    else:
        ids = np.floor((vocab - 1) * u ** 3)
    ids = np.minimum(ids, vocab - 2).astype(np.int32)
    ids[rs.random_sample((B, T)) < eot_p] = vocab - 1
    return ids


def edge_tokens(seed: int, B: int, T: int, vocab: int, eot_p: float = 0.1) -> np.ndarray:
    """Random ids with many EOTs and the position classes the survey lists: EOT at row
    start / end, twice in a row, rows without any EOT, zero-valid tokens (id 0)."""
    rs = np.random.RandomState(seed)
    ids = rs.randint(0, vocab - 1, size=(B, T)).astype(np.int32)
    ids[rs.random_sample((B, T)) < eot_p] = vocab - 1
    ids[rs.random_sample((B, T)) < 0.15] = 0          # empty tokens (synth tables: row 0)
    e = vocab - 1
    if T >= 1:
        ids[0, 0] = e
    if B >= 2 and T >= 1:
        ids[1, T - 1] = e
    if B >= 3 and T >= 4:
        ids[2, 1] = e
        ids[2, 2] = e
    if B >= 4:
        ids[3][ids[3] == e] = 1                       # a row with no EOT
    return ids


def raw_byte_tensor(seed: int, B: int, Tr: int, bpt: int, pad: int = PAD, eot: int = EOT) -> np.ndarray:
    """Arbitrary (B, Tr*bpt) int64 byte tensors that no table would produce: pads in the
    middle of tokens, tokens with some-but-not-all eot bytes, all-pad tokens, negative
    and >16-bit values -- pull_from_* are defined on any int64 input."""
    rs = np.random.RandomState(seed)
    x = rs.randint(0, 456, size=(B, Tr, bpt)).astype(np.int64)
    x[rs.random_sample((B, Tr, bpt)) < 0.5] = pad
    x[rs.random_sample((B, Tr, bpt)) < 0.05] = eot
    x[rs.random_sample((B, Tr, bpt)) < 0.02] = -3
    x[rs.random_sample((B, Tr, bpt)) < 0.02] = 70000
    x[rs.random_sample((B, Tr)) < 0.08] = eot         # whole EOT tokens
    x[rs.random_sample((B, Tr)) < 0.10] = pad         # whole empty tokens
    return x.reshape(B, Tr * bpt)


# --------------------------------------------------------------------------- floats
def normal_table(seed: int, rows: int, dim: int) -> np.ndarray:
    """~N(0,1) float64 master copy (nn.Embedding default init); cast to fp32 by callers."""
    return np.random.RandomState(seed).standard_normal((rows, dim))


def casted_linear_weight(seed: int, out_f: int, in_f: int) -> np.ndarray:
    """U(+-sqrt(3)*0.5/sqrt(in)) as CastedLinear.reset_parameters (train_gpt.py:179-183)."""
    bound = (3 ** 0.5) * 0.5 * in_f ** -0.5
    return np.random.RandomState(seed).uniform(-bound, bound, size=(out_f, in_f))


def linear_weight_bias(seed: int, out_f: int, in_f: int):
    """nn.Linear default init range U(+-1/sqrt(in)) for weight and bias (model.py:261)."""
    rs = np.random.RandomState(seed)
    b = in_f ** -0.5
    return rs.uniform(-b, b, size=(out_f, in_f)), rs.uniform(-b, b, size=(out_f,))


# --------------------------------------------------------------------------- case lists
# (name, bpt, B, T_tokens, vocab, seed)
SYNTH_INDEX_CASES = [
    ("b4_B1_T1", 4, 1, 1, 64, 101),
    ("b4_B3_T2", 4, 3, 2, 64, 102),
    ("b4_B4_T257", 4, 4, 257, 64, 103),
    ("b16_B1_T16", 16, 1, 16, 512, 104),
    ("b16_B4_T257", 16, 4, 257, 512, 105),
    ("b16_B3_T600", 16, 3, 600, 512, 106),
    ("b18_B3_T18", 18, 3, 18, 512, 107),
    ("b18_B4_T257", 18, 4, 257, 512, 108),
    ("b20_B4_T257", 20, 4, 257, 512, 109),
    ("b32_B1_T2", 32, 1, 2, 512, 110),
    ("b32_B4_T257", 32, 4, 257, 512, 111),
    ("b3_B4_T33", 3, 4, 33, 64, 112),
]

# (name, bpt, B, Tr, seed)
RAW_INDEX_CASES = [
    ("raw_b8", 8, 4, 97, 201),
    ("raw_b16", 16, 3, 300, 202),
    ("raw_b5", 5, 2, 64, 203),
]


def cross_weights(seed, D):
    """q_w (D, D), kv_w (2, D, D) ~ U(+-sqrt(3)*0.5/sqrt(D)) as CrossAttention.__init__ draws them
    (train_gpt.py:257-263), c_proj weight as CastedLinear; heads = D // 128."""
    rs = np.random.RandomState(seed)
    bound = (3 ** 0.5) * 0.5 * (D ** -0.5)
    q_w = rs.uniform(-bound, bound, (D, D)).astype(np.float32)
    kv_w = rs.uniform(-bound, bound, (2, D, D)).astype(np.float32)
    return q_w, kv_w, casted_linear_weight(seed + 1, D, D)


def digit_cross_weights(seed, D):
    """c_q, c_k, c_v, c_proj weights (D, D) ~ U(+-1/sqrt(D)), the nn.Linear default that mathblations' CrossAttention
    keeps (model.py:101-105)."""
    rs = np.random.RandomState(seed)
    bound = D ** -0.5
    return tuple(rs.uniform(-bound, bound, (D, D)).astype(np.float32) for _ in range(4))


DIGIT_CROSS_CASES = [
    # name, max_digits_per_token (= length_factor), max_tokens_per_num, D, heads, B, seed
    ("c1", 3, 10, 256, 2, 8, 811),      # config 1: GenerateEquations defaults, 8 x 32 tokens, 1003-token vocabulary
    ("runcfg", 4, 3, 256, 2, 8, 812),   # ablations-mixin.sh:2: 4 digits per token, 8 x 11 tokens, 10003-token vocabulary
]


CROSS_CASES = [
    # name, Vt, D (= token = byte = model dim; heads = D/128), bpt, T, seed
    ("h1", 97, 128, 4, 12, 701),        # one head: the .view() of train_gpt.py:283-284 is the identity permutation
    ("h2", 97, 256, 8, 24, 702),
    ("c2dims", 512, 768, 16, 40, 703),  # 6 heads, bpt 16
]
