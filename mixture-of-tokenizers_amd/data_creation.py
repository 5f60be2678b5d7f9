"""Drop-in for the byte-index producer of the reference loader: same function names, argument
order and return shapes/dtypes as scaled-pre-train/data_creation.py:43-330, so that
``from data_creation import make_embedding, tokens_to_bytes, pull_from_left, pull_from_right``
(train_gpt.py:31, data_loader.py:17) can point here unchanged.  All work runs in HIP kernels
on the tensors' device; there is no host synchronisation (the reference's per-row
``nonzero``/``searchsorted`` loop syncs at least twice per batch row).
"""
from __future__ import annotations

import json

import torch
from torch import nn

from . import functional as F_mot


def load_ttb(filename: str) -> dict[int, list[int]]:
    """data_creation.py:43-48: ``embeddings/<filename>`` JSON {token id: [byte ids]}."""
    with open(f"embeddings/{filename}", "r") as f:
        ttb = json.loads(f.read())
    return {int(k): [int(x) for x in v] for k, v in ttb.items()}


class TokenToBytesTable(nn.Embedding):
    """What make_embedding returns: an ``nn.Embedding`` (so ``.to(device)``, ``isinstance`` checks
    and ``.weight`` keep working, train_gpt.py:666-672) whose fp32 weight holds integers.  The
    kernels read an int16/int32 copy of it, rebuilt when the weight moves or changes."""

    def __init__(self, num_embeddings: int, embedding_dim: int):
        super().__init__(num_embeddings, embedding_dim)  # same RNG draw as the reference (quirk: rows
        self.weight.requires_grad = False                # absent from the JSON keep this random init)
        self._int_cache: tuple | None = None

    def int_table(self) -> torch.Tensor:
        w = self.weight
        key = (w.data_ptr(), w._version, w.device, w.dtype)
        if self._int_cache is None or self._int_cache[0] != key:
            self._int_cache = (key, int_table_of(w))
        return self._int_cache[1]


def int_table_of(weight: torch.Tensor) -> torch.Tensor:
    """fp32 table -> integer table with the reference's cast (``.to(torch.int64)`` = truncation,
    data_creation.py:63); int16 when every id fits (458 byte ids do), else int32."""
    t = weight.detach().to(torch.int64)
    if t.numel() and int(t.abs().max()) >= 2 ** 15:
        return t.to(torch.int32).contiguous()
    return t.to(torch.int16).contiguous()


def make_embedding(filename: str, vocab_size: int) -> nn.Embedding:
    """data_creation.py:51-58 (``dim`` is parsed from ``ttb_<dim>_...json``)."""
    dim = int(filename.split("_")[1])
    emb = TokenToBytesTable(vocab_size, dim)
    ttb = load_ttb(filename)
    idx = torch.tensor(sorted(ttb), dtype=torch.long)
    rows = torch.tensor([ttb[int(i)] for i in idx], dtype=emb.weight.dtype)
    with torch.no_grad():
        emb.weight.data[idx] = rows
    emb.weight.requires_grad = False
    return emb


def _table_of(emb) -> torch.Tensor:
    if isinstance(emb, TokenToBytesTable):
        return emb.int_table()
    if isinstance(emb, nn.Embedding):  # e.g. built by the reference's own make_embedding
        # the integer copy lives ON the module (it dies with it; a dict keyed by id(emb) could serve a later module
        # that reuses the id), and is rebuilt when the weight storage, version, device or dtype changes.  Writes
        # through ``weight.data[...] = `` do not bump the version: call ``del emb._mot_int_cache`` after those.
        w = emb.weight
        key = (w.data_ptr(), w._version, w.device, w.dtype)
        hit = emb.__dict__.get("_mot_int_cache")
        if hit is None or hit[0] != key:
            hit = (key, int_table_of(w))
            emb.__dict__["_mot_int_cache"] = hit
        return hit[1]
    if isinstance(emb, torch.Tensor):
        return emb if emb.dtype in (torch.int16, torch.int32) else int_table_of(emb)
    raise TypeError(f"unsupported token->byte table: {type(emb)}")


def tokens_to_bytes(tokens: torch.Tensor, emb) -> torch.Tensor:
    """data_creation.py:61-67: (B, T) -> (B, T*bpt) int64; 1-D (T,) -> (1, T*bpt)."""
    out = F_mot.tokens_to_bytes(tokens, _table_of(emb))
    if tokens.ndim == 2:
        return out.view(out.shape[0], -1)
    return out.view(-1).unsqueeze(0)


def pull_from_right(byte_tensor: torch.Tensor, bytes_per_token: int, pad_byte: int, eot_byte: int) -> torch.Tensor:
    """data_creation.py:71-176."""
    return F_mot.pull_bytes(byte_tensor, bytes_per_token, pad_byte, eot_byte, "right")


def pull_from_left(byte_tensor: torch.Tensor, bytes_per_token: int, pad_byte: int, eot_byte: int) -> torch.Tensor:
    """data_creation.py:179-305."""
    return F_mot.pull_bytes(byte_tensor, bytes_per_token, pad_byte, eot_byte, "left")


def create_batch(tokens: torch.Tensor, bytes_per_token: int, pad_byte: int, eot_byte: int,
                 tokens_to_bytes_right_pad, tokens_to_bytes_left_pad) -> torch.Tensor:
    """data_creation.py:308-330: (B, T, 1 + 4*bpt), dtype int64 (torch.cat promotes the int32 tokens)."""
    tl, tr = _table_of(tokens_to_bytes_left_pad), _table_of(tokens_to_bytes_right_pad)
    assert tl.shape[1] == bytes_per_token
    return F_mot.create_batch(tokens, tl, tr, pad_byte, eot_byte)


# ------------------------------------------------------------------------------------------------
# mathblations: token -> digit ids (mathblations/data.py:92-109) is the same gather with an arithmetic table
# ------------------------------------------------------------------------------------------------
def make_digit_table(max_digits_per_token: int = 3) -> torch.Tensor:
    """GenerateEquations.tokens_to_digits as a (vocab_size, max_digits_per_token) int16 table: a numeric token's
    decimal digits right-aligned, filled with 13; the operator token -> 10, '=' -> 11, the pad token -> 12 in the
    last slot (data.py:58-60, 96-107; vocab_size = 10**d + 3, data.py:72)."""
    d = int(max_digits_per_token)
    assert d > 0, f"max_digits_per_token must be > 0 (got {d})"          # data.py:41
    n_num = 10 ** d
    tab = torch.full((n_num + 3, d), 13, dtype=torch.int16)
    v = torch.arange(n_num, dtype=torch.int64)
    for i in range(d):                                                      # slot -1-i holds digit i of str(v), if v has it
        has = v >= 10 ** i if i else torch.ones_like(v, dtype=torch.bool)
        tab[:n_num, d - 1 - i] = torch.where(has, (v // 10 ** i) % 10, torch.tensor(13)).to(torch.int16)
    tab[n_num, -1], tab[n_num + 1, -1], tab[n_num + 2, -1] = 10, 11, 12    # op, eq, pad
    return tab


def tokens_to_digits(tokens: torch.Tensor, digit_table: torch.Tensor) -> torch.Tensor:
    """data.py:92-109 on the device: (T,) -> (T*d,) int64, (B, T) -> (B, T*d); `digit_table` from make_digit_table
    (moved to the tokens' device by the caller, like the token->byte table)."""
    out = F_mot.tokens_to_bytes(tokens.to(torch.int32), digit_table)
    return out.view(-1) if tokens.ndim == 1 else out.view(tokens.shape[0], -1)


# ------------------------------------------------------------------------------------------------
# Llama character front-end (inference/inference.py): the producer of config 5's character ids
# ------------------------------------------------------------------------------------------------
class CharTokenizer:
    """``chr_tokenize`` / ``create_char_matrix`` of TokenMixByCharStreamingDataset (inference.py:46-96) without the HF
    tokenizer object: its three tokenizer-dependent constants are arguments (Llama-3 defaults: the byte-level BPE's
    leading-space marker "Ġ" = 288, bos_token_id 128000, eos_token_id 128001).  The BPE tokenisation itself
    (``get_tokens``, lines 69-77: AutoTokenizer) stays with the caller -- it needs the tokenizer files.

    Character ids: 0-127 ASCII, 128 leading space, 129 BOS, 130 EOS / end-of-word, 131 other; 2 = fill (line 82)."""

    def __init__(self, num_char_positions: int = 8, leading_space_ind: int = 288, bos_token_id: int = 128000,
                 eos_token_id: int = 128001):
        self.max_char = int(num_char_positions)
        self.leading_space_ind, self.bos_token_id, self.eos_token_id = int(leading_space_ind), int(bos_token_id), int(eos_token_id)

    def chr_tokenize(self, x: str) -> int:          # inference.py:56-67 (host, one character; the batch path maps on the device)
        ind = ord(x)
        if ind <= 127:
            return ind
        if ind == self.leading_space_ind:
            return 128
        if ind == self.bos_token_id:
            return 129
        if ind == self.eos_token_id:
            return 130
        return 131

    def _launch(self, codes, tok_off, seq_off, n_seqs, seq_len, device) -> torch.Tensor:
        dev = torch.device(device)
        c = torch.tensor(codes if len(codes) else [0], dtype=torch.int32).to(dev)
        to, so = torch.tensor(tok_off, dtype=torch.int64).to(dev), torch.tensor(seq_off, dtype=torch.int64).to(dev)
        return F_mot.char_matrix(c, to, so, n_seqs, seq_len, self.max_char, self.leading_space_ind, self.bos_token_id, self.eos_token_id)

    def create_char_matrix(self, char_tokens, seq_len: int, device="cuda") -> torch.Tensor:
        """inference.py:79-96: ``char_tokens`` = one list of character IDS per token (what get_tokens returns) ->
        (seq_len, max_char) int64, as the reference's ``create_char_matrix(...).long()``."""
        codes, off = [], [0]
        for sub in char_tokens:
            codes.extend(-int(v) - 1 for v in sub)              # literal ids
            off.append(len(codes))
        return self._launch(codes, off, [0, len(char_tokens)], 1, seq_len, device)[0]

    def char_matrix_from_tokens(self, token_strings, seq_len: int, bos: bool = True, device="cuda") -> torch.Tensor:
        """get_tokens' character side (lines 72-76) + create_char_matrix for a BATCH of sequences in one launch:
        ``token_strings`` = per sequence, the tokenizer's token strings; ``bos`` prepends the [129] row of line 73.
        Code points travel to the device as they are; chr_tokenize runs there."""
        codes, toff, soff = [], [0], [0]
        for seq in token_strings:
            if bos:
                codes.append(-130)                                # literal id 129
                toff.append(len(codes))
            for tok in seq:
                codes.extend(ord(ch) for ch in tok)
                toff.append(len(codes))
            soff.append(len(toff) - 1)
        return self._launch(codes, toff, soff, len(token_strings), seq_len, device)
