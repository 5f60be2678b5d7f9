"""numpy front-end of the CPU oracle (oracle/libmot_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- never by the product package.  Each function
is a thin ctypes call into ``mot_oracle.c``, whose comments cite the reference lines
it restates (all under /root/reference, which is not needed at run time).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libmot_oracle.so"

MODE_NOOP, MODE_SUM, MODE_MEAN, MODE_CONCAT_LINEAR = 0, 1, 2, 3
_MODES = {"noop": MODE_NOOP, "sum": MODE_SUM, "mean": MODE_MEAN, "concat_linear": MODE_CONCAT_LINEAR}


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (a few seconds); no-op when up to date."""
    srcs = [_HERE / "mot_oracle.c", _HERE / "mot_oracle_float.inc"]
    if force or not _LIB_PATH.exists() or any(s.stat().st_mtime > _LIB_PATH.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", str(_HERE), "-B" if force else "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build()
        _lib = C.CDLL(str(_LIB_PATH))
    return _lib


def set_threads(n: int) -> None:
    """OpenMP thread count for the timed cpu_baseline (libgomp reads the env lazily)."""
    os.environ["OMP_NUM_THREADS"] = str(n)
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass


def set_eps(eps: float) -> None:
    """rms-norm eps override (0 restores the per-dtype default); 2**-7 restates the bf16 path."""
    lib().oracle_set_eps(C.c_double(eps))


def set_round_segments_bf16(on: bool) -> None:
    """bf16 F.linear emulation: round the (normalised, scaled) concat operand to bf16 before the contraction and the
    contraction's result to bf16 after it (both are bf16 tensors in the reference's eager bf16 run)."""
    lib().oracle_set_round_segments_bf16(C.c_int(int(on)))


def set_round_kv_bf16(on: bool) -> None:
    """cross_attn (forward): round norm(k) and lambda * v to bf16 where they are formed -- bf16 tensors in the reference's bf16 cast
    (train_gpt.py:278, 280) and what the HIP path keeps with matmul_dtype = MOT_BF16 (one id tensor, at most 16 keys per token)."""
    lib().oracle_set_round_kv_bf16(C.c_int(int(on)))


def set_rotary_f32_cast(on: bool) -> None:
    """Rotary casts each head to float32 before rotating (train_gpt.py:202, the default) or keeps its dtype
    (mathblations/model.py:51-58); a difference in the float64 functions only."""
    lib().oracle_set_rotary_f32_cast(C.c_int(int(on)))


def bf16_round(a) -> np.ndarray:
    """Round float32/float64 values to the nearest bfloat16 (ties to even), returned as float32."""
    x = np.ascontiguousarray(a, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(x.shape)


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _check(rc: int, what: str):
    if rc == -1:
        raise AssertionError(f"{what}: bad shape (reference asserts here)")
    if rc == -3:
        raise IndexError(f"{what}: index out of range")
    if rc != 0:
        raise RuntimeError(f"{what}: oracle error {rc}")


def tokens_to_bytes(tokens: np.ndarray, table_f32: np.ndarray) -> np.ndarray:
    """data_creation.py:61-67.  2-D tokens -> (B, T*bpt); 1-D -> (1, T*bpt)."""
    tok = _c(tokens, np.int32)
    tab = _c(table_f32, np.float32)
    vocab, bpt = tab.shape
    out = np.empty(tok.size * bpt, dtype=np.int64)
    _check(lib().oracle_tokens_to_bytes(_p(tok, C.c_int32), _p(tab, C.c_float), C.c_int64(vocab),
                                        C.c_int64(tok.size), C.c_int(bpt), _p(out, C.c_int64)),
           "tokens_to_bytes")
    return out.reshape(tok.shape[0], -1) if tok.ndim == 2 else out.reshape(1, -1)


def _pull(fn, byte_tensor, bpt, pad, eot):
    x = _c(byte_tensor, np.int64)
    assert x.ndim == 2
    B, T = x.shape
    if T == 0:
        return x
    out = np.empty_like(x)
    _check(fn(_p(x, C.c_int64), C.c_int64(B), C.c_int64(T), C.c_int(bpt), C.c_int64(pad), C.c_int64(eot),
              _p(out, C.c_int64)), "pull")
    return out


def pull_from_left(byte_tensor, bytes_per_token, pad_byte, eot_byte):
    """data_creation.py:179-305."""
    return _pull(lib().oracle_pull_from_left, byte_tensor, bytes_per_token, pad_byte, eot_byte)


def pull_from_right(byte_tensor, bytes_per_token, pad_byte, eot_byte):
    """data_creation.py:71-176."""
    return _pull(lib().oracle_pull_from_right, byte_tensor, bytes_per_token, pad_byte, eot_byte)


def create_batch(tokens, bytes_per_token, pad_byte, eot_byte, table_right_f32, table_left_f32):
    """data_creation.py:308-330 (argument order as the reference: right table, then left)."""
    tok = _c(tokens, np.int32)
    B, T = tok.shape
    tl, tr = _c(table_left_f32, np.float32), _c(table_right_f32, np.float32)
    out = np.empty((B, T, 1 + 4 * bytes_per_token), dtype=np.int64)
    _check(lib().oracle_create_batch(_p(tok, C.c_int32), _p(tl, C.c_float), _p(tr, C.c_float),
                                     C.c_int64(tl.shape[0]), C.c_int64(B), C.c_int64(T), C.c_int(bytes_per_token),
                                     C.c_int64(pad_byte), C.c_int64(eot_byte), _p(out, C.c_int64)), "create_batch")
    return out


def byte_stats(padded, pulled, pad_byte):
    """runs/79_mot-in_toks-valemb.py:484-488 -> (total, pads_before, pads_after)."""
    a, b = _c(padded, np.int64).ravel(), _c(pulled, np.int64).ravel()
    st = np.zeros(3, dtype=np.int64)
    lib().oracle_byte_stats(_p(a, C.c_int64), _p(b, C.c_int64), C.c_int64(a.size), C.c_int64(pad_byte),
                            _p(st, C.c_int64))
    return st


def tokens_to_digits(tokens, length_factor):
    """mathblations/data.py:92-109."""
    tok = _c(tokens, np.int64)
    out = np.empty(tok.size * length_factor, dtype=np.int64)
    _check(lib().oracle_tokens_to_digits(_p(tok, C.c_int64), C.c_int64(tok.size), C.c_int(length_factor),
                                         _p(out, C.c_int64)), "tokens_to_digits")
    return out.reshape(*tok.shape[:-1], -1) if tok.ndim > 1 else out


def rank_slice_shift(data, pos, batch, seq, rank, world):
    """train_gpt.py:795-805 + the [:, :-1] / [:, 1:] shift of 692-764."""
    d = _c(data, np.int32)
    assert batch % world == 0
    rows = batch // world
    assert pos + (rank + 1) * rows * (seq + 1) <= d.size
    ti = np.empty((rows, seq), dtype=np.int32)
    tg = np.empty((rows, seq), dtype=np.int32)
    _check(lib().oracle_rank_slice_shift(_p(d, C.c_int32), C.c_int64(pos), C.c_int64(batch), C.c_int64(seq),
                                         C.c_int(rank), C.c_int(world), _p(ti, C.c_int32), _p(tg, C.c_int32)),
           "rank_slice_shift")
    return ti, tg


def embed_mix(tokens, ids_a, ids_b, tok_table, byte_table, *, mode, bpt, weight=None, bias=None,
              bytes_first=False, norm_tok=False, norm_byte=False, norm_out=False,
              scale_tok=1.0, scale_byte=1.0, dtype=np.float32, return_seam=False):
    """Float path; see mot_oracle_float.inc for the formula and reference lines.

    tokens (...,) int; ids_a/ids_b (..., bpt) or (B, T*bpt) int (ids_b optional);
    tables in ``dtype`` (float32 -> the fp32 oracle, float64 -> the exact one).
    """
    real = C.c_float if dtype == np.float32 else C.c_double
    fn = lib().oracle_embed_mix_f32 if dtype == np.float32 else lib().oracle_embed_mix_f64
    tok = _c(tokens, np.int32)
    n = tok.size
    tt = _c(tok_table, dtype)
    Dt = tt.shape[1]
    m = _MODES[mode]
    if m == MODE_NOOP:
        bt = np.zeros((1, 1), dtype=dtype)
        ia = None
        ib = None
        bpt_ = 0
    else:
        bt = _c(byte_table, dtype)
        ia = _c(ids_a, np.int64).reshape(-1)
        ib = None if ids_b is None else _c(ids_b, np.int64).reshape(-1)
        bpt_ = bpt
        assert ia.size == n * bpt
    Db = bt.shape[1]
    if m == MODE_CONCAT_LINEAR:
        w = _c(weight, dtype)
        Dm = w.shape[0]
        assert w.shape[1] == Dt + bpt_ * Db, (w.shape, Dt, bpt_, Db)
    else:
        w = None
        Dm = Dt
    bs = None if bias is None else _c(bias, dtype)
    out = np.empty((n, Dm), dtype=dtype)
    te = np.empty((n, Dt), dtype=dtype) if return_seam else None
    be = np.empty((n * bpt_, Db), dtype=dtype) if (return_seam and m != MODE_NOOP) else None
    _check(fn(_p(tok, C.c_int32), C.c_int64(n), _p(ia, C.c_int64), _p(ib, C.c_int64), C.c_int(bpt_),
              _p(tt, real), C.c_int64(tt.shape[0]), C.c_int(Dt),
              _p(bt, real), C.c_int64(bt.shape[0]), C.c_int(Db),
              C.c_int(m), C.c_int(int(bytes_first)), _p(w, real), _p(bs, real), C.c_int(Dm),
              C.c_int(int(norm_tok)), C.c_int(int(norm_byte)), C.c_int(int(norm_out)),
              C.c_double(scale_tok), C.c_double(scale_byte),
              _p(out, real), _p(te, real), _p(be, real)), "embed_mix")
    out = out.reshape(*tok.shape, Dm)
    if return_seam:
        return out, te.reshape(*tok.shape, Dt), (None if be is None else be.reshape(*tok.shape[:-1], -1, Db))
    return out


def embed_mix_bwd(tokens, ids_a, ids_b, tok_table, byte_table, grad_out, *, mode, bpt, weight=None, bias=None,
                  bytes_first=False, norm_tok=False, norm_byte=False, norm_out=False,
                  scale_tok=1.0, scale_byte=1.0, dtype=np.float32):
    """Gradients of embed_mix w.r.t. (tok_table, byte_table, weight, bias, [scale_tok, scale_byte]) for
    upstream gradient `grad_out` (tokens.shape + (Dm,)); float64 arrays (see mot_oracle_float.inc)."""
    real = C.c_float if dtype == np.float32 else C.c_double
    fn = lib().oracle_embed_mix_bwd_f32 if dtype == np.float32 else lib().oracle_embed_mix_bwd_f64
    tok = _c(tokens, np.int32)
    n = tok.size
    tt = _c(tok_table, dtype)
    Dt = tt.shape[1]
    m = _MODES[mode]
    if m == MODE_NOOP:
        bt, ia, ib, bpt_ = np.zeros((1, 1), dtype=dtype), None, None, 0
    else:
        bt = _c(byte_table, dtype)
        ia = _c(ids_a, np.int64).reshape(-1)
        ib = None if ids_b is None else _c(ids_b, np.int64).reshape(-1)
        bpt_ = bpt
    Db = bt.shape[1]
    w = _c(weight, dtype) if m == MODE_CONCAT_LINEAR else None
    Dm = w.shape[0] if w is not None else Dt
    bs = None if bias is None else _c(bias, dtype)
    g = _c(grad_out, dtype).reshape(n, Dm)
    d_tok = np.zeros(tt.shape, dtype=np.float64)
    d_byte = np.zeros(bt.shape, dtype=np.float64)
    d_w = np.zeros(w.shape, dtype=np.float64) if w is not None else None
    d_b = np.zeros(Dm, dtype=np.float64) if bs is not None else None
    d_s = np.zeros(2, dtype=np.float64)
    _check(fn(_p(tok, C.c_int32), C.c_int64(n), _p(ia, C.c_int64), _p(ib, C.c_int64), C.c_int(bpt_),
              _p(tt, real), C.c_int64(tt.shape[0]), C.c_int(Dt), _p(bt, real), C.c_int64(bt.shape[0]), C.c_int(Db),
              C.c_int(m), C.c_int(int(bytes_first)), _p(w, real), _p(bs, real), C.c_int(Dm),
              C.c_int(int(norm_tok)), C.c_int(int(norm_byte)), C.c_int(int(norm_out)),
              C.c_double(scale_tok), C.c_double(scale_byte), _p(g, real),
              _p(d_tok, C.c_double), _p(d_byte, C.c_double), _p(d_w, C.c_double), _p(d_b, C.c_double),
              _p(d_s, C.c_double)), "embed_mix_bwd")
    return dict(tok_table=d_tok, byte_table=None if m == MODE_NOOP else d_byte, weight=d_w, bias=d_b, scales=d_s)


def cross_attn(tokens, ids_a, ids_b, tok_table, byte_table, q_w, kv_w, proj_w, lambda_factor,
               cos_q, sin_q, cos_k, sin_k, *, bpt, n_heads, head_dim=128, norm_tok=True, norm_byte=True,
               head_layout=0, dtype=np.float32):
    """ByteMixinCrossAttn on FlexibleEmbedding's outputs, batch 1 (mot_oracle_attn.inc; train_gpt.py:271-300,
    446-464).  tokens (T,), ids (T*bpt,), tables (rows, D), q_w (H*hd, D), kv_w (2, H*hd, D), proj_w (D, H*hd);
    cos/sin: the Rotary buffers, float32 (len, hd/2).  head_layout 0 = the reference's .view()."""
    real = C.c_float if dtype == np.float32 else C.c_double
    fn = lib().oracle_cross_attn_f32 if dtype == np.float32 else lib().oracle_cross_attn_f64
    tok = _c(tokens, np.int32).reshape(-1)
    T = tok.size
    tt, bt = _c(tok_table, dtype), _c(byte_table, dtype)
    D = tt.shape[1]
    assert bt.shape[1] == D
    ia = _c(ids_a, np.int64).reshape(-1)
    ib = None if ids_b is None else _c(ids_b, np.int64).reshape(-1)
    assert ia.size == T * bpt
    qw, kvw, pw = _c(q_w, dtype), _c(kv_w, dtype), _c(proj_w, dtype)
    HD = n_heads * head_dim
    assert qw.shape == (HD, D) and kvw.shape == (2, HD, D) and pw.shape == (D, HD)
    cq, sq, ck, sk = (_c(a, np.float32) for a in (cos_q, sin_q, cos_k, sin_k))
    assert cq.shape[0] >= T and ck.shape[0] >= T * bpt and cq.shape[1] == head_dim // 2
    out = np.empty((T, D), dtype=dtype)
    _check(fn(_p(tok, C.c_int32), C.c_int64(T), _p(ia, C.c_int64), _p(ib, C.c_int64), C.c_int(bpt),
              _p(tt, real), C.c_int64(tt.shape[0]), _p(bt, real), C.c_int64(bt.shape[0]), C.c_int(D),
              C.c_int(int(norm_tok)), C.c_int(int(norm_byte)),
              _p(qw, real), _p(kvw, real), _p(pw, real), C.c_int(n_heads), C.c_int(head_dim), C.c_double(float(lambda_factor)),
              _p(cq, C.c_float), _p(sq, C.c_float), _p(ck, C.c_float), _p(sk, C.c_float),
              C.c_int(head_layout), _p(out, real)), "cross_attn")
    return out


def cross_attn_bwd(tokens, ids_a, ids_b, tok_table, byte_table, q_w, kv_w, proj_w, lambda_factor,
                   cos_q, sin_q, cos_k, sin_k, grad_out, *, bpt, n_heads, head_dim=128, norm_tok=True, norm_byte=True,
                   head_layout=0, dtype=np.float64):
    """Gradients of cross_attn w.r.t. (tok_table, byte_table, q_w, kv_w, proj_w, lambda) for upstream gradient
    grad_out (T, D); float64 arrays (mot_oracle_attn.inc)."""
    real = C.c_float if dtype == np.float32 else C.c_double
    fn = lib().oracle_cross_attn_bwd_f32 if dtype == np.float32 else lib().oracle_cross_attn_bwd_f64
    tok = _c(tokens, np.int32).reshape(-1)
    T = tok.size
    tt, bt = _c(tok_table, dtype), _c(byte_table, dtype)
    D = tt.shape[1]
    ia = _c(ids_a, np.int64).reshape(-1)
    ib = None if ids_b is None else _c(ids_b, np.int64).reshape(-1)
    qw, kvw, pw = _c(q_w, dtype), _c(kv_w, dtype), _c(proj_w, dtype)
    HD = n_heads * head_dim
    cq, sq, ck, sk = (_c(a, np.float32) for a in (cos_q, sin_q, cos_k, sin_k))
    g = _c(grad_out, dtype).reshape(T, D)
    out = dict(tok_table=np.zeros(tt.shape), byte_table=np.zeros(bt.shape), q_w=np.zeros((HD, D)), kv_w=np.zeros((2, HD, D)),
               proj_w=np.zeros((D, HD)), lambda_factor=np.zeros(1))
    _check(fn(_p(tok, C.c_int32), C.c_int64(T), _p(ia, C.c_int64), _p(ib, C.c_int64), C.c_int(bpt),
              _p(tt, real), C.c_int64(tt.shape[0]), _p(bt, real), C.c_int64(bt.shape[0]), C.c_int(D),
              C.c_int(int(norm_tok)), C.c_int(int(norm_byte)),
              _p(qw, real), _p(kvw, real), _p(pw, real), C.c_int(n_heads), C.c_int(head_dim), C.c_double(float(lambda_factor)),
              _p(cq, C.c_float), _p(sq, C.c_float), _p(ck, C.c_float), _p(sk, C.c_float),
              C.c_int(head_layout), _p(g, real),
              _p(out["tok_table"], C.c_double), _p(out["byte_table"], C.c_double), _p(out["q_w"], C.c_double),
              _p(out["kv_w"], C.c_double), _p(out["proj_w"], C.c_double), _p(out["lambda_factor"], C.c_double)), "cross_attn_bwd")
    return out


# ----------------------------------------------------------------------------------------------
# Llama character front-end (config 5).  PARITY UNPINNED: inference/inference.py cannot be imported offline (hub login at
# line 34, from_pretrained at 52); these are line-by-line restatements checked against hand-derived vectors only.
# ----------------------------------------------------------------------------------------------
def chr_tokenize(x: str, leading_space_ind: int = 288, bos_token_id: int = 128000, eos_token_id: int = 128001) -> int:
    """inference/inference.py:56-67."""
    ind = ord(x)
    if ind <= 127:                      # ascii
        return ind
    if ind == leading_space_ind:        # leading character for token
        return 128
    if ind == bos_token_id:             # BOS
        return 129
    if ind == eos_token_id:             # EOS/PAD
        return 130
    return 131                          # unicode past 128


def create_char_matrix(char_tokens, seq_len: int, max_char: int = 8) -> np.ndarray:
    """inference/inference.py:79-96 (+ the .long() of line 102): rows of 2, the row's characters (truncated at max_char),
    one 130 after them when the row is not full."""
    mat = np.zeros((seq_len, max_char), dtype=np.float32) + 2
    for row, sublist in enumerate(char_tokens):
        if row >= seq_len:
            break
        ind = 0
        for char in sublist:
            if ind >= max_char:
                break
            mat[row][ind] = char
            ind += 1
        if ind < max_char:
            mat[row][ind] = 130
    return mat.astype(np.int64)


def _rotary_rotate(t: np.ndarray, rot_dim: int, theta: float = 10000.0) -> np.ndarray:
    """rotary_embedding_torch.RotaryEmbedding(dim=rot_dim).rotate_queries_or_keys(t) with seq_dim = -2 -- a third-party
    dependency absent from /root/reference (imported at inference.py:19, version not pinned by the reference); restated
    from its published algorithm: freqs_f = theta ** -(2f / rot_dim), f < rot_dim / 2; angle(pos, 2f) = angle(pos, 2f+1) =
    pos * freqs_f; the first rot_dim elements are rotated in ADJACENT pairs (x1, x2) -> (x1 cos - x2 sin, x2 cos + x1 sin),
    the rest pass through."""
    seq = t.shape[-2]
    freqs = 1.0 / (theta ** (np.arange(0, rot_dim, 2)[: rot_dim // 2].astype(np.float64) / rot_dim))
    ang = np.repeat(np.arange(seq, dtype=np.float64)[:, None] * freqs[None, :], 2, axis=-1)      # (seq, rot_dim): n -> (n r), r = 2
    x = t[..., :rot_dim]
    xr = x.reshape(*x.shape[:-1], rot_dim // 2, 2)
    half = np.stack((-xr[..., 1], xr[..., 0]), axis=-1).reshape(x.shape)                          # rotate_half
    out = t.copy()
    out[..., :rot_dim] = x * np.cos(ang) + half * np.sin(ang)
    return out


def char_swa(tokens, char_ids, tok_table, char_table, attn_norm_w, char_norm_w, wq, wk, wv, wo, *, n_heads, head_dim, window=8,
             norm_eps=1e-5, version="two_residual", lambda_tok=1.0, lambda_char=1.0, round_token_products_bf16=False) -> np.ndarray:
    """float64 restatement, line by line, of CustomLlamaModel.forward's gathers (inference.py:323-327) ->
    TokenMixByCharBMMBlock.forward up to `h` (260-267) -> TokenMixByCharBMM.forward (189-238, swa_transform 174-179).
    tokens (B, T), char_ids (B, T, c_v).  PARITY UNPINNED (see the section header).  numpy, small sizes only.
    round_token_products_bf16: the row operands of the two products over the tokens (the normalised token rows in front of wq,
    the attention output in front of wo), the projected queries, keys and values and the output of wo (bf16 tensors out of wq / wk /
    wv / wo in the reference's bf16 cast) are rounded to bf16 -- what the HIP path does with matmul_dtype = MOT_BF16."""
    f = np.float64
    tok_table, char_table = np.asarray(tok_table, f), np.asarray(char_table, f)
    toks = tok_table[np.asarray(tokens)]                                   # (b, t, d)        line 323
    chars = char_table[np.asarray(char_ids)]                               # (b, t, c_v, d)   line 327
    rms = lambda x, w: x / np.sqrt((x ** 2).mean(-1, keepdims=True) + norm_eps) * np.asarray(w, f)   # RMSNorm, 126-132
    x, cn = rms(toks, attn_norm_w), rms(chars, char_norm_w)
    b, t, c_v, _ = chars.shape
    if round_token_products_bf16:
        x = np.asarray(bf16_round(x), f)
    xq = x @ np.asarray(wq, f).T                                            # (b, t, bmm)      199
    if round_token_products_bf16:
        xq = np.asarray(bf16_round(xq), f)
    xk, xv = cn @ np.asarray(wk, f).T, cn @ np.asarray(wv, f).T             # (b, t, c_v, bmm) 200
    if round_token_products_bf16:
        xk, xv = np.asarray(bf16_round(xk), f), np.asarray(bf16_round(xv), f)

    def swa(a):                                                             # 174-179
        pad = np.zeros((b, window - 1, c_v, a.shape[-1]), f)
        a = np.concatenate([pad, a], axis=1).reshape(b, t + window - 1, -1)
        unf = np.stack([a[:, i:i + t] for i in range(window)], axis=-1)     # unfold(1, window, 1): (b, t, c_v * bmm, window)
        return unf.transpose(0, 1, 3, 2).reshape(b, t, c_v * window, -1)

    xk, xv = swa(xk), swa(xv)                                               # (b, t, c_v * window, bmm) 201
    nk = c_v * window
    xq = xq.reshape(b, t, n_heads, head_dim).transpose(0, 2, 1, 3)          # (b, h, t, dh)   204, 211
    xk = xk.reshape(b, t, nk, n_heads, head_dim).transpose(0, 3, 2, 1, 4)   # (b, h, nk, t, dh) 205, 213
    xv = xv.reshape(b, t, nk, n_heads, head_dim)
    xq = _rotary_rotate(xq, head_dim // 2)                                  # 215  (RotaryEmbedding(dim = head_dim // 2), 311)
    xk = _rotary_rotate(xk, head_dim // 2)                                  # 216: the sequence axis (-2) is t here too
    xk = xk.transpose(0, 1, 3, 2, 4)                                        # (b, h, t, nk, dh) 218
    qk = np.einsum("bhtd,bhtkd->bhtk", xq, xk) / head_dim ** 0.5            # 221-222
    qk = np.exp(qk - qk.max(-1, keepdims=True))
    qk = qk / qk.sum(-1, keepdims=True)                                     # 223
    xv = xv.transpose(0, 3, 1, 2, 4)                                        # (b, h, t, nk, dh) 228
    y = np.einsum("bhtk,bhtkd->bhtd", qk, xv)                               # 231-232
    y = y.transpose(0, 2, 1, 3).reshape(b, t, n_heads * head_dim)           # 233
    if round_token_products_bf16:
        y = np.asarray(bf16_round(y), f)
    h = y @ np.asarray(wo, f).T                                             # 235
    if round_token_products_bf16:
        h = np.asarray(bf16_round(h), f)                                    # (self.wo(...) is a bf16 tensor before the residuals are added)
    if version == "one_residual":
        h = h + toks                                                        # 264
    elif version == "two_residual":
        h = h + lambda_tok * toks + lambda_char * chars.mean(axis=-2)       # 267
    return h
