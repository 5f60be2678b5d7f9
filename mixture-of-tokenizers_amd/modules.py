"""``nn.Module`` boundary of the embedding front-end: the reference's own classes for this path,
same constructor arguments, attribute names, state-dict keys and ``forward`` signatures, with the
work done by the fused HIP kernels.

scaled-pre-train (train_gpt.py):  ByteHyperparameters, ModelDims (146-169), norm (172-173),
    CastedLinear (175-186), FlexibleEmbedding (327-379), ByteMixinNoop / ByteMixinConcat /
    ByteMixin (421-443, 467-480); call site ``xt, xb = self.embed(...); x = self.byte_mixin(xt, xb)``
    (605-606) works unchanged.
mathblations (model.py):  GPTConfig (16-29), DigitMixinConcat / DigitMixinNoOp / make_digit_mixin
    (256-284), and the ``wte`` / ``dte`` / ``digit_mixin`` triple of GPT (304-306, 323-327).
modded-nanogpt (runs/71*.py):  the SUM mixin ``norm(embed_tokens(tok) + concat_k embed_bytes(b_k))``
    (227-230, 312-314) and its per-embedding-norm / lambda variants (runs/71041, 71081).

Fusion across the two-module seam: ``FlexibleEmbedding.forward`` (and ``LazyEmbedding.forward``)
return an :class:`EmbedHandle` -- ids plus references to the live Parameters -- instead of
materialised (B,T,D) tensors; the mixin's ``forward`` launches ONE kernel that gathers, normalises,
mixes and writes x.  Anything that needs the tensors themselves calls ``handle.materialize()``.
Parameters are read from their live storage at launch time, so weight tying
(``wte.weight = lm_head.weight``, model.py:316-317) and in-place optimizer updates are honoured.

Autograd: the fused paths are differentiable with one backward call that produces dense fp32 gradient sums for the tables /
weight / bias / scalars, rounded once to the parameter dtype:
  * sum, tokens-only (noop), concat + linear incl. norm(emb(padded) + emb(pulled)): fp32 or bf16 parameters;
  * the MEAN residual (config 5): fp32 parameters, no output norm;
  * the cross-attention mixin over one id tensor OR two (norm(emb(padded) + emb(pulled))): fp32 arithmetic, bf16 tables
    accepted (widened once).
What has no backward raises at FORWARD time instead of silently dropping the graph: MEAN with bf16 tables or an output norm,
the character mixer (inference-only in the reference), and the materialised (non-fused) seam tensors.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Literal

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import functional as F_mot
from .data_creation import _table_of


# ------------------------------------------------------------------------------------------------
# configuration dataclasses (field-for-field the reference's)
# ------------------------------------------------------------------------------------------------
@dataclass
class ByteHyperparameters:  # train_gpt.py:146-161
    bytes_per_token: int = 16
    vocab_size: int = 458
    byte_mixin_method: Literal["cross_attn", "concat", "noop"] = "noop"
    byte_mixout_method: Literal["noop", "copy", "split"] = "noop"
    use_byte_self_attn: bool = False
    padding_in: Literal["left", "right"] = "left"
    padding_out: Literal["left", "right"] = "left"
    pull_in: bool = True
    pull_out: bool = True
    add_padded_and_pulled: bool = False
    sliding_window_tokens: int = 8
    n_layer_out: int = 1
    mix_bytes_within_tok_in: bool = False
    mix_bytes_within_tok_out: bool = False


@dataclass
class ModelDims:  # train_gpt.py:164-169
    model_dim: int = 768
    byte_dim: int = 768
    token_dim: int = 768
    expansion_factor: float = 4.0


@dataclass
class GPTConfig:  # mathblations/model.py:16-29
    vocab_size: int = 50304
    n_layer: int = 12
    n_head: int = 6
    n_embd_tok: int = 768
    n_embd_digit: int = 768
    T: int = 1024
    length_factor: int = 3
    k_gt_q: bool = True
    n_layer_output: int = 1
    digit_mixout_method: Literal["self_attn", "cross_attn", "noop"] = "noop"
    digit_mixin_method: Literal["cross_attn", "concat", "noop"] = "noop"
    use_digit_self_attn: bool = False


def norm(x: Tensor) -> Tensor:
    """train_gpt.py:172-173 (a plain torch op: the trunk calls it on arbitrary activations)."""
    return F.rms_norm(x, (x.size(-1),))


class CastedLinear(nn.Linear):  # train_gpt.py:175-186
    def __init__(self, in_features: int, out_features: int):
        super().__init__(in_features, out_features, bias=False)

    def reset_parameters(self) -> None:
        std = 0.5 * (self.in_features ** -0.5)
        bound = (3 ** 0.5) * std
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)

    def forward(self, x: Tensor):
        return F.linear(x, self.weight.type_as(x))


# ------------------------------------------------------------------------------------------------
# the handle that crosses the embed / mixin seam
# ------------------------------------------------------------------------------------------------
def _check_forward_only(*params) -> None:
    """The materialised seam tensors (fused=False) come from a plain gather launch that records no autograd node."""
    if torch.is_grad_enabled() and any(p is not None and p.requires_grad for p in params):
        raise RuntimeError(
            "mixture-of-tokenizers_amd: the materialised (non-fused) seam tensors have no backward -- gradients flow through "
            "the fused path (FlexibleEmbedding(fused=True) + ByteMixin); call this under torch.no_grad() or with frozen parameters")


def _f32(p: Tensor, what: str) -> Tensor:
    """fp32 (parity mode) and bf16 (the production cast, train_gpt.py:1124-1126) tables are built."""
    if p.dtype not in (torch.float32, torch.bfloat16):
        raise NotImplementedError(f"{what} is {p.dtype}: only float32 and bfloat16 are built")
    return p


class EmbedHandle:
    """What a fused embedding returns in place of its (B, T, D) output: the ids and the table."""

    def __init__(self, *, tokens: Tensor | None = None, tok_weight: Tensor | None = None, norm_tok: bool = False,
                 ids_a: Tensor | None = None, ids_b: Tensor | None = None, byte_weight: Tensor | None = None,
                 norm_byte: bool = False, bpt: int = 0, scale_tok: Tensor | None = None, scale_byte: Tensor | None = None):
        self.tokens, self.tok_weight, self.norm_tok = tokens, tok_weight, norm_tok
        self.ids_a, self.ids_b, self.byte_weight, self.norm_byte, self.bpt = ids_a, ids_b, byte_weight, norm_byte, bpt
        self.scale_tok, self.scale_byte = scale_tok, scale_byte

    def merged(self, other: "EmbedHandle") -> "EmbedHandle":
        """token-side handle + byte-side handle (mathblations: wte(idx), dte(digits))."""
        return EmbedHandle(tokens=self.tokens, tok_weight=self.tok_weight, norm_tok=self.norm_tok,
                           ids_a=other.ids_a, ids_b=other.ids_b, byte_weight=other.byte_weight,
                           norm_byte=other.norm_byte, bpt=other.bpt, scale_tok=self.scale_tok, scale_byte=other.scale_byte)

    def materialize(self) -> tuple[Tensor | None, Tensor | None]:
        """(tok_embs (B,T,Dt) | None, byte_embs (B,T*bpt,Db) | None) exactly as the reference's
        FlexibleEmbedding.forward returns them (train_gpt.py:342-379)."""
        te = be = None
        if self.tokens is not None:
            te = F_mot.gather_rows(_f32(self.tok_weight, "token table"), self.tokens, rms_norm=self.norm_tok, scale=self.scale_tok)
        if self.ids_a is not None:
            be = F_mot.gather_rows(_f32(self.byte_weight, "byte table"), self.ids_a, self.ids_b, rms_norm=self.norm_byte,
                                   scale=self.scale_byte)
        return te, be


# ------------------------------------------------------------------------------------------------
# scaled-pre-train
# ------------------------------------------------------------------------------------------------
class FlexibleEmbedding(nn.Module):
    """train_gpt.py:327-379.  ``fused=True`` (default) defers the gathers to the mixin's kernel."""

    def __init__(self, dims: ModelDims, vocab_size, byte_params: ByteHyperparameters, fused: bool = True):
        super().__init__()
        noop = byte_params.byte_mixin_method == "noop"
        self.embed_tokens = nn.Embedding(vocab_size, dims.token_dim if not noop else dims.model_dim)
        self.embed_bytes = nn.Embedding(byte_params.vocab_size, dims.byte_dim) if not noop else nn.Identity()
        self.byte_params = byte_params
        self.fused = fused
        if noop:
            self._mode = "tokens"
        elif not byte_params.pull_in:
            self._mode = "padded"
        elif not byte_params.add_padded_and_pulled:
            self._mode = "pulled"
        else:
            self._mode = "padded_and_pulled"

    def handle(self, tokens: Tensor, byte_tensor: Tensor | None, byte_tensor_pulled: Tensor | None) -> EmbedHandle:
        bp = self.byte_params
        if self._mode == "tokens":
            return EmbedHandle(tokens=tokens, tok_weight=self.embed_tokens.weight, norm_tok=True)
        ids_a, ids_b = {"padded": (byte_tensor, None), "pulled": (byte_tensor_pulled, None),
                        "padded_and_pulled": (byte_tensor, byte_tensor_pulled)}[self._mode]
        return EmbedHandle(tokens=tokens, tok_weight=self.embed_tokens.weight, norm_tok=True, ids_a=ids_a, ids_b=ids_b,
                           byte_weight=self.embed_bytes.weight, norm_byte=True, bpt=bp.bytes_per_token)

    def forward(self, tokens: Tensor, byte_tensor: Tensor | None, byte_tensor_pulled: Tensor | None):
        h = self.handle(tokens, byte_tensor, byte_tensor_pulled)
        if self.fused:
            return h, None
        _check_forward_only(self.embed_tokens.weight, getattr(self.embed_bytes, "weight", None))
        return h.materialize()


class ByteMixinNoop(nn.Module):  # train_gpt.py:421-427
    def __init__(self, dims: ModelDims, max_seq_len: int, byte_params: ByteHyperparameters):
        super().__init__()
        self.attention = self.mixin = nn.Identity()

    def forward(self, x, *args):
        if isinstance(x, EmbedHandle):
            return F_mot.embed_mix(x.tokens, _f32(x.tok_weight, "token table"), mode="noop", norm_tok=x.norm_tok,
                                   scale_tok=x.scale_tok)
        return x


def _mix_concat(h_or_tok, byte_embs, *, bpt: int, weight: Tensor, bias: Tensor | None, bytes_first: bool, norm_out: bool):
    """Shared by ByteMixinConcat and DigitMixinConcat: fused when given a handle, otherwise the same
    kernel over the already materialised seam tensors (rows addressed by arange ids)."""
    if isinstance(h_or_tok, EmbedHandle):
        h = h_or_tok
        if h.tok_weight.dtype != weight.dtype:   # CastedLinear.forward: self.weight.type_as(x)  (train_gpt.py:185-186)
            weight = weight.to(h.tok_weight.dtype)
            bias = None if bias is None else bias.to(h.tok_weight.dtype)
        return F_mot.embed_mix(h.tokens, _f32(h.tok_weight, "token table"), _f32(h.byte_weight, "byte table"),
                               mode="concat_linear", bpt=h.bpt, ids_a=h.ids_a.reshape(h.tokens.shape[0], -1),
                               ids_b=None if h.ids_b is None else h.ids_b.reshape(h.tokens.shape[0], -1),
                               weight=_f32(weight, "mixin weight"), bias=bias, bytes_first=bytes_first,
                               norm_tok=h.norm_tok, norm_byte=h.norm_byte, norm_out=norm_out,
                               scale_tok=h.scale_tok, scale_byte=h.scale_byte)
    tok_embs = h_or_tok
    _check_forward_only(weight, bias)
    if tok_embs.requires_grad or byte_embs.requires_grad:
        _check_forward_only(tok_embs, byte_embs)
    if tok_embs.dtype != weight.dtype:       # CastedLinear.forward: self.weight.type_as(x)  (train_gpt.py:185-186), as on the fused path
        weight = weight.to(tok_embs.dtype)
        bias = None if bias is None else bias.to(tok_embs.dtype)
    B, T, Dt = tok_embs.shape
    Db = byte_embs.shape[-1]
    dev = tok_embs.device
    tok_rows = torch.arange(B * T, dtype=torch.int32, device=dev).view(B, T)
    byte_rows = torch.arange(B * T * bpt, dtype=torch.int64, device=dev).view(B, T * bpt)
    return F_mot.embed_mix(tok_rows, tok_embs.reshape(B * T, Dt), byte_embs.reshape(B * T * bpt, Db), mode="concat_linear",
                           bpt=bpt, ids_a=byte_rows, weight=_f32(weight, "mixin weight"), bias=bias, bytes_first=bytes_first,
                           norm_out=norm_out)


class ByteMixinConcat(nn.Module):  # train_gpt.py:430-443
    def __init__(self, dims: ModelDims, max_seq_len: int, byte_params: ByteHyperparameters):
        super().__init__()
        self.byte_params = byte_params
        if byte_params.use_byte_self_attn:
            raise NotImplementedError("use_byte_self_attn (ByteSelfAttn, train_gpt.py:382-418) is outside the front-end path")
        self.attention = nn.Identity()
        self.mixin = CastedLinear(dims.token_dim + dims.byte_dim * byte_params.bytes_per_token, dims.model_dim)

    def forward(self, tok_embs, byte_embs=None) -> Tensor:
        return _mix_concat(tok_embs, byte_embs, bpt=self.byte_params.bytes_per_token, weight=self.mixin.weight, bias=None,
                           bytes_first=False, norm_out=True)


class Rotary(nn.Module):  # train_gpt.py:188-207: the buffers, built with the same torch expressions
    def __init__(self, dim: int, max_seq_len: int):
        super().__init__()
        angular_freq = (1 / 1024) ** torch.linspace(0, 1, steps=dim // 4, dtype=torch.float32)
        angular_freq = torch.cat([angular_freq, angular_freq.new_zeros(dim // 4)])
        t = torch.arange(max_seq_len, dtype=torch.float32)
        theta = torch.einsum("i,j -> ij", t, angular_freq)
        self.cos = nn.Buffer(theta.cos(), persistent=False)
        self.sin = nn.Buffer(theta.sin(), persistent=False)


class CrossAttention(nn.Module):
    """train_gpt.py:243-300: every token attends to its own ``chars_per_token`` byte embeddings.  Same parameters
    (q_w, kv_w, lambda_factor, c_proj.weight) and non-persistent Rotary buffers; forward takes the embedding
    handle so the gathers, projections, per-head norm, RoPE and the softmax run in libmot_hip.so."""

    def __init__(self, dim: int, num_heads: int, max_seq_len_q: int, max_seq_len_kv: int, head_dim=128,
                 head_layout: Literal["as_viewed", "per_token"] = "as_viewed"):
        super().__init__()
        if head_dim != 128:
            raise NotImplementedError("CrossAttention: head_dim 128 only (every instance the reference builds, train_gpt.py:459)")
        self.num_heads, self.head_dim, self.head_layout = num_heads, head_dim, head_layout
        hdim = num_heads * head_dim
        bound = (3 ** 0.5) * 0.5 * (dim ** -0.5)
        self.q_w = nn.Parameter(torch.empty(hdim, dim).uniform_(-bound, bound))
        self.kv_w = nn.Parameter(torch.empty(2, hdim, dim).uniform_(-bound, bound))
        self.lambda_factor = nn.Parameter(torch.tensor(0.5))
        self.rotary_q = Rotary(head_dim, max_seq_len_q)
        self.rotary_k = Rotary(head_dim, max_seq_len_kv)
        self.c_proj = CastedLinear(hdim, dim)
        self.attn_scale = 0.12   # kept for state parity; the reference's forward divides by sqrt(head_dim) instead (line 286)
        self._kv_cache: dict = {}   # per-byte-row K/V tables, reused by no-grad calls while the parameters are unchanged

    def forward(self, xq, xkv=None) -> Tensor:
        if not isinstance(xq, EmbedHandle):
            raise NotImplementedError("CrossAttention: pass the EmbedHandle of a fused FlexibleEmbedding (materialised inputs are not built)")
        h = xq
        if h.scale_tok is not None or h.scale_byte is not None:
            raise NotImplementedError("CrossAttention: learned embedding scalars are not part of this mixin")
        return F_mot.cross_attn(h.tokens, h.ids_a, _f32(h.tok_weight, "token table"), _f32(h.byte_weight, "byte table"),
                                ids_b=h.ids_b, q_w=self.q_w, kv_w=self.kv_w, proj_w=self.c_proj.weight, lambda_factor=self.lambda_factor,
                                cos_q=self.rotary_q.cos, sin_q=self.rotary_q.sin, cos_k=self.rotary_k.cos, sin_k=self.rotary_k.sin,
                                bpt=h.bpt, n_heads=self.num_heads, norm_tok=h.norm_tok, norm_byte=h.norm_byte, head_layout=self.head_layout,
                                kv_cache=self._kv_cache)


class ByteMixinCrossAttn(nn.Module):  # train_gpt.py:446-464
    def __init__(self, dims: ModelDims, max_seq_len: int, byte_params: ByteHyperparameters):
        super().__init__()
        assert dims.byte_dim == dims.token_dim == dims.model_dim
        self.byte_params = byte_params
        if byte_params.use_byte_self_attn:
            raise NotImplementedError("use_byte_self_attn (ByteSelfAttn, train_gpt.py:382-418) is outside the front-end path")
        self.attention = nn.Identity()
        self.mixin = CrossAttention(dim=dims.model_dim, num_heads=dims.model_dim // 128,
                                    max_seq_len_kv=max_seq_len * byte_params.bytes_per_token, max_seq_len_q=max_seq_len, head_dim=128)

    def forward(self, token_embs, byte_embs=None) -> Tensor:
        return self.mixin(xq=token_embs, xkv=byte_embs)


class ByteMixin(nn.Module):  # train_gpt.py:467-480
    def __init__(self, dims: ModelDims, max_seq_len: int, byte_params: ByteHyperparameters):
        super().__init__()
        if byte_params.byte_mixin_method == "noop":
            self.mixin = ByteMixinNoop(dims, max_seq_len, byte_params)
        elif byte_params.byte_mixin_method == "cross_attn":
            self.mixin = ByteMixinCrossAttn(dims, max_seq_len, byte_params)
        elif byte_params.byte_mixin_method == "concat":
            self.mixin = ByteMixinConcat(dims, max_seq_len, byte_params)
        else:
            raise RuntimeError(f"Invalid byte mixin method: {byte_params.byte_mixin_method}")

    def forward(self, tok_embs, byte_embs=None) -> Tensor:
        return self.mixin(tok_embs, byte_embs)


class FusedFrontEnd(nn.Module):
    """Loader + embed + mixin in one library call: tokens (B,T) -> x (B,T,model_dim), with the
    token->byte gather and the pull done on the device inside that call (index kernels into scratch,
    then gather, contraction and output norm; ``MOT_LIN_FUSED=1`` selects the one-launch tile kernel in
    which the byte ids never leave LDS).  Holds the same submodules under the reference's names
    (``embed``, ``byte_mixin``), so a GPT can adopt its parameters directly; the token->byte table is
    a non-persistent integer buffer."""

    def __init__(self, dims: ModelDims, vocab_size: int, byte_params: ByteHyperparameters, ttb, max_seq_len: int = 1024,
                 pad_byte: int = 456, eot_byte: int = 457):
        super().__init__()
        self.embed = FlexibleEmbedding(dims, vocab_size, byte_params, fused=True)
        self.byte_mixin = ByteMixin(dims, max_seq_len, byte_params)
        self.byte_params = byte_params
        self.pad_byte, self.eot_byte = pad_byte, eot_byte
        self.register_buffer("ttb", _table_of(ttb).clone() if ttb is not None else None, persistent=False)

    def forward(self, tokens: Tensor, return_ids: bool = False):
        bp, emb = self.byte_params, self.embed
        if bp.byte_mixin_method == "noop":
            return self.byte_mixin(*emb(tokens, None, None))
        pull = None if not bp.pull_in else bp.padding_in          # left-padded bytes are pulled from the left
        return F_mot.embed_mix(tokens, _f32(emb.embed_tokens.weight, "token table"), _f32(emb.embed_bytes.weight, "byte table"),
                               mode="concat_linear", bpt=bp.bytes_per_token, ttb=self.ttb, pull=pull,
                               add_padded=bp.pull_in and bp.add_padded_and_pulled, pad_byte=self.pad_byte,
                               eot_byte=self.eot_byte, weight=_f32(self.byte_mixin.mixin.mixin.weight, "mixin weight"),
                               norm_tok=True, norm_byte=True, norm_out=True, return_ids=return_ids)


# ------------------------------------------------------------------------------------------------
# mathblations
# ------------------------------------------------------------------------------------------------
class LazyEmbedding(nn.Embedding):
    """Drop-in for ``GPT.wte`` / ``GPT.dte`` (model.py:304-305): forward returns an EmbedHandle that
    ``digit_mixin`` consumes; ``.weight`` stays an ordinary Parameter and may be tied (model.py:314-317)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, role: Literal["tokens", "bytes"] = "tokens", slots: int = 0):
        super().__init__(num_embeddings, embedding_dim)
        self.role, self.slots = role, slots

    def forward(self, ids: Tensor) -> EmbedHandle:  # type: ignore[override]
        if self.role == "tokens":
            return EmbedHandle(tokens=ids, tok_weight=self.weight)
        return EmbedHandle(ids_a=ids, byte_weight=self.weight, bpt=self.slots)


class DigitMixinConcat(nn.Module):  # model.py:256-268
    def __init__(self, config: GPTConfig):
        super().__init__()
        self.config = config
        if config.use_digit_self_attn:
            raise NotImplementedError("use_digit_self_attn (model.py:260,264-265) is outside the front-end path")
        self.digit_attn = nn.Identity()
        self.fc = nn.Linear(config.n_embd_tok + config.n_embd_digit * config.length_factor, config.n_embd_tok)

    def forward(self, we, de):
        if isinstance(we, EmbedHandle) and isinstance(de, EmbedHandle):
            we = we.merged(de)
            we.bpt = self.config.length_factor
            de = None
        elif isinstance(we, EmbedHandle) or isinstance(de, EmbedHandle):
            we = we.materialize()[0] if isinstance(we, EmbedHandle) else we
            de = de.materialize()[1] if isinstance(de, EmbedHandle) else de
        return _mix_concat(we, de, bpt=self.config.length_factor, weight=self.fc.weight, bias=self.fc.bias, bytes_first=True,
                           norm_out=False)


class DigitMixinNoOp(nn.Module):  # model.py:271-276
    def __init__(self, config: GPTConfig):
        super().__init__()

    def forward(self, x, *args):
        if isinstance(x, EmbedHandle):
            return F_mot.embed_mix(x.tokens, _f32(x.tok_weight, "token table"), mode="noop")
        return x


class DigitRotary(nn.Module):  # model.py:32-49
    """cos/sin tables of mathblations' Rotary: built on the host with the reference's expressions (its inv_freq is a plain
    CPU attribute, so the reference computes them there too), rounded to bfloat16 (lines 47-48) and handed to the kernel
    as fp32 -- the product of an fp32 head with a bf16 table promotes to fp32 in apply_rotary_emb (51-58)."""

    def __init__(self, dim: int, base=10000):
        super().__init__()
        self.inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        self._cached: dict = {}

    def tables(self, seq_len: int, rows: int, device) -> tuple[Tensor, Tensor]:
        """(cos, sin), each (rows * seq_len, dim / 2): the positions 0..seq_len-1 once per batch row."""
        key = (seq_len, rows, str(device))
        if key not in self._cached:
            t = torch.arange(seq_len).type_as(self.inv_freq)
            freqs = torch.outer(t, self.inv_freq)
            if len(self._cached) >= 8:   # q and k lengths of a few batch shapes; the reference caches one length
                self._cached.clear()
            self._cached[key] = tuple(f.bfloat16().float().repeat(rows, 1).contiguous().to(device) for f in (freqs.cos(), freqs.sin()))
        return self._cached[key]


class DigitCrossAttention(nn.Module):
    """mathblations/model.py:89-154 with k_gt_q (the mixin side): token t attends to its own ``length_factor`` digit
    embeddings (block mask ``q_idx == kv_idx // length_factor``, line 111).  Same parameters (c_q, c_k, c_v, c_proj:
    nn.Linear without bias) under the same names; forward takes the two embedding handles so that the gathers, input
    norms, projections, per-head norm, RoPE and the softmax run in libmot_hip.so.  head_dim 128 (n_embd / n_head of
    every run in ablations-mixin.sh and of the GPTConfig defaults)."""

    def __init__(self, config: GPTConfig):
        super().__init__()
        self.config = config
        self.n_head, self.n_embd = config.n_head, config.n_embd_tok
        assert self.n_embd % self.n_head == 0
        self.head_dim = self.n_embd // self.n_head
        self.length_factor = config.length_factor
        if not config.k_gt_q:
            raise NotImplementedError("CrossAttention with k_gt_q=False is the output side (DigitMixoutCrossAttention, model.py:213-228)")
        if self.head_dim != 128:
            raise NotImplementedError("DigitCrossAttention: head_dim 128 only (n_embd_tok / n_head)")
        self.c_q = nn.Linear(self.n_embd, self.n_embd, bias=False)
        self.c_k = nn.Linear(self.n_embd, self.n_embd, bias=False)
        self.c_v = nn.Linear(self.n_embd, self.n_embd, bias=False)
        self.c_proj = nn.Linear(self.n_embd, self.n_embd, bias=False)
        self.rotary = DigitRotary(self.head_dim)

    def forward(self, x_q, x_kv) -> Tensor:
        if not (isinstance(x_q, EmbedHandle) and isinstance(x_kv, EmbedHandle)):
            raise NotImplementedError("DigitCrossAttention: pass the handles of LazyEmbedding wte / dte (materialised inputs are not built)")
        tokens, digits = x_q.tokens, x_kv.ids_a
        if tokens.ndim == 1:
            tokens, digits = tokens[None], digits[None]
        B, T = tokens.shape
        lf = self.length_factor
        assert digits.shape[0] == B, f"Batch sizes must match: {B} vs {digits.shape[0]}"                       # model.py:129
        assert digits.shape[1] == T * lf, f"KV length {digits.shape[1]} must be {lf}x Q length {T}"             # model.py:132
        dev = tokens.device
        (cos_q, sin_q), (cos_k, sin_k) = self.rotary.tables(T, B, dev), self.rotary.tables(T * lf, B, dev)
        one = torch.ones((), dtype=torch.float32, device=dev)
        # rows of the batch laid end to end: a token only ever sees its own digits and the rotary tables repeat per row
        x = F_mot.cross_attn(tokens.reshape(1, B * T), digits.reshape(1, B * T * lf), _f32(x_q.tok_weight, "token table"),
                             _f32(x_kv.byte_weight, "digit table"), q_w=self.c_q.weight, kv_w=torch.stack([self.c_k.weight, self.c_v.weight]),
                             proj_w=self.c_proj.weight, lambda_factor=one, cos_q=cos_q, sin_q=sin_q, cos_k=cos_k, sin_k=sin_k,
                             bpt=lf, n_heads=self.n_head, norm_tok=True, norm_byte=True, head_layout="per_token")
        return x.view(B, T, self.n_embd)


class DigitMixinCrossAttention(nn.Module):  # model.py:239-253
    def __init__(self, config: GPTConfig):
        assert config.n_embd_digit == config.n_embd_tok
        super().__init__()
        self.config = config
        if config.use_digit_self_attn:
            raise NotImplementedError("use_digit_self_attn (model.py:244,248-249) is outside the front-end path")
        self.digit_attn = nn.Identity()
        self.cross_attn = DigitCrossAttention(config)

    def forward(self, we, de) -> Tensor:
        # F.rms_norm of both inputs (model.py:250-253) happens inside the launch
        return self.cross_attn(x_q=we, x_kv=de)


def make_digit_mixin(config: GPTConfig):  # model.py:279-284
    return {"noop": DigitMixinNoOp, "cross_attn": DigitMixinCrossAttention, "concat": DigitMixinConcat}[config.digit_mixin_method](config)


class DigitFrontEnd(nn.Module):
    """``wte`` / ``dte`` / ``digit_mixin`` of mathblations' GPT and the first lines of its forward
    (model.py:304-306, 319-327), nothing else of that model."""

    def __init__(self, config: GPTConfig):
        super().__init__()
        self.config = config
        self.wte = LazyEmbedding(config.vocab_size, config.n_embd_tok, "tokens")
        self.dte = (LazyEmbedding(14, config.n_embd_digit, "bytes", config.length_factor)
                    if config.digit_mixin_method != "noop" else nn.Identity())
        self.digit_mixin = make_digit_mixin(config)

    def forward(self, idx: Tensor, digits: Tensor | None = None) -> Tensor:
        if self.config.digit_mixin_method != "noop":
            assert digits is not None, "Digits must be provided"
        we = self.wte(idx)
        de = self.dte(digits)
        return self.digit_mixin(we, de)


# ------------------------------------------------------------------------------------------------
# modded-nanogpt SUM mixin
# ------------------------------------------------------------------------------------------------
class SumFrontEnd(nn.Module):
    """``x = norm(embed_tokens(tok) + concat_k embed_bytes(byte_k))`` (runs/71_*.py:227-230, 312-314) with
    the per-token byte semantics of train_gpt.py:442 / runs/7_*.py:227-231 (SURVEY section 7, quirk iii).
    variant "71": plain; "71041": norm each embedding, scale by learned scalars, norm the sum
    (runs/71041_*.py:311-313); "71081": s_t*norm(E_t) + s_b*concat(norm(E_b)), no outer norm
    (runs/71081_*.py:302-315).  byte_dim * bytes_per_token must equal model_dim."""

    def __init__(self, token_vocab_size: int, byte_vocab_size: int, model_dim: int, byte_dim: int, bytes_per_token: int = 16,
                 variant: Literal["71", "71041", "71081"] = "71", ttb=None, pad_byte: int = 456, eot_byte: int = 457):
        super().__init__()
        assert byte_dim * bytes_per_token == model_dim
        self.embed_tokens = nn.Embedding(token_vocab_size, model_dim)
        self.embed_bytes = nn.Embedding(byte_vocab_size, byte_dim)
        self.variant, self.bpt, self.pad_byte, self.eot_byte = variant, bytes_per_token, pad_byte, eot_byte
        self.scalars = nn.Parameter(torch.ones(2)) if variant != "71" else None  # [-2] bytes, [-1] tokens
        self.register_buffer("ttb", _table_of(ttb).clone() if ttb is not None else None, persistent=False)

    def forward(self, token_inputs: Tensor, byte_inputs: Tensor | None = None) -> Tensor:
        """token_inputs (T,) or (B,T); byte_inputs (.., T*bpt) per-token-ordered pulled ids, or None to
        produce them in-kernel from the attached token->byte table."""
        pre = self.variant != "71"
        kw = dict(norm_tok=pre, norm_byte=pre, norm_out=self.variant != "71081")
        if pre:
            kw.update(scale_tok=self.scalars[-1:], scale_byte=self.scalars[-2:-1])
        if byte_inputs is None:
            kw.update(ttb=self.ttb, pull="left", pad_byte=self.pad_byte, eot_byte=self.eot_byte)
        else:
            kw.update(ids_a=byte_inputs.to(torch.int64).reshape(1 if token_inputs.ndim == 1 else token_inputs.shape[0], -1))
        return F_mot.embed_mix(token_inputs, _f32(self.embed_tokens.weight, "token table"),
                               _f32(self.embed_bytes.weight, "byte table"), mode="sum", bpt=self.bpt, **kw)


# ------------------------------------------------------------------------------------------------
# Llama character mixer (inference/inference.py): BASELINE config 5's front-end
# ------------------------------------------------------------------------------------------------
@dataclass
class ModelArgs:  # inference.py:36-43
    version: Literal["no_residual", "one_residual", "two_residual"]
    n_heads: int = 32
    dim: int = 2048
    intermediate_dim: int = 8192
    head_dim: int = 64
    norm_eps: float = 1e-5


class RMSNorm(nn.Module):  # inference.py:119-132 (a plain torch module: the trunk uses it on arbitrary activations)
    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def _norm(self, x):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.eps)

    def forward(self, x):
        return self._norm(x.float()).type_as(x) * self.weight


class FeedForward(nn.Module):  # inference.py:135-144, SwiGLU: plain dense layers, outside the embedding path (torch ops)
    def __init__(self, args: ModelArgs):
        super().__init__()
        self.w1 = nn.Linear(args.dim, args.intermediate_dim, bias=False)
        self.w2 = nn.Linear(args.intermediate_dim, args.dim, bias=False)
        self.w3 = nn.Linear(args.dim, args.intermediate_dim, bias=False)

    def forward(self, x) -> Tensor:
        return self.w2(F.silu(self.w1(x)) * self.w3(x))


class TokenMixByCharBMM(nn.Module):
    """inference.py:146-224: the parameters (wq, wk, wv, wo; window_size 8) under the reference's names.  The work itself
    -- including the two RMSNorms in front and the residuals behind, which the reference's block applies around this module --
    runs in ONE library call issued by TokenMixByCharBMMBlock.forward; this class only holds the weights."""

    def __init__(self, args: ModelArgs):
        super().__init__()
        self.args = args
        self.n_heads, self.head_dim = args.n_heads, args.head_dim
        self.bmm_dim = args.n_heads * args.head_dim
        self.wq = nn.Linear(args.dim, self.bmm_dim, bias=False)
        self.wk = nn.Linear(args.dim, self.bmm_dim, bias=False)
        self.wv = nn.Linear(args.dim, self.bmm_dim, bias=False)
        self.wo = nn.Linear(self.bmm_dim, args.dim, bias=False)
        self.attention_scores = None      # the reference keeps the last softmax here (line 225); the fused kernel does not materialise it
        self.window_size = 8


class TokenMixByCharBMMBlock(nn.Module):
    """inference.py:226-270 with the same parameters and state-dict keys.  ``forward(toks, chars, rotary_emb_fn=None)`` takes
    the EmbedHandles of the two embeddings (CharMixerFrontEnd below / LazyEmbedding) instead of materialised (b, t, d) and
    (b, t, c_v, d) tensors: keys and values are projected once per character-table ROW, which needs the ids.
    ``rotary_emb_fn`` is accepted and unused: the reference rotates the query and each of its keys by the same angle (both sit
    at the query's position, lines 209-217), which leaves their product unchanged (mot_swa.hip).  Forward only."""

    def __init__(self, args: ModelArgs):
        super().__init__()
        self.n_heads, self.dim = args.n_heads, args.dim
        self.tok_attention = TokenMixByCharBMM(args)
        self.feed_forward = FeedForward(args=args)
        self.attention_norm = RMSNorm(args.dim, eps=args.norm_eps)
        self.char_norm = RMSNorm(args.dim, eps=args.norm_eps)
        self.ffn_norm = RMSNorm(args.dim, eps=args.norm_eps)
        self.args, self.version = args, args.version
        self._kv_cache: dict = {}   # per-character K / V tables of the mixer, kept across calls while their inputs are unchanged
        if self.version in ["two_residual", "no_residual"]:
            self.lambda_tok = nn.Parameter(torch.ones(1))
            self.lambda_char = nn.Parameter(torch.ones(1))
        if self.version == "two_residual":
            self.register_buffer("current_step", torch.tensor(0))

    def get_residual_scale(self):  # inference.py:247-250
        return min(self.current_step.item() / 5000, 1.0)

    def mix(self, toks: EmbedHandle, chars: EmbedHandle) -> Tensor:
        """``h`` of the reference's forward (lines 260-267): attention + residuals, before the feed-forward."""
        if not (isinstance(toks, EmbedHandle) and isinstance(chars, EmbedHandle)):
            raise NotImplementedError("TokenMixByCharBMMBlock: pass the embedding handles (token ids + table, character ids + table); "
                                      "materialised embeddings are not built")
        ta = self.tok_attention
        tokens, cid = toks.tokens, chars.ids_a
        if tokens.ndim == 1:
            tokens, cid = tokens[None], cid[None]
        cid = cid.reshape(tokens.shape[0], tokens.shape[1], -1)
        two = self.version == "two_residual"
        return F_mot.char_swa(tokens, cid, toks.tok_weight, chars.byte_weight, attn_norm_w=self.attention_norm.weight,
                              char_norm_w=self.char_norm.weight, wq=ta.wq.weight, wk=ta.wk.weight, wv=ta.wv.weight, wo=ta.wo.weight,
                              n_heads=ta.n_heads, head_dim=ta.head_dim, window=ta.window_size, norm_eps=self.attention_norm.eps,
                              version=self.version, lambda_tok=self.lambda_tok if two else None, lambda_char=self.lambda_char if two else None,
                              kv_cache=self._kv_cache)   # (forward-only path: the per-character K / V tables are kept while their inputs are unchanged)

    def forward(self, toks, chars, rotary_emb_fn=None) -> Tensor:
        h = self.mix(toks, chars)
        return h + self.feed_forward.forward(self.ffn_norm(h))          # line 269: plain torch


class CharMixerFrontEnd(nn.Module):
    """The embedding front of CustomLlamaModel (inference.py:273-335): ``embed_tokens`` (the Llama table), ``char_embeddings``
    (132 rows) and ``char_token_mixer``, and the first lines of its forward (323-335) -> ``mixed_embeddings`` for the trunk.
    The pretrained trunk itself (``self.model``) is out of scope; adopt its ``embed_tokens.weight`` into this table."""

    def __init__(self, vocab_size: int, char_vocab_size: int, model_args: ModelArgs, max_char: int = 8):
        super().__init__()
        self.embed_tokens = LazyEmbedding(vocab_size, model_args.dim, "tokens")
        self.char_embeddings = LazyEmbedding(char_vocab_size, model_args.dim, "bytes", max_char)
        self.max_char = max_char
        self.char_token_mixer = TokenMixByCharBMMBlock(model_args)

    def forward(self, input_ids: Tensor, char_ids: Tensor) -> Tensor:
        return self.char_token_mixer(self.embed_tokens(input_ids), self.char_embeddings(char_ids), None)
