"""Tensor-level wrappers of the C ABI: shape/dtype/device checks, output allocation on the
caller's device and stream (torch caching allocator), then one call into libmot_hip.so.

Nothing here computes: every result comes from a HIP kernel.  Functions are wrapped in
``torch.compiler.disable`` so that a caller under ``torch.compile`` (train_gpt.py:1195) sees
them as opaque calls.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import torch

from . import _capi as capi

# Kernel-selection switches, read ONCE at import (the C library itself reads no environment variable; the choice travels
# in MotEmbedMixDesc.flags).  Every call can override them with one_launch= / mean_generic= / du_fp32=.
_ENV_FLAGS = ((capi.FLAG_LINEAR_ONE_LAUNCH if os.environ.get("MOT_LIN_FUSED") else 0)
              | (capi.FLAG_MEAN_GENERIC if os.environ.get("MOT_NO_MEAN_LDS") else 0)
              | (capi.FLAG_BWD_DU_FP32 if os.environ.get("MOT_NO_DU16") else 0)
              | (capi.FLAG_LINEAR_COMPOSED if os.environ.get("MOT_LIN_COMPOSED") else 0))


def _flags(one_launch=None, mean_generic=None, du_fp32=None, composed=None) -> int:
    f = _ENV_FLAGS
    for bit, v in ((capi.FLAG_LINEAR_ONE_LAUNCH, one_launch), (capi.FLAG_MEAN_GENERIC, mean_generic), (capi.FLAG_BWD_DU_FP32, du_fp32),
                   (capi.FLAG_LINEAR_COMPOSED, composed)):
        if v is not None:
            f = (f | bit) if v else (f & ~bit)
    return f


_MODES = {"noop": capi.MIX_NOOP, "sum": capi.MIX_SUM, "mean": capi.MIX_MEAN, "concat_linear": capi.MIX_CONCAT_LINEAR}
_PULLS = {None: capi.PULL_NONE, "none": capi.PULL_NONE, "left": capi.PULL_LEFT, "right": capi.PULL_RIGHT}


def _contig(t: torch.Tensor, dtype: torch.dtype, what: str) -> torch.Tensor:
    if t.dtype != dtype:
        raise TypeError(f"{what}: expected {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _table(t: torch.Tensor, what: str, like: torch.dtype | None = None) -> torch.Tensor:
    """fp32 (parity mode) or bf16 (what the training loop runs) table; all tables of a call agree."""
    capi.dtype_code(t.dtype)
    if like is not None and t.dtype != like:
        raise TypeError(f"{what}: {t.dtype} but the token table is {like}")
    return t if t.is_contiguous() else t.contiguous()


def _int_table(table: torch.Tensor, what: str) -> torch.Tensor:
    if table.dtype not in (torch.int16, torch.int32):
        raise TypeError(f"{what}: token->byte table must be int16 or int32, got {table.dtype}")
    if table.ndim != 2:
        raise ValueError(f"{what}: token->byte table must be (vocab, bytes_per_token)")
    return table if table.is_contiguous() else table.contiguous()


@torch.compiler.disable
def tokens_to_bytes(tokens: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """int64 byte ids of `tokens` (any shape) from an integer table (vocab, bpt) -> tokens.shape + (bpt,)."""
    table = _int_table(table, "tokens_to_bytes")
    dev = capi.require_device(tokens, table)
    tok = tokens.to(torch.int32) if tokens.dtype != torch.int32 else tokens
    tok = tok if tok.is_contiguous() else tok.contiguous()
    bpt = table.shape[1]
    out = torch.empty(tok.shape + (bpt,), dtype=torch.int64, device=dev)
    capi.check(capi.lib.mot_tokens_to_bytes(capi.ptr(tok), tok.numel(), capi.ptr(table), table.element_size(),
                                            table.shape[0], bpt, capi.ptr(out), capi.ptr(capi.status_word(dev)),
                                            capi.stream_of(dev)))
    capi.after_call(dev)
    return out


@torch.compiler.disable
def pull_bytes(byte_tensor: torch.Tensor, bytes_per_token: int, pad_byte: int, eot_byte: int, direction: str) -> torch.Tensor:
    """pull_from_left / pull_from_right on a (B, T) int64 tensor (data_creation.py:71-305)."""
    if byte_tensor.ndim != 2:
        raise ValueError("byte_tensor must be (B, T)")
    B, T = byte_tensor.shape
    if T == 0:
        return byte_tensor  # data_creation.py:82-83, 190
    dev = capi.require_device(byte_tensor)
    x = _contig(byte_tensor, torch.int64, "byte_tensor")
    out = torch.empty_like(x)
    capi.check(capi.lib.mot_pull_bytes(capi.ptr(x), capi.ptr(out), B, T, int(bytes_per_token), int(pad_byte),
                                       int(eot_byte), _PULLS[direction], capi.stream_of(dev)))
    return out


@torch.compiler.disable
def create_batch(tokens: torch.Tensor, table_left: torch.Tensor, table_right: torch.Tensor, pad_byte: int, eot_byte: int) -> torch.Tensor:
    """(B, T, 1+4*bpt) int64 packed batch (data_creation.py:308-330) in one launch."""
    tl, tr = _int_table(table_left, "create_batch"), _int_table(table_right, "create_batch")
    if tl.shape != tr.shape or tl.dtype != tr.dtype:
        raise ValueError("left/right tables must have the same shape and dtype")
    dev = capi.require_device(tokens, tl, tr)
    tok = _contig(tokens.to(torch.int32), torch.int32, "tokens")
    B, T = tok.shape
    bpt = tl.shape[1]
    out = torch.empty((B, T, 1 + 4 * bpt), dtype=torch.int64, device=dev)
    capi.check(capi.lib.mot_create_batch(capi.ptr(tok), B, T, capi.ptr(tl), capi.ptr(tr), tl.element_size(), tl.shape[0],
                                         bpt, int(pad_byte), int(eot_byte), capi.ptr(out),
                                         capi.ptr(capi.status_word(dev)), capi.stream_of(dev)))
    capi.after_call(dev)
    return out


@torch.compiler.disable
def char_matrix(codes: torch.Tensor, tok_offsets: torch.Tensor, seq_offsets: torch.Tensor, n_seqs: int, seq_len: int, max_char: int,
                leading_space: int, bos_token_id: int, eos_token_id: int) -> torch.Tensor:
    """(n_seqs, seq_len, max_char) int64 character ids (inference.py:56-67, 79-96); see mot_char_matrix in include/mot.h."""
    dev = capi.require_device(codes, tok_offsets, seq_offsets)
    c, to, so = _contig(codes, torch.int32, "codes"), _contig(tok_offsets, torch.int64, "tok_offsets"), _contig(seq_offsets, torch.int64, "seq_offsets")
    if so.numel() != n_seqs + 1:
        raise ValueError("seq_offsets must hold n_seqs + 1 entries")
    out = torch.empty((n_seqs, seq_len, max_char), dtype=torch.int64, device=dev)
    capi.check(capi.lib.mot_char_matrix(capi.ptr(c), capi.ptr(to), capi.ptr(so), int(n_seqs), int(seq_len), int(max_char), int(leading_space),
                                        int(bos_token_id), int(eos_token_id), capi.ptr(out), capi.stream_of(dev)))
    return out


@torch.compiler.disable
def gather_rows(table: torch.Tensor, ids: torch.Tensor, ids_b: torch.Tensor | None = None, *, rms_norm: bool = False,
                eps: float | None = None, scale: torch.Tensor | None = None) -> torch.Tensor:
    """scale * rms_norm?(table[ids] (+ table[ids_b])) -> ids.shape + (dim,)  (train_gpt.py:342-379)."""
    dev = capi.require_device(table, ids, ids_b, scale)
    tab = _table(table, "table")
    if ids.dtype not in (torch.int32, torch.int64):
        raise TypeError(f"ids must be int32/int64, got {ids.dtype}")
    ia = ids if ids.is_contiguous() else ids.contiguous()
    ib = None
    if ids_b is not None:
        if ids_b.shape != ids.shape or ids_b.dtype != ids.dtype:
            raise ValueError("ids_b must match ids in shape and dtype")
        ib = ids_b if ids_b.is_contiguous() else ids_b.contiguous()
    out = torch.empty(ids.shape + (tab.shape[1],), dtype=tab.dtype, device=dev)
    capi.check(capi.lib.mot_gather_rows(capi.ptr(ia), capi.ptr(ib), ia.element_size(), ia.numel(), capi.ptr(tab),
                                        tab.shape[0], tab.shape[1], int(rms_norm), float(eps or 0.0), capi.ptr(scale),
                                        capi.ptr(out), capi.ptr(capi.status_word(dev)), capi.dtype_code(tab.dtype),
                                        capi.stream_of(dev)))
    capi.after_call(dev)
    return out


@dataclass
class MixResult:
    x: torch.Tensor
    ids_padded: torch.Tensor | None = None
    ids_pulled: torch.Tensor | None = None


_workspaces: dict[tuple[int, int], torch.Tensor] = {}
_retired: dict[tuple[int, int], list[torch.Tensor]] = {}


def _workspace(dev: torch.device, nbytes: int) -> torch.Tensor | None:
    """Per-(device, stream) scratch reused across calls (kernels on one stream are ordered).

    A captured hipGraph bakes the raw workspace pointer into its kernel nodes, so a buffer that a larger request
    replaces is NOT handed back to the allocator: it moves to a retired list and stays valid for whatever graph
    still points at it (workspaces only ever grow, geometrically, so the list holds a handful of buffers whose
    sizes sum to less than the live one).  `release_workspaces()` drops everything once no graph needs them."""
    if nbytes == 0:
        return None
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), capi.stream_of(dev))
    w = _workspaces.get(key)
    if w is None or w.numel() < nbytes:
        if w is not None:
            _retired.setdefault(key, []).append(w)
            nbytes = max(nbytes, 2 * w.numel())
        w = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=dev)
        _workspaces[key] = w
    return w


def release_workspaces() -> None:
    """Frees every library workspace (live and retired).  Call only when no captured graph that used them will be replayed."""
    _workspaces.clear()
    _retired.clear()


@torch.compiler.disable
def _embed_mix_fwd(tokens: torch.Tensor, tok_table: torch.Tensor, byte_table: torch.Tensor | None = None, *,
              mode: str, bpt: int = 0,
              ttb: torch.Tensor | None = None, pull: str | None = None, add_padded: bool = False,
              pad_byte: int = 456, eot_byte: int = 457,
              ids_a: torch.Tensor | None = None, ids_b: torch.Tensor | None = None,
              weight: torch.Tensor | None = None, bias: torch.Tensor | None = None, bytes_first: bool = False,
              norm_tok: bool = False, norm_byte: bool = False, norm_out: bool = False, eps: float | None = None,
              scale_tok: torch.Tensor | None = None, scale_byte: torch.Tensor | None = None,
              out: torch.Tensor | None = None, return_ids: bool = False,
              counters: torch.Tensor | None = None, row_rnorm: torch.Tensor | None = None,
              one_launch: bool | None = None, mean_generic: bool | None = None, composed: bool | None = None,
              _plan: bool = False) -> torch.Tensor | MixResult:
    """One fused launch of mot_embed_mix_fwd; see include/mot.h for the per-token formula.

    tokens (B, T) integer.  Byte ids either come from `ttb` (+ `pull` = "left" | "right" | None,
    + `add_padded`) inside the kernel, or are given as `ids_a` / `ids_b` (B, T*bpt) int64.
    `scale_*` are 0-dim/1-element DEVICE tensors (learned scalars are read on the device).
    `composed` (bf16 concat_linear: the separate gather / GEMM / norm kernels instead of the one gather-GEMM),
    `one_launch` (concat_linear: the one-launch tile kernel instead of the composed kernels) and `mean_generic` (mean: the
    whole-row kernel instead of the LDS column-slice kernel) override the import-time defaults (MotEmbedMixDesc.flags).
    """
    m = _MODES[mode]
    if tokens.ndim == 1:
        tokens = tokens[None]
    if tokens.ndim != 2:
        raise ValueError("tokens must be (B, T) or (T,)")
    dev = capi.require_device(tokens, tok_table, byte_table, ttb, ids_a, ids_b, weight, bias, scale_tok, scale_byte)
    for what, sc in (("scale_tok", scale_tok), ("scale_byte", scale_byte)):
        if sc is not None and (sc.dtype != torch.float32 or sc.numel() != 1):
            raise TypeError(f"{what}: expected a 1-element float32 device tensor, got {sc.dtype} x {sc.numel()}")
    tok = tokens.to(torch.int32) if tokens.dtype != torch.int32 else tokens
    tok = tok if tok.is_contiguous() else tok.contiguous()
    B, T = tok.shape
    tt = _table(tok_table, "tok_table")
    fdt = tt.dtype
    d = capi.MotEmbedMixDesc()
    d.struct_size = C.sizeof(capi.MotEmbedMixDesc)
    d.dtype = capi.dtype_code(fdt)
    d.flags = _flags(one_launch, mean_generic, composed=composed)
    d.n_rows, d.tokens_per_row, d.bpt, d.mode = B, T, int(bpt), m
    d.tokens = capi.ptr(tok)
    d.tok_table, d.tok_rows, d.tok_dim = capi.ptr(tt), tt.shape[0], tt.shape[1]
    keep = [tok, tt]
    ids_padded = ids_pulled = None
    if m != capi.MIX_NOOP:
        if byte_table is None:
            raise ValueError("byte_table is required unless mode == 'noop'")
        bt = _table(byte_table, "byte_table", fdt)
        keep.append(bt)
        d.byte_table, d.byte_rows, d.byte_dim = capi.ptr(bt), bt.shape[0], bt.shape[1]
        if ttb is not None:
            tab = _int_table(ttb, "embed_mix")
            if tab.shape[1] != bpt:
                raise ValueError(f"ttb has {tab.shape[1]} slots per token, bpt={bpt}")
            keep.append(tab)
            d.id_source, d.pull_dir = capi.IDS_FROM_TTB, _PULLS[pull]
            d.ttb, d.ttb_rows, d.ttb_elem_bytes = capi.ptr(tab), tab.shape[0], tab.element_size()
            d.add_padded = int(add_padded)
            if return_ids:
                ids_padded = torch.empty((B, T * bpt), dtype=torch.int64, device=dev)
                ids_pulled = torch.empty((B, T * bpt), dtype=torch.int64, device=dev)
                d.out_ids_padded, d.out_ids_pulled = capi.ptr(ids_padded), capi.ptr(ids_pulled)
        else:
            if ids_a is None:
                raise ValueError("either ttb or ids_a must be given")
            ia = _contig(ids_a, torch.int64, "ids_a")
            assert ia.numel() == B * T * bpt, "byte ids must hold bytes_per_token ids per token"
            keep.append(ia)
            d.id_source, d.ids_a = capi.IDS_GIVEN, capi.ptr(ia)
            if ids_b is not None:
                ib = _contig(ids_b, torch.int64, "ids_b")
                assert ib.numel() == ia.numel()
                keep.append(ib)
                d.ids_b = capi.ptr(ib)
    d.pad_byte, d.eot_byte = int(pad_byte), int(eot_byte)
    if m == capi.MIX_CONCAT_LINEAR:
        if weight is None:
            raise ValueError("weight is required for mode == 'concat_linear'")
        w = _table(weight, "weight", fdt)
        keep.append(w)
        d.weight, d.model_dim = capi.ptr(w), w.shape[0]
        if w.shape[1] != tt.shape[1] + bpt * d.byte_dim:
            raise ValueError(f"weight has {w.shape[1]} input features, expected {tt.shape[1]} + {bpt}*{d.byte_dim}")
        if bias is not None:
            bs = _table(bias, "bias", fdt)
            keep.append(bs)
            d.bias = capi.ptr(bs)
        d.bytes_first = int(bytes_first)
    else:
        d.model_dim = tt.shape[1]
    d.norm_tok, d.norm_byte, d.norm_out = int(norm_tok), int(norm_byte), int(norm_out)
    d.eps = float(eps or 0.0)
    d.scale_tok, d.scale_byte = capi.ptr(scale_tok), capi.ptr(scale_byte)
    if out is None:
        out = torch.empty((B, T, d.model_dim), dtype=fdt, device=dev)
    else:
        if out.shape != (B, T, d.model_dim) or out.dtype != fdt or not out.is_contiguous() or out.device != dev:
            raise ValueError("out must be a contiguous (B, T, model_dim) tensor of the tables' dtype on their device")
    d.out = capi.ptr(out)
    if counters is not None:
        if counters.dtype != torch.int64 or counters.numel() < 4 or counters.device != dev:
            raise ValueError("counters must be an int64[4] tensor on the inputs' device")
        d.counters = capi.ptr(counters)
    d.status = capi.ptr(capi.status_word(dev))
    if row_rnorm is not None:   # (B, T) fp32, written by the concat_linear kernel when norm_out (saved for the backward)
        assert row_rnorm.dtype == torch.float32 and row_rnorm.numel() == B * T and row_rnorm.is_contiguous()
        d.out_row_rnorm = capi.ptr(row_rnorm)
    ws = _workspace(dev, capi.lib.mot_embed_mix_workspace_bytes(C.byref(d)))
    if ws is not None:
        d.workspace, d.workspace_bytes = capi.ptr(ws), ws.numel()
        keep.append(ws)
    result = MixResult(out, ids_padded, ids_pulled) if return_ids else out
    if _plan:
        return EmbedMixPlan(d, keep + [out, ids_padded, ids_pulled, counters, row_rnorm, scale_tok, scale_byte], dev, result)
    capi.check(capi.lib.mot_embed_mix_fwd(C.byref(d), capi.stream_of(dev)))
    capi.after_call(dev)
    return result


class EmbedMixPlan:
    """A validated, fully bound descriptor of one fused-forward call: ``plan()`` is a single C call on the
    current stream (a few microseconds of host time instead of the ~50 us the checked wrapper spends),
    reading the CURRENT contents of the bound tensors (update tokens / tables in place between calls).
    Build with :func:`embed_mix_plan`; forward only (no autograd)."""

    def __init__(self, desc, keepalive, device, result):
        self._d, self._keep, self._dev, self.result = desc, keepalive, device, result
        self._ref = C.byref(desc)

    def __call__(self):
        rc = capi.lib.mot_embed_mix_fwd(self._ref, torch.cuda.current_stream(self._dev).cuda_stream)
        if rc:
            capi.check(rc)
        return self.result


def embed_mix_plan(tokens, tok_table, byte_table=None, **kw) -> EmbedMixPlan:
    """Same arguments as :func:`embed_mix`; returns an :class:`EmbedMixPlan` instead of launching."""
    return _embed_mix_fwd(tokens, tok_table, byte_table, _plan=True, **kw)


@torch.compiler.disable
def token_order(tokens: torch.Tensor, tok_rows: int) -> torch.Tensor:
    """One call of mot_token_order: the batch's positions grouped by token id (opaque int32 buffer), which the table-gradient
    scatter of every backward over these tokens walks.  Depends on `tokens` only; runs on the current stream."""
    dev = capi.require_device(tokens)
    tok = tokens.to(torch.int32) if tokens.dtype != torch.int32 else tokens
    tok = tok if tok.is_contiguous() else tok.contiguous()
    order = torch.empty(capi.lib.mot_token_order_ints(tok.numel(), int(tok_rows)), dtype=torch.int32, device=dev)
    capi.check(capi.lib.mot_token_order(capi.ptr(tok), tok.numel(), int(tok_rows), capi.ptr(order), capi.ptr(capi.status_word(dev)),
                                        capi.stream_of(dev)))
    capi.after_call(dev)
    return order


class _TokenOrderCache:
    """The token order of the last few token tensors seen by the autograd node, produced BESIDE the forward: on a side stream that
    waits for the tokens and runs while the forward kernel streams (the sort is three small latency-bound kernels, the forward is
    HBM-bound), so the backward finds it ready instead of spending 0.08 ms of a 0.55 ms call on it.  One order serves every
    backward over the same tokens: several embedding tables indexed by one token tensor (modded-nanogpt/runs/71_*.py value
    embeddings), gradient accumulation, the two front-ends of a mixin/mixout pair.  An entry is valid for the SAME tensor object
    at the SAME version with the same table height; entries keep their token tensor alive (a few MB), so an address cannot be
    reused under a stale entry."""

    def __init__(self, keep: int = 4):
        self.keep, self.entries, self.side = keep, [], {}

    def get(self, tokens: torch.Tensor, tok_rows: int):
        for e in self.entries:
            if e[0] is tokens and e[1] == tokens._version and e[2] == tok_rows:
                return e[3], e[4]
        dev = tokens.device
        cur = torch.cuda.current_stream(dev)
        if torch.cuda.is_current_stream_capturing():          # inside a graph capture: no side stream, the sort is captured in line
            order, ev = token_order(tokens, tok_rows), None
        else:
            side = self.side.get(dev.index)
            if side is None:
                side = self.side[dev.index] = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)                              # the tokens are ready when the current stream gets here
            with torch.cuda.stream(side):
                order = token_order(tokens, tok_rows)
                ev = torch.cuda.Event()
                ev.record(side)
        self.entries.insert(0, (tokens, tokens._version, tok_rows, order, ev))
        del self.entries[self.keep:]
        return order, ev

    def clear(self):
        self.entries.clear()


_token_orders = _TokenOrderCache()
_HOIST_SORT = not os.environ.get("MOT_NO_ORDER_HOIST")            # dev switch: let every backward group the positions itself


_BWD_MODES = ("sum", "noop", "concat_linear", "mean")


class _EmbedMixFn(torch.autograd.Function):
    """Autograd node of the fused front-end: forward = one mot_embed_mix_fwd launch, backward = one
    mot_embed_mix_bwd launch (dense gradients, like nn.Embedding(sparse=False) in the reference)."""

    @staticmethod
    def forward(ctx, tok_table, byte_table, scale_tok, scale_byte, weight, bias, tokens, kw):
        kw = dict(kw)
        mode = kw["mode"]
        want_ids = kw.get("ttb") is not None and mode != "noop"
        user_return_ids = kw.pop("return_ids", False)
        kw.pop("weight", None); kw.pop("bias", None)
        rn = None
        if mode == "concat_linear" and kw.get("norm_out"):
            t2 = tokens if tokens.ndim == 2 else tokens[None]
            rn = torch.empty(t2.shape, dtype=torch.float32, device=tok_table.device)
        r = _embed_mix_fwd(tokens, tok_table.detach(), None if byte_table is None else byte_table.detach(),
                           scale_tok=None if scale_tok is None else scale_tok.detach(),
                           scale_byte=None if scale_byte is None else scale_byte.detach(),
                           weight=None if weight is None else weight.detach(), bias=None if bias is None else bias.detach(),
                           return_ids=want_ids or user_return_ids, row_rnorm=rn, **kw)
        x = r.x if isinstance(r, MixResult) else r
        ids_a, ids_b = kw.get("ids_a"), kw.get("ids_b")
        if want_ids:   # the byte ids the kernel produced in LDS, written out once for the backward
            ids_a = r.ids_pulled if kw.get("pull") not in (None, "none") else r.ids_padded
            ids_b = r.ids_padded if kw.get("add_padded") else None
        # the positions grouped by token id, for the backward: requested here so that it runs beside the forward launch above
        ctx.order = _token_orders.get(tokens, tok_table.shape[0]) if _HOIST_SORT else None
        ctx.save_for_backward(tok_table, byte_table, scale_tok, scale_byte, tokens, ids_a, ids_b, weight, bias,
                              x if mode == "concat_linear" else None, rn)
        ctx.kw = {k: kw[k] for k in ("mode", "bpt", "norm_tok", "norm_byte", "norm_out", "eps", "bytes_first") if k in kw}
        if user_return_ids:
            ctx.mark_non_differentiable(r.ids_padded, r.ids_pulled)
            return x, r.ids_padded, r.ids_pulled
        return x

    @staticmethod
    def backward(ctx, gx, *_):
        tok_table, byte_table, scale_tok, scale_byte, tokens, ids_a, ids_b, weight, bias, x, rn = ctx.saved_tensors
        # Parameters whose .grad is managed by grad_sync.GradBucket (an explicit opt-in: GradBucket(..., in_place=True) marks them)
        # take the kernel's += directly (mot_embed_mix_bwd only ever adds into its outputs): no temporary table-sized gradient,
        # no zero fill, no AccumulateGrad pass over it.  The price, documented on GradBucket: for those parameters autograd sees
        # no gradient -- tensor hooks and post-accumulate-grad hooks do not fire, and torch.autograd.grad(...) would find .grad
        # changed -- so the direct path is refused under torch.autograd.grad (ctx.needs_input_grad is all there is to go by: it is
        # taken only when backward() was asked to accumulate into leaves).  Everything else gets a fresh fp32 gradient handed to
        # autograd as usual.
        direct = {k: p.grad for k, p in (("tok_table", tok_table), ("byte_table", byte_table), ("weight", weight), ("bias", bias))
                  if _accumulates_in_place(p)}
        order = None
        if ctx.order is not None:
            order, ev = ctx.order
            if ev is not None:
                torch.cuda.current_stream(gx.device).wait_event(ev)
            order.record_stream(torch.cuda.current_stream(gx.device))
        g = embed_mix_backward(gx, tokens, tok_table.detach(), None if byte_table is None else byte_table.detach(),
                               ids_a=ids_a, ids_b=ids_b, token_order=order,
                               scale_tok=None if scale_tok is None else scale_tok.detach(),
                               scale_byte=None if scale_byte is None else scale_byte.detach(),
                               weight=None if weight is None else weight.detach(), bias=None if bias is None else bias.detach(),
                               out=None if x is None else x.detach(), row_rnorm=rn, into=direct, **ctx.kw)
        def like(k, p):  # bf16 parameters get their gradient rounded once, from the fp32 sums
            t = g.get(k)
            return None if t is None or p is None or k in direct else t.to(p.dtype).reshape(p.shape)
        return (like("tok_table", tok_table), like("byte_table", byte_table), like("scale_tok", scale_tok),
                like("scale_byte", scale_byte), like("weight", weight), like("bias", bias), None, None)


def _accumulates_in_place(p) -> bool:
    """True for a leaf parameter that grad_sync.GradBucket has bound to its flat buffer: fp32, contiguous .grad present."""
    if p is None or not getattr(p, "_mot_grad_in_place", False) or not p.requires_grad or not p.is_leaf:
        return False
    gr = p.grad
    return gr is not None and gr.dtype == torch.float32 and gr.is_contiguous() and gr.shape == p.shape and gr.device == p.device


@torch.compiler.disable
def embed_mix_backward(grad_out: torch.Tensor, tokens: torch.Tensor, tok_table: torch.Tensor, byte_table: torch.Tensor | None = None, *,
                       mode: str, bpt: int = 0, ids_a: torch.Tensor | None = None, ids_b: torch.Tensor | None = None,
                       norm_tok: bool = False, norm_byte: bool = False, norm_out: bool = False, eps: float | None = None,
                       scale_tok: torch.Tensor | None = None, scale_byte: torch.Tensor | None = None,
                       weight: torch.Tensor | None = None, bias: torch.Tensor | None = None, bytes_first: bool = False,
                       out: torch.Tensor | None = None, row_rnorm: torch.Tensor | None = None,
                       into: dict | None = None, du_fp32: bool | None = None, one_launch: bool | None = None,
                       token_order: torch.Tensor | None = None) -> dict:
    """One launch of mot_embed_mix_bwd.  Returns dense fp32 gradients {tok_table, byte_table, scale_tok,
    scale_byte, weight, bias} -- fp32 also when the tables are bfloat16 (accumulated in fp32; the autograd
    node rounds once to the parameter dtype); pass `into` (same keys, fp32) to accumulate into existing
    buffers such as ``param.grad``.  `token_order`: what :func:`token_order` returned for these tokens and this table height (the
    caller orders streams); without it the call groups the positions itself."""
    m = _MODES[mode]
    if tokens.ndim == 1:
        tokens = tokens[None]
    dev = capi.require_device(grad_out, tokens, tok_table, byte_table, ids_a, ids_b, scale_tok, scale_byte)
    tok = tokens.to(torch.int32) if tokens.dtype != torch.int32 else tokens
    tok = tok if tok.is_contiguous() else tok.contiguous()
    B, T = tok.shape
    dt = tok_table.dtype
    code = capi.dtype_code(dt)
    tt = _contig(tok_table, dt, "tok_table")
    g = _contig(grad_out, dt, "grad_out")
    d = capi.MotEmbedMixDesc()
    d.struct_size = C.sizeof(capi.MotEmbedMixDesc)
    d.dtype = code
    d.flags = _flags(one_launch=one_launch, du_fp32=du_fp32)
    d.n_rows, d.tokens_per_row, d.bpt, d.mode = B, T, int(bpt), m
    d.tokens = capi.ptr(tok)
    d.tok_table, d.tok_rows, d.tok_dim, d.model_dim = capi.ptr(tt), tt.shape[0], tt.shape[1], tt.shape[1]
    into = into or {}
    fwd_out = out
    out = {"tok_table": into.get("tok_table", None), "byte_table": into.get("byte_table", None),
           "scale_tok": into.get("scale_tok", None), "scale_byte": into.get("scale_byte", None)}
    if out["tok_table"] is None:
        out["tok_table"] = torch.zeros_like(tt, dtype=torch.float32)
    keep = [tok, tt, g]
    gr = capi.MotEmbedMixGrads()
    gr.struct_size = C.sizeof(capi.MotEmbedMixGrads)
    gr.grad_out = capi.ptr(g)
    gr.d_tok_table = capi.ptr(out["tok_table"])
    if m != capi.MIX_NOOP:
        bt = _contig(byte_table, dt, "byte_table")
        ia = _contig(ids_a, torch.int64, "ids_a")
        keep += [bt, ia]
        d.byte_table, d.byte_rows, d.byte_dim = capi.ptr(bt), bt.shape[0], bt.shape[1]
        d.id_source, d.ids_a = capi.IDS_GIVEN, capi.ptr(ia)
        if ids_b is not None:
            ib = _contig(ids_b, torch.int64, "ids_b")
            keep.append(ib)
            d.ids_b = capi.ptr(ib)
        if out["byte_table"] is None:
            out["byte_table"] = torch.zeros_like(bt, dtype=torch.float32)
        gr.d_byte_table = capi.ptr(out["byte_table"])
    if m == capi.MIX_CONCAT_LINEAR:
        w = _contig(weight, dt, "weight")
        keep.append(w)
        d.weight, d.model_dim, d.bytes_first = capi.ptr(w), w.shape[0], int(bytes_first)
        out["weight"] = into.get("weight") if into.get("weight") is not None else torch.zeros_like(w, dtype=torch.float32)
        gr.d_weight = capi.ptr(out["weight"])
        if bias is not None:
            bs = _contig(bias, dt, "bias")
            keep.append(bs)
            d.bias = capi.ptr(bs)
            out["bias"] = into.get("bias") if into.get("bias") is not None else torch.zeros_like(bs, dtype=torch.float32)
            gr.d_bias = capi.ptr(out["bias"])
        if norm_out:
            if fwd_out is None or row_rnorm is None:
                raise ValueError("concat_linear backward with norm_out needs the forward's output and row_rnorm")
            xo = _contig(fwd_out, dt, "out")
            keep.append(xo)
            d.out, d.out_row_rnorm = capi.ptr(xo), capi.ptr(row_rnorm)
    d.norm_tok, d.norm_byte, d.norm_out = int(norm_tok), int(norm_byte), int(norm_out)
    d.eps = float(eps or 0.0)
    d.scale_tok, d.scale_byte = capi.ptr(scale_tok), capi.ptr(scale_byte)
    for k, sc in (("scale_tok", scale_tok), ("scale_byte", scale_byte)):
        if sc is not None and out[k] is None:
            out[k] = torch.zeros(1, dtype=torch.float32, device=dev)
    gr.d_scale_tok, gr.d_scale_byte = capi.ptr(out["scale_tok"]), capi.ptr(out["scale_byte"])
    if token_order is not None:
        need = capi.lib.mot_token_order_ints(B * T, tt.shape[0])
        if token_order.dtype != torch.int32 or token_order.numel() != need or token_order.device != dev or not token_order.is_contiguous():
            raise ValueError(f"token_order must be the contiguous int32[{need}] tensor token_order(tokens, {tt.shape[0]}) returned")
        gr.token_order = capi.ptr(token_order)
        keep.append(token_order)
    d.status = capi.ptr(capi.status_word(dev))
    ws = _workspace(dev, capi.lib.mot_embed_mix_bwd_workspace_bytes(C.byref(d)))
    if ws is not None:
        d.workspace, d.workspace_bytes = capi.ptr(ws), ws.numel()
    capi.check(capi.lib.mot_embed_mix_bwd(C.byref(d), C.byref(gr), capi.stream_of(dev)))
    capi.after_call(dev)
    return out


def embed_mix(tokens: torch.Tensor, tok_table: torch.Tensor, byte_table: torch.Tensor | None = None, *,
              scale_tok: torch.Tensor | None = None, scale_byte: torch.Tensor | None = None, **kw):
    """The fused front-end (see `_embed_mix_fwd` for the arguments).  With autograd enabled and
    differentiable parameters it records one backward node: modes "sum", "noop", "concat_linear" with
    float32 or bfloat16 tables; "mean" without an output norm.  Anything else raises here,
    at forward time, rather than in backward()."""
    params = (tok_table, byte_table, scale_tok, scale_byte, kw.get("weight"), kw.get("bias"))
    if torch.is_grad_enabled() and any(p is not None and p.requires_grad for p in params):
        if kw["mode"] not in _BWD_MODES:
            raise RuntimeError(
                f"mixture-of-tokenizers_amd: backward of mode '{kw['mode']}' is not built yet (forward only); "
                "call it under torch.no_grad() or with frozen parameters")
        if kw["mode"] == "mean" and kw.get("norm_out"):
            raise RuntimeError(
                "mixture-of-tokenizers_amd: the backward of the MEAN mix is built without an output norm (the reference's residual, "
                "inference.py:267, has none; mot_embed_mix_bwd, include/mot.h); call it under torch.no_grad() or with frozen parameters")
        if kw.get("out") is not None or kw.get("counters") is not None:
            raise ValueError("out= / counters= cannot be combined with autograd")
        r = _EmbedMixFn.apply(tok_table, byte_table, scale_tok, scale_byte, kw.get("weight"), kw.get("bias"), tokens, kw)
        if kw.get("return_ids"):
            return MixResult(*r)
        return r
    return _embed_mix_fwd(tokens, tok_table, byte_table, scale_tok=scale_tok, scale_byte=scale_byte, **kw)


def _cross_attn_desc(tokens, ids_a, ids_b, tok_table, byte_table, q_w, kv_w, proj_w, lambda_factor, cos_q, sin_q, cos_k, sin_k,
                     bpt, n_heads, norm_tok, norm_byte, head_layout, eps, matmul=None, widened=None, wide_cache=None):
    """Validated MotCrossAttnDesc for both directions; returns (desc, keepalive list, device, T, D).  `widened`: the fp32 copies
    (tables, weights, lambda) an earlier call of the same autograd node made of the same bf16 operands -- keep[1:7] -- reused
    instead of made again."""
    dev = capi.require_device(tokens, ids_a, ids_b, tok_table, byte_table, q_w, kv_w, proj_w, lambda_factor, cos_q, sin_q, cos_k, sin_k)
    T = tokens.shape[1]
    tok = tokens.to(torch.int32) if tokens.dtype != torch.int32 else tokens
    tok = tok if tok.is_contiguous() else tok.contiguous()
    f32 = torch.float32
    # bf16 tables (the production cast, train_gpt.py:1124-1126): the attention kernels of this mixin are fp32, so the operands are
    # widened once per call -- the tables as they are (bf16 values), the fp32 master weights rounded to bf16 first, as
    # `self.q_w.type_as(x)` does (lines 277-278, 185-186) -- and the result is rounded once to bf16 by the caller.  The products
    # over the tokens (q, c_proj and their four backward products) then run on the bf16 MFMA (`matmul_dtype`, include/mot.h):
    # their row operands are bf16 tensors in the reference too.  `matmul="fp32"` keeps them on the fp32 MFMA.
    bf = tok_table.dtype == torch.bfloat16
    if matmul not in (None, "fp32", "bf16"):
        raise ValueError(f"cross_attn: matmul must be None, 'fp32' or 'bf16' (got {matmul!r})")
    mm_bf16 = (bf and tok_table.shape[1] % 8 == 0) if matmul is None else matmul == "bf16"
    if byte_table.dtype != tok_table.dtype:
        raise TypeError(f"cross_attn: byte table is {byte_table.dtype} but the token table is {tok_table.dtype}")
    wide = (lambda t: t.detach().float()) if bf else (lambda t: t.detach())
    as_used = (lambda w: w.detach().to(torch.bfloat16).float()) if bf else (lambda w: w.detach())
    if bf and wide_cache is not None:   # no-grad calls with a caller-kept cache: the fp32 copies live beside the K / V tables
        def _kept(fn):
            def get(t):
                key = (t.data_ptr(), t._version, tuple(t.shape), str(t.dtype))
                hit = wide_cache.get(id(t))
                if hit is None or hit[0] != key:
                    hit = wide_cache[id(t)] = (key, fn(t), t)
                return hit[1]
            return get
        wide, as_used = _kept(wide), _kept(as_used)
    if widened is not None:   # (the backward of a bf16 step: the forward's copies; autograd has checked that the originals are unchanged)
        wide = as_used = None
        tt, bt = widened[0], widened[1]
    else:
        tt, bt = _contig(wide(tok_table), f32, "tok_table"), _contig(wide(byte_table), f32, "byte_table")
    D = tt.shape[1]
    if bt.shape[1] != D:
        raise AssertionError("cross_attn: byte_dim == token_dim == model_dim (train_gpt.py:449)")
    HD = n_heads * 128
    if widened is not None:
        qw, kvw, pw = widened[2], widened[3], widened[4]
    else:
        qw, kvw, pw = _contig(as_used(q_w), f32, "q_w"), _contig(as_used(kv_w), f32, "kv_w"), _contig(as_used(proj_w), f32, "proj_w")
    if qw.shape != (HD, D) or kvw.shape != (2, HD, D) or pw.shape != (D, HD):
        raise AssertionError(f"cross_attn: weights {tuple(qw.shape)}, {tuple(kvw.shape)}, {tuple(pw.shape)} do not fit heads={n_heads}, dim={D}")
    lam = widened[5] if widened is not None else _contig(as_used(lambda_factor).reshape(1), f32, "lambda_factor")
    ia = _contig(ids_a.reshape(-1), torch.int64, "ids_a")
    ib = None if ids_b is None else _contig(ids_b.reshape(-1), torch.int64, "ids_b")
    if ia.numel() != T * bpt or (ib is not None and ib.numel() != T * bpt):
        raise AssertionError(f"cross_attn: byte ids must hold T*bpt = {T * bpt} entries")
    rot = [_contig(t, f32, "rotary buffer") for t in (cos_q, sin_q, cos_k, sin_k)]
    if any(r.ndim != 2 or r.shape[1] != 64 for r in rot):
        raise AssertionError("cross_attn: rotary buffers must be (len, 64)")
    d = capi.MotCrossAttnDesc()
    d.struct_size = C.sizeof(capi.MotCrossAttnDesc)
    d.dtype, d.n_tokens, d.bpt, d.n_heads, d.dim = capi.F32, T, int(bpt), int(n_heads), D
    d.matmul_dtype = capi.BF16 if mm_bf16 else capi.F32
    d.io_dtype = capi.BF16 if (bf and mm_bf16) else capi.F32   # bf16 tables: the result and its gradient cross the boundary in bf16
    tt16 = None
    if bf and mm_bf16:   # the bf16 table itself: the normalised token rows are then gathered in bf16 directly
        tt16 = _contig(tok_table.detach(), torch.bfloat16, "tok_table")
        d.tok_table_bf16 = capi.ptr(tt16)
    d.head_layout = {"as_viewed": capi.HEADS_AS_VIEWED, "per_token": capi.HEADS_PER_TOKEN}[head_layout]
    d.tokens, d.ids_a, d.ids_b = capi.ptr(tok), capi.ptr(ia), capi.ptr(ib)
    d.tok_table, d.tok_rows, d.byte_table, d.byte_rows = capi.ptr(tt), tt.shape[0], capi.ptr(bt), bt.shape[0]
    d.norm_tok, d.norm_byte = int(norm_tok), int(norm_byte)
    d.q_w, d.kv_w, d.proj_w, d.lambda_factor = capi.ptr(qw), capi.ptr(kvw), capi.ptr(pw), capi.ptr(lam)
    d.cos_q, d.sin_q, d.cos_k, d.sin_k = (capi.ptr(r) for r in rot)
    d.rot_q_len, d.rot_k_len = rot[0].shape[0], rot[2].shape[0]
    # norm() is F.rms_norm(x, eps=None): eps = finfo(x.dtype).eps, and with bf16 tables every tensor it is applied to here (the
    # embeddings, q, k) is bf16 in the reference -- 2^-7, not the float32 epsilon of the widened copies (train_gpt.py:172-173)
    d.eps = float(eps or (2.0 ** -7 if bf else 0.0))
    d.status = capi.ptr(capi.status_word(dev))
    return d, [tok, tt, bt, qw, kvw, pw, lam, ia, ib] + rot + [tt16], dev, T, D


class _CrossAttnFn(torch.autograd.Function):
    """One autograd node for the cross-attention mixin: forward = mot_cross_attn_fwd, backward = mot_cross_attn_bwd
    (everything is recomputed from the inputs; dense fp32 gradients for the two tables, q_w, kv_w, c_proj and lambda)."""

    @staticmethod
    def forward(ctx, tok_table, byte_table, q_w, kv_w, proj_w, lambda_factor, tokens, ids_a, ids_b, rot, kw):
        # the projected queries and the attention output are kept for the backward (2 x T x hdim floats)
        saved = torch.empty(2 * tokens.shape[-1] * kw["n_heads"] * 128, dtype=torch.float32, device=tok_table.device)
        ctx.save_for_backward(tok_table, byte_table, q_w, kv_w, proj_w, lambda_factor, tokens, ids_a, saved, *rot)
        ctx.ids_b = ids_b          # (an integer tensor or None: nothing autograd tracks)
        ctx.kw = kw
        # bf16 tables: the fp32 copies of the operands this call makes serve the backward too (six conversions less per step)
        ctx.widened = [] if tok_table.dtype == torch.bfloat16 else None
        x = _cross_attn_fwd(tokens, ids_a, ids_b, tok_table, byte_table, q_w, kv_w, proj_w, lambda_factor, *rot, saved_qy=saved, _keep_widened=ctx.widened, **kw)
        return x.to(tok_table.dtype)

    @staticmethod
    def backward(ctx, gx):
        tok_table, byte_table, q_w, kv_w, proj_w, lambda_factor, tokens, ids_a, saved, cq, sq, ck, sk = ctx.saved_tensors
        g = cross_attn_backward(gx, tokens, ids_a, tok_table, byte_table, q_w=q_w, kv_w=kv_w, proj_w=proj_w, lambda_factor=lambda_factor,
                                cos_q=cq, sin_q=sq, cos_k=ck, sin_k=sk, saved_qy=saved, ids_b=ctx.ids_b, widened=ctx.widened or None, **ctx.kw)
        # fp32 sums; the tables' gradients are rounded once to the tables' dtype (bf16 in production), the weights stay fp32 masters
        return (g["tok_table"].to(tok_table.dtype), g["byte_table"].to(byte_table.dtype), g["q_w"], g["kv_w"], g["proj_w"],
                g["lambda_factor"].reshape(lambda_factor.shape).to(lambda_factor.dtype), None, None, None, None, None)


@torch.compiler.disable
def cross_attn_backward(grad_out, tokens, ids_a, tok_table, byte_table, *, q_w, kv_w, proj_w, lambda_factor, cos_q, sin_q, cos_k, sin_k,
                        bpt, n_heads, norm_tok=True, norm_byte=True, head_layout="as_viewed", eps=None, saved_qy=None, ids_b=None,
                        matmul=None, widened=None) -> dict:
    """One call of mot_cross_attn_bwd: dense fp32 gradients {tok_table, byte_table, q_w, kv_w, proj_w, lambda_factor}.
    `saved_qy`: the buffer the forward filled (projected queries + attention output); without it they are recomputed.
    `ids_b`: the second id tensor of the add_padded_and_pulled embedding (train_gpt.py:364-372)."""
    if tokens.ndim == 1:
        tokens = tokens[None]
    d, keep, dev, T, D = _cross_attn_desc(tokens, ids_a, ids_b, tok_table, byte_table, q_w, kv_w, proj_w, lambda_factor, cos_q, sin_q, cos_k, sin_k,
                                          bpt, n_heads, norm_tok, norm_byte, head_layout, eps, matmul, widened)
    io = torch.bfloat16 if d.io_dtype == capi.BF16 else torch.float32
    g = _contig(grad_out.reshape(T, D).to(io), io, "grad_out")
    out = {"tok_table": torch.zeros_like(keep[1]), "byte_table": torch.zeros_like(keep[2]), "q_w": torch.zeros_like(keep[3]),
           "kv_w": torch.zeros_like(keep[4]), "proj_w": torch.zeros_like(keep[5]), "lambda_factor": torch.zeros(1, dtype=torch.float32, device=dev)}
    gr = capi.MotCrossAttnGrads()
    gr.struct_size = C.sizeof(capi.MotCrossAttnGrads)
    gr.grad_out = capi.ptr(g)
    gr.d_tok_table, gr.d_byte_table = capi.ptr(out["tok_table"]), capi.ptr(out["byte_table"])
    gr.d_q_w, gr.d_kv_w, gr.d_proj_w, gr.d_lambda = capi.ptr(out["q_w"]), capi.ptr(out["kv_w"]), capi.ptr(out["proj_w"]), capi.ptr(out["lambda_factor"])
    if saved_qy is not None:
        d.saved_qy = capi.ptr(_contig(saved_qy, torch.float32, "saved_qy"))
    ws = _workspace(dev, capi.lib.mot_cross_attn_bwd_workspace_bytes(C.byref(d)))
    if ws is not None:
        d.workspace, d.workspace_bytes = capi.ptr(ws), ws.numel()
    capi.check(capi.lib.mot_cross_attn_bwd(C.byref(d), C.byref(gr), capi.stream_of(dev)))
    capi.after_call(dev)
    return out


def _cross_attn_fwd(tokens, ids_a, ids_b, tok_table, byte_table, q_w, kv_w, proj_w, lambda_factor, cos_q, sin_q, cos_k, sin_k, *,
                    bpt, n_heads, norm_tok=True, norm_byte=True, head_layout="as_viewed", eps=None, kv_cache=None, saved_qy=None, matmul=None,
                    _keep_widened=None):
    d, keep, dev, T, D = _cross_attn_desc(tokens, ids_a, ids_b, tok_table, byte_table, q_w, kv_w, proj_w, lambda_factor, cos_q, sin_q, cos_k, sin_k,
                                          bpt, n_heads, norm_tok, norm_byte, head_layout, eps, matmul,
                                          wide_cache=None if kv_cache is None else kv_cache.setdefault("widened", {}))
    if _keep_widened is not None:
        _keep_widened.extend(keep[1:7])   # tt, bt, qw, kvw, pw, lam
    out = torch.empty((1, T, D), dtype=torch.bfloat16 if d.io_dtype == capi.BF16 else torch.float32, device=dev)
    d.out = capi.ptr(out)
    if saved_qy is not None:
        d.saved_qy = capi.ptr(saved_qy)
    if kv_cache is not None and ids_b is None:
        # the per-byte-row key / value tables depend on (byte_table, kv_w, lambda_factor) only: kept across calls while those are unchanged
        key = tuple((t.data_ptr(), t._version) for t in (byte_table, kv_w, lambda_factor)) + (bool(norm_byte), float(eps or 0.0), str(dev))
        n = 2 * keep[2].shape[0] * n_heads * 128
        if kv_cache.get("buf") is None or kv_cache["buf"].numel() != n or kv_cache["buf"].device != dev:
            kv_cache["buf"], kv_cache["key"] = torch.empty(n, dtype=torch.float32, device=dev), None
        d.kv_tables, d.kv_tables_ready = capi.ptr(kv_cache["buf"]), int(kv_cache.get("key") == key)
        kv_cache["key"] = key
    ws = _workspace(dev, capi.lib.mot_cross_attn_workspace_bytes(C.byref(d)))
    if ws is not None:
        d.workspace, d.workspace_bytes = capi.ptr(ws), ws.numel()
    capi.check(capi.lib.mot_cross_attn_fwd(C.byref(d), capi.stream_of(dev)))
    capi.after_call(dev)
    return out


@torch.compiler.disable
def cross_attn(tokens: torch.Tensor, ids_a: torch.Tensor, tok_table: torch.Tensor, byte_table: torch.Tensor, *,
               q_w: torch.Tensor, kv_w: torch.Tensor, proj_w: torch.Tensor, lambda_factor: torch.Tensor,
               cos_q: torch.Tensor, sin_q: torch.Tensor, cos_k: torch.Tensor, sin_k: torch.Tensor,
               bpt: int, n_heads: int, ids_b: torch.Tensor | None = None, norm_tok: bool = True, norm_byte: bool = True,
               head_layout: str = "as_viewed", eps: float | None = None, kv_cache: dict | None = None,
               matmul: str | None = None) -> torch.Tensor:
    """The cross-attention byte mixin on top of the two embedding gathers (train_gpt.py:342-379, 446-464, 271-300):
    tokens (1, T) -> (1, T, dim).  The reference asserts batch 1 (line 275).  fp32.  With autograd enabled and
    differentiable parameters it records one backward node (either embedding: one id tensor, or norm(emb(padded) + emb(pulled))).
    head_layout "as_viewed" reproduces the reference's reshape of k and v (lines 283-284); "per_token" is the
    rearrange its comment names.  bfloat16 tables are accepted (operands widened once per call, bf16 result and table gradients):
    the attention itself stays fp32, the products over the tokens -- q, c_proj and their backward products -- then run on the
    bf16 MFMA with fp32 accumulation, their row operands rounded to bf16 where the reference's are bf16 tensors
    (`matmul="fp32"` keeps them on the fp32 MFMA; `matmul="bf16"` asks for the bf16 MFMA with fp32 tables too)."""
    if tokens.ndim == 1:
        tokens = tokens[None]
    assert tokens.shape[0] == 1, "Must use batch size = 1 for FlexAttention"      # train_gpt.py:275
    kw = dict(bpt=bpt, n_heads=n_heads, norm_tok=norm_tok, norm_byte=norm_byte, head_layout=head_layout, eps=eps, matmul=matmul)
    params = (tok_table, byte_table, q_w, kv_w, proj_w, lambda_factor)
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        return _CrossAttnFn.apply(tok_table, byte_table, q_w, kv_w, proj_w, lambda_factor, tokens, ids_a, ids_b, (cos_q, sin_q, cos_k, sin_k), kw)
    # `kv_cache` (a dict the caller keeps, e.g. on the module): inference calls reuse the per-byte-row K/V tables while the
    # byte table, kv_w and lambda_factor are unchanged (tensor versions are checked)
    return _cross_attn_fwd(tokens, ids_a, ids_b, tok_table, byte_table, q_w, kv_w, proj_w, lambda_factor, cos_q, sin_q, cos_k, sin_k,
                           kv_cache=kv_cache, **kw).to(tok_table.dtype)


_SWA_VERSIONS = {"no_residual": capi.SWA_NO_RESIDUAL, "one_residual": capi.SWA_ONE_RESIDUAL, "two_residual": capi.SWA_TWO_RESIDUAL}


@torch.compiler.disable
def char_swa(tokens: torch.Tensor, char_ids: torch.Tensor, tok_table: torch.Tensor, char_table: torch.Tensor, *,
             attn_norm_w: torch.Tensor, char_norm_w: torch.Tensor, wq: torch.Tensor, wk: torch.Tensor, wv: torch.Tensor, wo: torch.Tensor,
             n_heads: int, head_dim: int, window: int = 8, norm_eps: float = 1e-5, version: str = "two_residual",
             lambda_tok: torch.Tensor | None = None, lambda_char: torch.Tensor | None = None, matmul: str | None = None,
             kv_cache: dict | None = None) -> torch.Tensor:
    """The Llama character mixer up to the feed-forward (inference.py:146-224 + 260-267 on the gathers of 323-327):
    tokens (B, T) int, char_ids (B, T, c_v) int64 -> h (B, T, dim) fp32.  bf16 tables: bf16 result; operands widened once, the
    attention in fp32, the two products over the tokens (wq, wo) on the bf16 MFMA with their row operands rounded to bf16
    (`matmul="fp32"` keeps them on the fp32 MFMA, `matmul="bf16"` asks for the bf16 MFMA with fp32 tables).  Forward only (the
    file is the reference's inference path); see mot_char_swa_fwd in include/mot.h.  `kv_cache` (a dict the caller keeps, e.g. on
    the module): the per-character key / value tables are reused across calls while char_table, char_norm_w, wk and wv are unchanged
    (storage and version are checked)."""
    if matmul not in (None, "fp32", "bf16"):
        raise ValueError(f"char_swa: matmul must be None, 'fp32' or 'bf16' (got {matmul!r})")
    if tokens.ndim == 1:
        tokens, char_ids = tokens[None], char_ids[None]
    if char_ids.ndim != 3 or char_ids.shape[:2] != tokens.shape:
        raise ValueError(f"char_ids must be (B, T, c_v) matching tokens {tuple(tokens.shape)}, got {tuple(char_ids.shape)}")
    kv_key = tuple((t.data_ptr(), t._version) for t in (char_table, char_norm_w, wk, wv)) + (float(norm_eps), str(char_table.dtype))
    params = (tok_table, char_table, attn_norm_w, char_norm_w, wq, wk, wv, wo, lambda_tok, lambda_char)
    if torch.is_grad_enabled() and any(p is not None and p.requires_grad for p in params):
        raise RuntimeError("mixture-of-tokenizers_amd: the character mixer (inference/inference.py) is built forward-only; call it under "
                           "torch.no_grad() or with frozen parameters")
    dev = capi.require_device(tokens, char_ids, *params)
    f32 = torch.float32
    tok = _contig(tokens.to(torch.int32), torch.int32, "tokens")
    cid = _contig(char_ids, torch.int64, "char_ids")
    B, T = tok.shape
    # bfloat16 tables / weights (round 3; the reference script itself runs in the default float32): every operand is widened once
    # per call -- the values a bf16 model holds -- the arithmetic is the fp32 kernels', and the result is rounded once to bf16
    bf = tok_table.dtype == torch.bfloat16
    if bf:
        if char_table.dtype != torch.bfloat16:
            raise TypeError(f"char_swa: char_table is {char_table.dtype} but the token table is bfloat16")
        # (with `kv_cache` the widened copies are kept beside the tables while the originals are unchanged: the file is an inference
        #  path, and the Llama token table alone is 0.5 GB to read and 1 GB to write per call otherwise)
        wide_cache = None if kv_cache is None else kv_cache.setdefault("widened", {})

        def widen(t):
            if t is None:
                return None
            if wide_cache is None:
                return t.detach().to(torch.bfloat16).float()
            key = (t.data_ptr(), t._version, tuple(t.shape), str(t.dtype))
            hit = wide_cache.get(id(t))
            if hit is None or hit[0] != key:
                hit = wide_cache[id(t)] = (key, t.detach().to(torch.bfloat16).float(), t)   # (t itself: the id stays its own)
            return hit[1]
        tok_table, char_table, attn_norm_w, char_norm_w, wq, wk, wv, wo, lambda_tok, lambda_char = (
            widen(t) for t in (tok_table, char_table, attn_norm_w, char_norm_w, wq, wk, wv, wo, lambda_tok, lambda_char))
    tt, ct = _contig(tok_table.detach(), f32, "tok_table"), _contig(char_table.detach(), f32, "char_table")
    D, hdim = tt.shape[1], n_heads * head_dim
    if ct.shape[1] != D:
        raise ValueError("char_table and tok_table must have the same number of columns (hidden_size)")
    ws_ = [_contig(w.detach(), f32, n) for w, n in ((attn_norm_w, "attn_norm_w"), (char_norm_w, "char_norm_w"), (wq, "wq"), (wk, "wk"), (wv, "wv"), (wo, "wo"))]
    if ws_[0].shape != (D,) or ws_[1].shape != (D,) or any(w.shape != (hdim, D) for w in ws_[2:5]) or ws_[5].shape != (D, hdim):
        raise ValueError(f"char_swa: weight shapes do not fit dim {D}, heads {n_heads} x {head_dim}")
    lams = [None if l is None else _contig(l.detach().reshape(1), f32, "lambda") for l in (lambda_tok, lambda_char)]
    d = capi.MotCharSwaDesc()
    d.struct_size = C.sizeof(capi.MotCharSwaDesc)
    d.dtype, d.n_rows, d.tokens_per_row = capi.F32, B, T
    d.c_v, d.window, d.n_heads, d.head_dim, d.dim = cid.shape[2], int(window), int(n_heads), int(head_dim), D
    d.version = _SWA_VERSIONS[version]
    mm_bf16 = (bf and D % 8 == 0 and hdim % 8 == 0) if matmul is None else matmul == "bf16"
    d.matmul_dtype = capi.BF16 if mm_bf16 else capi.F32
    d.tokens, d.char_ids = capi.ptr(tok), capi.ptr(cid)
    d.tok_table, d.tok_rows, d.char_table, d.char_rows = capi.ptr(tt), tt.shape[0], capi.ptr(ct), ct.shape[0]
    d.norm_eps = float(norm_eps)
    d.attn_norm_w, d.char_norm_w, d.wq, d.wk, d.wv, d.wo = (capi.ptr(w) for w in ws_)
    d.lambda_tok, d.lambda_char = capi.ptr(lams[0]), capi.ptr(lams[1])
    d.io_dtype = capi.BF16 if (bf and mm_bf16) else capi.F32   # bf16 tables: the last product writes the bf16 result itself
    out = torch.empty((B, T, D), dtype=torch.bfloat16 if d.io_dtype == capi.BF16 else f32, device=dev)
    d.out = capi.ptr(out)
    d.status = capi.ptr(capi.status_word(dev))
    if kv_cache is not None:
        n = 2 * ct.shape[0] * hdim
        if kv_cache.get("buf") is None or kv_cache["buf"].numel() != n or kv_cache["buf"].device != dev:
            kv_cache["buf"], kv_cache["key"] = torch.empty(n, dtype=f32, device=dev), None
        d.kv_tables, d.kv_tables_ready = capi.ptr(kv_cache["buf"]), int(kv_cache.get("key") == kv_key + (str(dev),))
        kv_cache["key"] = kv_key + (str(dev),)
    ws = _workspace(dev, capi.lib.mot_char_swa_workspace_bytes(C.byref(d)))
    if ws is not None:
        d.workspace, d.workspace_bytes = capi.ptr(ws), ws.numel()
    capi.check(capi.lib.mot_char_swa_fwd(C.byref(d), capi.stream_of(dev)))
    capi.after_call(dev)
    return out.to(torch.bfloat16) if bf else out
