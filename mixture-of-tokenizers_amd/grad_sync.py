"""Gradient exchange of the front-end's parameters across data-parallel ranks.

The reference averages gradients with one `dist.all_reduce(param.grad, op=AVG)` per parameter
(scaled-pre-train/train_gpt.py:1320-1321).  The front-end owns one large table (token embedding,
154 MB fp32 at GPT-2 vocab x 768) and a few tiny tensors (byte table 88 KB, learned scalars, the
mixin weight); over xGMI a ring all-reduce is per-link bound, so the tiny tensors are pure launch
latency.  GradBucket backs every `.grad` with a view into ONE flat buffer: the backward kernels
accumulate straight into it (`mot_embed_mix_bwd` only ever adds into its outputs) and the whole
front-end is exchanged in one collective on RCCL ("nccl" backend) -- or gloo in the CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradBucket:
    """One flat gradient buffer behind the `.grad` of every given parameter (views), one all-reduce for all of them.

    `in_place=True` is an explicit opt-in to the kernel-level accumulate: the fused backward (functional._EmbedMixFn) then adds
    its fp32 sums straight into these views and hands autograd NO gradient for the parameters -- no table-sized temporary, no
    AccumulateGrad pass -- which also means that tensor hooks / post-accumulate-grad hooks registered on them do not fire and
    that `.grad` changes as a side effect of the backward call itself (do not combine with torch.autograd.grad).  With the default
    `in_place=False` the views still back `.grad` (autograd accumulates into them as into any existing .grad) and everything
    autograd promises holds."""

    def __init__(self, params, process_group=None, in_place: bool = False):
        seen, self.params = set(), []
        for p in params:                       # tied weights (mathblations/model.py:316-317) appear once
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        if not self.params:
            raise ValueError("GradBucket: no trainable parameters")
        dt, dev = self.params[0].dtype, self.params[0].device
        if any(p.dtype != dt or p.device != dev for p in self.params):
            raise ValueError("GradBucket: parameters must share one dtype and device")
        # 256-byte aligned slots so every view can be handed to the kernels' 16-byte vector stores
        al = 256 // self.params[0].element_size()
        offs, o = [], 0
        for p in self.params:
            offs.append(o)
            o += (p.numel() + al - 1) // al * al
        self.flat = torch.zeros(o, dtype=dt, device=dev)
        self.group = process_group
        for p, off in zip(self.params, offs):
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            # fp32 views: the fused backward adds into them directly (functional._EmbedMixFn.backward) instead of
            # materialising a table-sized temporary for autograd's AccumulateGrad to add
            p._mot_grad_in_place = bool(in_place) and dt == torch.float32

    def zero_(self) -> None:
        """Replaces optimizer.zero_grad(set_to_none=True) for these parameters (keeps the views)."""
        self.flat.zero_()

    def all_reduce(self, async_op: bool = False):
        """Average over the ranks of the group.  Returns the Work handle when async_op."""
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return None
        if dist.get_backend(self.group) == "nccl":           # RCCL has AVG; one kernel
            return dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
        self.flat.mul_(1.0 / dist.get_world_size(self.group))
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
