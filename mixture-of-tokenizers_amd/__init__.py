"""mixture-of-tokenizers_amd -- the mixture-of-tokenizers embedding front-end for MI355X (gfx950).

Hand-written HIP kernels behind a C ABI (include/mot.h, csrc/), called through ctypes.  The
Python layer mirrors the reference's interfaces for this path and nothing else:

  data_creation   make_embedding / tokens_to_bytes / pull_from_left / pull_from_right / create_batch
  functional      tensor-level wrappers (embed_mix = the fused gather + mix forward)
  modules         FlexibleEmbedding / ByteMixin* / CastedLinear (scaled-pre-train), DigitMixin* / GPTConfig
                  (mathblations), SumFrontEnd (modded-nanogpt), FusedFrontEnd (tokens -> x in one launch)
  loader          shard reader, rank slice, input/target shift (distributed_data_generator)
  grad_sync       GradBucket: one flat all-reduce for the front-end's gradients (train_gpt.py:1320-1321)

The directory name carries a hyphen; import it as ``mixture_of_tokenizers_amd`` (the loader
shim at the repo root maps that name onto this directory).
"""
from . import _capi
from . import data_creation, functional, grad_sync, loader, modules
from ._capi import build_info, check_status, set_debug_ids
from .functional import create_batch, embed_mix, embed_mix_plan, gather_rows, pull_bytes, tokens_to_bytes

__all__ = [
    "build_info", "check_status", "set_debug_ids", "data_creation", "functional", "grad_sync", "loader", "modules",
    "create_batch", "embed_mix", "embed_mix_plan", "gather_rows", "pull_bytes", "tokens_to_bytes",
]
