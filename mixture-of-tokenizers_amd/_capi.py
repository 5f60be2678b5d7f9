"""ctypes binding of libmot_hip.so (include/mot.h) -- the only place the C ABI is called from.

The library is hand-written HIP for gfx950; there is NO CPU or PyTorch fallback in this
package: if the shared object is missing, importing this module raises, and every wrapper
refuses tensors that are not on a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch

_HERE = Path(__file__).resolve().parent
# The shipped binding loads the in-tree library.  Kernel-variant A/B runs (tools/variants.sh) may point MOT_DEV_LIB at another
# build, but only together with the explicit dev switch MOT_DEV=1; build_info() and bench.py's JSON name the file that was loaded.
_dev_lib = os.environ.get("MOT_DEV_LIB")
if _dev_lib and os.environ.get("MOT_DEV") != "1":
    import sys as _sys
    print(f"mixture-of-tokenizers_amd: MOT_DEV_LIB={_dev_lib} ignored (set MOT_DEV=1 to load a dev build)", file=_sys.stderr)
    _dev_lib = None
LIB_PATH = Path(_dev_lib or (_HERE / "libmot_hip.so"))
IS_DEV_LIB = _dev_lib is not None

# ---- enums of include/mot.h
MOT_OK, MOT_EINVAL, MOT_ESHAPE, MOT_EUNSUPPORTED, MOT_EHIP, MOT_EWORKSPACE = 0, -1, -2, -3, -4, -5
STATUS_TOKEN_OOR, STATUS_BYTE_OOR = 1, 2
PULL_NONE, PULL_LEFT, PULL_RIGHT = 0, 1, 2
MIX_NOOP, MIX_SUM, MIX_MEAN, MIX_CONCAT_LINEAR = 0, 1, 2, 3
IDS_NONE, IDS_FROM_TTB, IDS_GIVEN = 0, 1, 2
F32, BF16 = 0, 1
MAX_BPT = 64
ABI_VERSION = 12
FLAG_LINEAR_ONE_LAUNCH, FLAG_MEAN_GENERIC, FLAG_BWD_DU_FP32, FLAG_LINEAR_COMPOSED = 1, 2, 4, 8
HEADS_AS_VIEWED, HEADS_PER_TOKEN = 0, 1


def dtype_code(t: torch.dtype) -> int:
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise TypeError(f"tables must be float32 or bfloat16, got {t}")

EXPORTS = (
    "mot_version", "mot_last_error", "mot_build_info", "mot_tokens_to_bytes", "mot_pull_bytes",
    "mot_create_batch", "mot_char_matrix", "mot_gather_rows", "mot_embed_mix_desc_size", "mot_embed_mix_workspace_bytes",
    "mot_embed_mix_fwd", "mot_embed_mix_bwd_workspace_bytes", "mot_embed_mix_bwd", "mot_token_order_ints", "mot_token_order",
    "mot_cross_attn_desc_size", "mot_cross_attn_workspace_bytes", "mot_cross_attn_fwd",
    "mot_cross_attn_bwd_workspace_bytes", "mot_cross_attn_bwd",
    "mot_char_swa_desc_size", "mot_char_swa_workspace_bytes", "mot_char_swa_fwd",
)
SWA_NO_RESIDUAL, SWA_ONE_RESIDUAL, SWA_TWO_RESIDUAL = 0, 1, 2


class MotEmbedMixDesc(C.Structure):
    """Field-for-field mirror of struct MotEmbedMixDesc (include/mot.h)."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("dtype", C.c_int32), ("flags", C.c_uint32), ("reserved0", C.c_uint32),
        ("n_rows", C.c_int64), ("tokens_per_row", C.c_int64), ("bpt", C.c_int32), ("mode", C.c_int32),
        ("tokens", C.c_void_p), ("id_source", C.c_int32), ("pull_dir", C.c_int32),
        ("ttb", C.c_void_p), ("ttb_rows", C.c_int64), ("ttb_elem_bytes", C.c_int32), ("add_padded", C.c_int32),
        ("pad_byte", C.c_int32), ("eot_byte", C.c_int32),
        ("ids_a", C.c_void_p), ("ids_b", C.c_void_p),
        ("tok_table", C.c_void_p), ("tok_rows", C.c_int64), ("tok_dim", C.c_int32), ("byte_dim", C.c_int32),
        ("byte_table", C.c_void_p), ("byte_rows", C.c_int64),
        ("model_dim", C.c_int32), ("bytes_first", C.c_int32), ("weight", C.c_void_p), ("bias", C.c_void_p),
        ("norm_tok", C.c_int32), ("norm_byte", C.c_int32), ("norm_out", C.c_int32), ("eps", C.c_float),
        ("scale_tok", C.c_void_p), ("scale_byte", C.c_void_p),
        ("out", C.c_void_p), ("out_ids_padded", C.c_void_p), ("out_ids_pulled", C.c_void_p),
        ("counters", C.c_void_p), ("status", C.c_void_p), ("out_row_rnorm", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class MotEmbedMixGrads(C.Structure):
    """Mirror of struct MotEmbedMixGrads (include/mot.h)."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("reserved", C.c_uint32), ("grad_out", C.c_void_p),
        ("d_tok_table", C.c_void_p), ("d_byte_table", C.c_void_p), ("d_weight", C.c_void_p), ("d_bias", C.c_void_p),
        ("d_scale_tok", C.c_void_p), ("d_scale_byte", C.c_void_p), ("token_order", C.c_void_p),
    ]


class MotCrossAttnDesc(C.Structure):
    """Mirror of struct MotCrossAttnDesc (include/mot.h)."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("dtype", C.c_int32), ("n_tokens", C.c_int64),
        ("bpt", C.c_int32), ("n_heads", C.c_int32), ("head_layout", C.c_int32), ("dim", C.c_int32),
        ("tokens", C.c_void_p), ("ids_a", C.c_void_p), ("ids_b", C.c_void_p),
        ("tok_table", C.c_void_p), ("tok_rows", C.c_int64), ("byte_table", C.c_void_p), ("byte_rows", C.c_int64),
        ("norm_tok", C.c_int32), ("norm_byte", C.c_int32),
        ("q_w", C.c_void_p), ("kv_w", C.c_void_p), ("proj_w", C.c_void_p), ("lambda_factor", C.c_void_p),
        ("cos_q", C.c_void_p), ("sin_q", C.c_void_p), ("cos_k", C.c_void_p), ("sin_k", C.c_void_p),
        ("rot_q_len", C.c_int64), ("rot_k_len", C.c_int64), ("eps", C.c_float), ("kv_tables_ready", C.c_int32),
        ("out", C.c_void_p), ("status", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("kv_tables", C.c_void_p), ("saved_qy", C.c_void_p), ("matmul_dtype", C.c_int32), ("io_dtype", C.c_int32),
        ("tok_table_bf16", C.c_void_p),
    ]


class MotCrossAttnGrads(C.Structure):
    """Mirror of struct MotCrossAttnGrads (include/mot.h)."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("reserved", C.c_uint32), ("grad_out", C.c_void_p),
        ("d_tok_table", C.c_void_p), ("d_byte_table", C.c_void_p), ("d_q_w", C.c_void_p), ("d_kv_w", C.c_void_p),
        ("d_proj_w", C.c_void_p), ("d_lambda", C.c_void_p),
    ]


class MotCharSwaDesc(C.Structure):
    """Mirror of struct MotCharSwaDesc (include/mot.h)."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("dtype", C.c_int32), ("n_rows", C.c_int64), ("tokens_per_row", C.c_int64),
        ("c_v", C.c_int32), ("window", C.c_int32), ("n_heads", C.c_int32), ("head_dim", C.c_int32), ("dim", C.c_int32), ("version", C.c_int32),
        ("tokens", C.c_void_p), ("char_ids", C.c_void_p), ("tok_table", C.c_void_p), ("tok_rows", C.c_int64),
        ("char_table", C.c_void_p), ("char_rows", C.c_int32), ("norm_eps", C.c_float),
        ("attn_norm_w", C.c_void_p), ("char_norm_w", C.c_void_p), ("wq", C.c_void_p), ("wk", C.c_void_p), ("wv", C.c_void_p), ("wo", C.c_void_p),
        ("lambda_tok", C.c_void_p), ("lambda_char", C.c_void_p), ("out", C.c_void_p), ("status", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("matmul_dtype", C.c_int32), ("kv_tables_ready", C.c_int32),
        ("kv_tables", C.c_void_p), ("io_dtype", C.c_int32), ("reserved1", C.c_int32),
    ]


def _load() -> C.CDLL:
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (or `make -C "
            f"{_HERE / 'csrc'}`).  This package has no fallback path.")
    lib = C.CDLL(str(LIB_PATH))
    vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float
    lib.mot_version.restype = C.c_int
    lib.mot_last_error.restype = C.c_char_p
    lib.mot_build_info.restype = C.c_char_p
    lib.mot_tokens_to_bytes.argtypes = [vp, i64, vp, i32, i64, i32, vp, vp, vp]
    lib.mot_pull_bytes.argtypes = [vp, vp, i64, i64, i32, i64, i64, i32, vp]
    lib.mot_create_batch.argtypes = [vp, i64, i64, vp, vp, i32, i64, i32, i64, i64, vp, vp, vp]
    lib.mot_char_matrix.argtypes = [vp, vp, vp, i64, i64, i32, i32, i32, i32, vp, vp]
    lib.mot_gather_rows.argtypes = [vp, vp, i32, i64, vp, i64, i32, i32, f32, vp, vp, vp, i32, vp]
    lib.mot_embed_mix_desc_size.restype = C.c_size_t
    lib.mot_embed_mix_workspace_bytes.restype = C.c_size_t
    lib.mot_embed_mix_workspace_bytes.argtypes = [C.POINTER(MotEmbedMixDesc)]
    lib.mot_embed_mix_fwd.argtypes = [C.POINTER(MotEmbedMixDesc), vp]
    lib.mot_embed_mix_bwd_workspace_bytes.restype = C.c_size_t
    lib.mot_embed_mix_bwd_workspace_bytes.argtypes = [C.POINTER(MotEmbedMixDesc)]
    lib.mot_embed_mix_bwd.argtypes = [C.POINTER(MotEmbedMixDesc), C.POINTER(MotEmbedMixGrads), vp]
    lib.mot_token_order_ints.restype = C.c_size_t
    lib.mot_token_order_ints.argtypes = [i64, i64]
    lib.mot_token_order.argtypes = [vp, i64, i64, vp, vp, vp]
    lib.mot_cross_attn_desc_size.restype = C.c_size_t
    lib.mot_cross_attn_workspace_bytes.restype = C.c_size_t
    lib.mot_cross_attn_workspace_bytes.argtypes = [C.POINTER(MotCrossAttnDesc)]
    lib.mot_cross_attn_fwd.argtypes = [C.POINTER(MotCrossAttnDesc), vp]
    lib.mot_cross_attn_fwd.restype = C.c_int
    lib.mot_cross_attn_bwd_workspace_bytes.restype = C.c_size_t
    lib.mot_cross_attn_bwd_workspace_bytes.argtypes = [C.POINTER(MotCrossAttnDesc)]
    lib.mot_cross_attn_bwd.argtypes = [C.POINTER(MotCrossAttnDesc), C.POINTER(MotCrossAttnGrads), vp]
    lib.mot_cross_attn_bwd.restype = C.c_int
    lib.mot_char_swa_desc_size.restype = C.c_size_t
    lib.mot_char_swa_workspace_bytes.restype = C.c_size_t
    lib.mot_char_swa_workspace_bytes.argtypes = [C.POINTER(MotCharSwaDesc)]
    lib.mot_char_swa_fwd.argtypes = [C.POINTER(MotCharSwaDesc), vp]
    lib.mot_char_swa_fwd.restype = C.c_int
    for name in ("mot_tokens_to_bytes", "mot_pull_bytes", "mot_create_batch", "mot_char_matrix", "mot_gather_rows", "mot_embed_mix_fwd",
                 "mot_embed_mix_bwd"):
        getattr(lib, name).restype = C.c_int
    if lib.mot_version() != ABI_VERSION:
        raise ImportError(f"libmot_hip.so ABI {lib.mot_version()} != binding ABI {ABI_VERSION}")
    if lib.mot_embed_mix_desc_size() != C.sizeof(MotEmbedMixDesc):
        raise ImportError("MotEmbedMixDesc layout mismatch between include/mot.h and _capi.py")
    if lib.mot_char_swa_desc_size() != C.sizeof(MotCharSwaDesc):
        raise ImportError("MotCharSwaDesc layout mismatch between include/mot.h and _capi.py")
    if lib.mot_cross_attn_desc_size() != C.sizeof(MotCrossAttnDesc):
        raise ImportError("MotCrossAttnDesc layout mismatch between include/mot.h and _capi.py")
    return lib


lib = _load()

_EXC = {MOT_EINVAL: ValueError, MOT_ESHAPE: AssertionError, MOT_EUNSUPPORTED: NotImplementedError,
        MOT_EHIP: RuntimeError, MOT_EWORKSPACE: RuntimeError}


def check(rc: int) -> None:
    """Map a MotStatus to the exception type the reference would raise at that point."""
    if rc != MOT_OK:
        raise _EXC.get(rc, RuntimeError)(lib.mot_last_error().decode())


def build_info() -> str:
    return lib.mot_build_info().decode() + f" | loaded {LIB_PATH}" + (" (DEV build via MOT_DEV_LIB)" if IS_DEV_LIB else "")


# ----------------------------------------------------------------------------------------------
# device-side status word: out-of-range ids are flagged, not faulted on (kernels clamp to row 0)
# ----------------------------------------------------------------------------------------------
_status_words: dict[int, torch.Tensor] = {}
_debug_ids = os.environ.get("MOT_DEBUG_IDS", "0") not in ("0", "")


def set_debug_ids(on: bool) -> None:
    """When on, every call synchronises and raises IndexError on an out-of-range id, like
    nn.Embedding does on CPU.  Off (default): call check_status() when convenient."""
    global _debug_ids
    _debug_ids = bool(on)


def status_word(device: torch.device) -> torch.Tensor:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    w = _status_words.get(idx)
    if w is None:
        w = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
        _status_words[idx] = w
    return w


def check_status(device=None) -> None:
    """Synchronising check of the status word(s); raises IndexError as nn.Embedding would."""
    for idx, w in list(_status_words.items()):
        if device is not None and torch.device(device).index not in (None, idx):
            continue
        v = int(w.item())
        if v:
            w.zero_()
            what = [n for bit, n in ((STATUS_TOKEN_OOR, "token id"), (STATUS_BYTE_OOR, "byte id")) if v & bit]
            raise IndexError(f"index out of range in self ({' and '.join(what)} out of range)")


def after_call(device: torch.device) -> None:
    if _debug_ids:
        check_status(device)


# ----------------------------------------------------------------------------------------------
# tensor plumbing
# ----------------------------------------------------------------------------------------------
def require_device(*tensors: torch.Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "mixture-of-tokenizers_amd runs on a HIP device only (got a CPU tensor); "
                "there is deliberately no CPU path in this package")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")
    assert dev is not None
    return dev


def ptr(t) -> int | None:
    return None if t is None else t.data_ptr()


def stream_of(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream
