"""CPU-side checks of the product: the C-ABI library loads and exports every symbol that
include/mot.h declares, the ctypes descriptor matches the C struct, argument validation maps
to the reference's exception types, and CPU tensors are refused (no fallback path exists)."""
import ctypes as C
import re
from pathlib import Path

import pytest
import torch

REPO = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    import mixture_of_tokenizers_amd as mot
    from mixture_of_tokenizers_amd import _capi
    header = (REPO / "include" / "mot.h").read_text()
    declared = set(re.findall(r"\b(mot_[a-z_]+)\s*\(", header))
    assert declared == set(_capi.EXPORTS), declared ^ set(_capi.EXPORTS)
    lib = C.CDLL(str(_capi.LIB_PATH))
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported"
    assert _capi.lib.mot_version() == _capi.ABI_VERSION
    assert _capi.lib.mot_embed_mix_desc_size() == C.sizeof(_capi.MotEmbedMixDesc)
    assert "gfx950" in mot.build_info()


def test_product_does_not_import_the_oracle():
    pkg = REPO / "mixture-of-tokenizers_amd"
    for py in pkg.rglob("*.py"):
        txt = py.read_text()
        assert "import oracle" not in txt and "from oracle" not in txt, py
    for src in (pkg / "csrc").iterdir():
        if src.suffix in (".hip", ".hpp", ".cpp", ".h") or src.name == "Makefile":
            assert "oracle" not in src.read_text().lower(), src


def test_cpu_tensors_are_refused():
    import mixture_of_tokenizers_amd as mot
    with pytest.raises(RuntimeError, match="HIP device only"):
        mot.pull_bytes(torch.zeros(1, 8, dtype=torch.int64), 8, 456, 457, "left")
    with pytest.raises(RuntimeError, match="HIP device only"):
        mot.tokens_to_bytes(torch.zeros(4, dtype=torch.int32), torch.zeros(8, 8, dtype=torch.int16))


def test_validation_without_a_gpu():
    """Argument checks run before any HIP call, so they are testable on a CPU-only host."""
    from mixture_of_tokenizers_amd import _capi
    lib = _capi.lib
    one = C.c_void_p(64)  # never dereferenced: validation fails first
    assert lib.mot_pull_bytes(one, one, 1, 16, 8, 456, 457, _capi.PULL_LEFT, None) == _capi.MOT_EINVAL  # in == out
    two = C.c_void_p(128)
    assert lib.mot_pull_bytes(one, two, 1, 12, 8, 456, 457, _capi.PULL_LEFT, None) == _capi.MOT_ESHAPE  # T % bpt
    assert b"divisible" in lib.mot_last_error()
    with pytest.raises(AssertionError):
        _capi.check(_capi.MOT_ESHAPE)
    assert lib.mot_pull_bytes(one, two, 2, 0, 8, 456, 457, _capi.PULL_LEFT, None) == _capi.MOT_OK  # T == 0 no-op
    assert lib.mot_pull_bytes(one, two, 1, 16, 8, 456, 457, 7, None) == _capi.MOT_EINVAL
    assert lib.mot_tokens_to_bytes(one, 4, two, 3, 10, 8, one, None, None) == _capi.MOT_EINVAL  # elem size
    assert lib.mot_tokens_to_bytes(one, 4, two, 2, 10, 65, one, None, None) == _capi.MOT_EUNSUPPORTED  # bpt > 64
    d = _capi.MotEmbedMixDesc()
    assert lib.mot_embed_mix_fwd(C.byref(d), None) == _capi.MOT_EINVAL  # struct_size 0
    d.struct_size = C.sizeof(d)
    d.mode = _capi.MIX_SUM
    d.tokens = d.tok_table = d.out = d.byte_table = d.ttb = 64
    d.n_rows, d.tokens_per_row, d.bpt = 1, 4, 16
    d.tok_rows, d.tok_dim, d.model_dim, d.byte_rows, d.byte_dim = 10, 768, 768, 458, 40
    d.id_source, d.ttb_rows, d.ttb_elem_bytes = _capi.IDS_FROM_TTB, 10, 2
    assert lib.mot_embed_mix_fwd(C.byref(d), None) == _capi.MOT_ESHAPE  # 16*40 != 768
    assert b"bpt*byte_dim" in lib.mot_last_error()
