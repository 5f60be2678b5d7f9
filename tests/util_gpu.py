"""Helpers for the -m gpu parity tests: move seeded numpy inputs to cuda:0 and compare the HIP
path (through the package -> ctypes -> C ABI) with the CPU oracle."""
import numpy as np
import torch

DEV = torch.device("cuda", 0)


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def host(t):
    return t.detach().cpu().numpy()


def f32(a):
    return np.asarray(a, dtype=np.float64).astype(np.float32)


# The 1e-6 bar of BASELINE.json's north_star for the fp32 mixed embeddings, written out:
# |hip - ref| <= 1e-6 + 1e-6*|ref| elementwise (outputs are rms-normalised, i.e. O(1)).
RTOL = ATOL = 1e-6


def assert_close(got, ref, rtol=RTOL, atol=ATOL):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref) - (atol + rtol * np.abs(ref))
    assert got.shape == ref.shape
    assert np.isfinite(got).all()
    assert (err <= 0).all(), f"max excess {err.max():.3e}; max abs err {np.abs(got - ref).max():.3e}"
