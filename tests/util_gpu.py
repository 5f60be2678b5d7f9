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


class RelErr(float):
    """max|got - ref| / max|ref| that explains itself when a bar is not met: ``assert rel(got, ref) < TOL`` raises with
    the worst element's index, both values and how many elements are over the bar, so a failure in an ordinary run
    names what was wrong (pytest shows the expression, i.e. which tensor, on the failing line)."""

    def __new__(cls, got, ref):
        ref = np.asarray(ref, dtype=np.float64)
        got = np.asarray(got, dtype=np.float64).reshape(ref.shape)
        diff = np.abs(got - ref)
        scale = max(float(np.abs(ref).max()) if ref.size else 0.0, 1e-30)
        self = super().__new__(cls, (float(diff.max()) if ref.size else 0.0) / scale)
        self._got, self._ref, self._diff, self._scale = got, ref, diff, scale
        return self

    def detail(self, bar) -> str:
        if not self._ref.size:
            return "empty tensors"
        idx = np.unravel_index(int(np.argmax(self._diff)), self._diff.shape)
        over = int((self._diff > bar * self._scale).sum())
        return (f"relative error {float(self):.3e} is not below {bar:.3e}: worst element at index {tuple(int(i) for i in idx)} of shape "
                f"{self._ref.shape}: got {self._got[idx]!r}, reference {self._ref[idx]!r}, |difference| {self._diff[idx]:.3e}, "
                f"max|reference| {self._scale:.3e}; {over} of {self._ref.size} elements over the bar; non-finite got: "
                f"{int((~np.isfinite(self._got)).sum())}")

    def __lt__(self, bar):
        if float(self) < float(bar):
            return True
        raise AssertionError(self.detail(float(bar)))

    __le__ = __lt__


def rel(got, ref) -> RelErr:
    return RelErr(got, ref)
