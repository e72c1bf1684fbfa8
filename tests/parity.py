"""Bit-exact comparison helpers (test infrastructure)."""
import numpy as np


def bits_equal_mask(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Element-wise: identical bit patterns, or both NaN (NaN payload/sign is not part of the contract:
    x86 produces the negative default NaN for 0/0, gfx950 the positive one)."""
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.dtype == np.float32:
        return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    return a == b


def assert_bits_equal(got: np.ndarray, want: np.ndarray, what: str = ""):
    m = bits_equal_mask(got, want)
    if not m.all():
        bad = np.argwhere(~m)
        first = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {m.size} elements differ; first at {first}: got {got[first]!r} want {want[first]!r}")


def psnr(a: np.ndarray, b: np.ndarray, peak: float = 1.0) -> float:
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(peak * peak / mse)
