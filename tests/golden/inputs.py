"""Deterministic, platform-independent test inputs shared by make_goldens.py and the tests.

Dense inputs come from an integer hash mapped to 24-bit dyadic rationals in [-1, 1): every
value is exactly representable in float32 and the construction uses integer arithmetic only,
so the arrays are bit-identical on every machine / numpy version.  That lets the golden files
store OUTPUTS only (inputs are regenerated)."""
import numpy as np


def hash_matrix(n, d, seed=0, order="C"):
    i = np.arange(n, dtype=np.uint64)[:, None]
    k = np.arange(d, dtype=np.uint64)[None, :]
    h = (i * np.uint64(2654435761) + k * np.uint64(40503) + np.uint64(seed) * np.uint64(97531) + np.uint64(12345))
    h ^= h >> np.uint64(13)
    h *= np.uint64(0x9E3779B1)
    h ^= h >> np.uint64(17)
    v = (h & np.uint64((1 << 24) - 1)).astype(np.int64)
    x = ((v - (1 << 23)).astype(np.float64) / float(1 << 23)).astype(np.float32)
    return np.asfortranarray(x) if order == "F" else np.ascontiguousarray(x)


def hash_positive(n, d, seed=0):
    """values in (0, 1]: used where the reference expects probability-like rows"""
    return ((hash_matrix(n, d, seed).astype(np.float64) + 1.0) / 2.0 + 2.0 ** -24).astype(np.float32)
