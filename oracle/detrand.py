"""Deterministic pseudo-random fills that need no RNG library (oracle / tests only).

Golden fixtures would be tens of MB if they stored hash tables and network
weights.  Instead both the fixture generator and the tests rebuild those
arrays from (n, seed) with pure integer arithmetic, so the values are the
same on every machine and every numpy version.
"""
import numpy as np

_MUL = np.uint64(0x9E3779B97F4A7C15)
_SEED_MUL = np.uint64(0xD1B54A32D192ED03)


def unit24(n: int, seed: int) -> np.ndarray:
    """n floats in [0, 1), each an exact multiple of 2^-24 (float32-exact)."""
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (i + np.uint64(1)) * _MUL + np.uint64(seed + 1) * _SEED_MUL
        z ^= z >> np.uint64(29)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(32)
    u = (z >> np.uint64(40)).astype(np.uint32)  # top 24 bits
    return u.astype(np.float32) * np.float32(2.0 ** -24)


def uniform(n: int, seed: int, lo: float, hi: float) -> np.ndarray:
    """float32 values in [lo, hi); plain float32 multiply-add, no fused ops."""
    u = unit24(n, seed)
    return (np.float32(lo) + (np.float32(hi) - np.float32(lo)) * u).astype(np.float32)


def integers(n: int, seed: int, lo: int, hi: int) -> np.ndarray:
    """int64 values in [lo, hi)."""
    u = unit24(n, seed).astype(np.float64)
    return (lo + np.floor(u * (hi - lo))).astype(np.int64)
