"""Counter-based PRNG shared bit-for-bit by the Python harness, the C oracle and the HIP kernels.

The stream is SplitMix64: the ``idx``-th output of a generator seeded with ``seed`` is
``mix64(seed + (idx + 1) * GAMMA)`` (all arithmetic mod 2**64).  Because the output is a pure
function of ``(seed, idx)`` any shard of a synthetic batch can be regenerated on any rank or on
the device without shipping data (SURVEY.md §8d "counter-based so any shard can be regenerated").

The same three constants and the same shifts appear in ``oracle/bp_oracle.c`` (``sm64``) and in
``ldpc_amd/csrc/bp_hip.hip`` (``sm64``); tests pin them against each other.

Not taken from the reference: ``ldpc.noise_models.bsc.generate_bsc_error`` (bsc.py:4-23) uses
``np.random.binomial``, whose stream is neither counter-based nor reproducible on a GPU.
"""
from __future__ import annotations

import numpy as np

GAMMA = 0x9E3779B97F4A7C15
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB
_MASK = (1 << 64) - 1


def mix64_int(z: int) -> int:
    """SplitMix64 finaliser on a Python int (mod 2**64)."""
    z &= _MASK
    z = ((z ^ (z >> 30)) * _M1) & _MASK
    z = ((z ^ (z >> 27)) * _M2) & _MASK
    return z ^ (z >> 31)


def sm64_int(seed: int, idx: int) -> int:
    """idx-th SplitMix64 output for ``seed`` (scalar, Python ints)."""
    return mix64_int((seed + (idx + 1) * GAMMA) & _MASK)


def sm64(seed: int, idx: np.ndarray) -> np.ndarray:
    """Vectorised ``sm64_int`` over a uint64 index array."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (idx + np.uint64(1)) * np.uint64(GAMMA) + np.uint64(seed & _MASK)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, idx: np.ndarray) -> np.ndarray:
    """53-bit uniform doubles in [0, 1): ``(sm64 >> 11) * 2**-53``."""
    return (sm64(seed, idx) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def bernoulli_threshold(p: float) -> int:
    """Integer threshold T such that ``(sm64 >> 11) < T``  <=>  ``uniform01 < p``.

    Comparing 53-bit integers keeps host, oracle and device decisions identical without relying on
    any floating-point conversion on the device.
    """
    if not (0.0 <= p <= 1.0):
        raise ValueError("p must be in [0, 1]")
    t = int(np.ceil(p * 9007199254740992.0))
    return min(max(t, 0), 1 << 53)
