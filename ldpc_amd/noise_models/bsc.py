"""Binary-symmetric-channel noise.

``generate_bsc_error`` keeps the reference's name and meaning (noise_models/bsc.py:4-23: ``n`` iid
Bernoulli(error_rate) bits, uint8).  ``generate_bsc_batch`` is the counter-based batch form the
benchmarks and parity tests use: bit ``(shot, j)`` depends only on
``(seed, shot * n + j)``, so host, C oracle and HIP kernel regenerate identical shots.
"""
from __future__ import annotations

import numpy as np

from ldpc_amd.prng import bernoulli_threshold, sm64


def generate_bsc_error(n: int, error_rate: float) -> np.ndarray:
    return np.random.binomial(1, error_rate, n).astype(np.uint8)


def generate_bsc_batch(n: int, error_rate: float, seed: int, shot0: int, shots: int) -> np.ndarray:
    """``(shots, n)`` uint8 errors for shots ``shot0 .. shot0 + shots - 1`` of stream ``seed``."""
    idx = (np.arange(shot0, shot0 + shots, dtype=np.uint64)[:, None] * np.uint64(n)
           + np.arange(n, dtype=np.uint64)[None, :])
    thr = np.uint64(bernoulli_threshold(error_rate))
    return ((sm64(seed, idx) >> np.uint64(11)) < thr).astype(np.uint8)
