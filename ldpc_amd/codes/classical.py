"""Small classical codes with the reference's constructor names and conventions."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def hamming_code(rank: int) -> sp.csr_matrix:
    """Hamming-code parity-check matrix of the given rank (``rank x (2**rank - 1)``).

    Column ``c`` (0-based) is the big-endian binary expansion of ``c + 1``: row 0 holds the most
    significant bit.  Same matrix as the reference's ``ldpc.codes.hamming_code``
    (src_python/ldpc/codes/hamming_code.py:5-63); raises ``TypeError`` for a non-int rank as it does.
    """
    if not isinstance(rank, int):
        raise TypeError("The input variable 'rank' must be of type 'int'.")
    n = (1 << rank) - 1
    value = np.arange(1, n + 1, dtype=np.int64)[None, :]
    shift = np.arange(rank - 1, -1, -1, dtype=np.int64)[:, None]
    dense = ((value >> shift) & 1).astype(np.uint8)
    return sp.csr_matrix(dense, dtype=np.uint8)


def _chain(distance: int, closed: bool) -> sp.csr_matrix:
    if distance < 2:
        raise ValueError("Distance should be greater than or equal to 2.")
    m = distance if closed else distance - 1
    r = np.repeat(np.arange(m), 2)
    c = np.stack([np.arange(m), (np.arange(m) + 1) % distance], axis=1).reshape(-1)
    return sp.csr_matrix(
        (np.ones(2 * m, dtype=np.uint8), (r, c)), shape=(m, distance), dtype=np.uint8
    )


def rep_code(distance: int) -> sp.csr_matrix:
    """Repetition code: check ``i`` touches bits ``i`` and ``i + 1`` (rep_code.py:5-40)."""
    return _chain(distance, closed=False)


def ring_code(distance: int) -> sp.csr_matrix:
    """Closed-loop repetition code: extra check on bits ``0`` and ``distance - 1`` (rep_code.py:43-84)."""
    return _chain(distance, closed=True)
