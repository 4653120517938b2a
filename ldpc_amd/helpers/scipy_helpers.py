"""Parity-check-matrix ingest: the validation ``ldpc.BpDecoder`` applies to ``pcm``.

Behavioural twin of the reference's ``convert_to_binary_sparse``
(src_python/ldpc/helpers/scipy_helpers.py:6-68, called from Py2BpSparse, _bp_decoder.pyx:23):
same accepted container types and dtypes, same exception types for the same bad inputs, explicit
zeros dropped.  Written against that contract, not from its code.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse

_OK_DTYPES = (np.dtype(np.uint8), np.dtype(np.int8), np.dtype(int), np.dtype(float))


def convert_to_binary_sparse(matrix):
    """Return ``matrix`` as a scipy sparse matrix holding only ones.

    Raises ``TypeError`` if ``matrix`` is neither ``np.ndarray`` nor scipy sparse, or its dtype is not
    one of uint8 / int8 / int / float; ``ValueError`` if any stored value is not 0 or 1.
    """
    is_dense = isinstance(matrix, np.ndarray)
    if not (is_dense or isinstance(matrix, scipy.sparse.spmatrix)):
        raise TypeError(
            f"Input must be a binary numpy array or scipy sparse matrix, not {type(matrix)}")
    if np.dtype(matrix.dtype) not in _OK_DTYPES:
        raise TypeError(f"Input matrix must have dtype uint8, int8, or int, not {matrix.dtype}")
    # a dense input is narrowed to uint8 BEFORE the binary test, as in the reference (so e.g. 256
    # wraps to an explicit zero and is dropped); a sparse input is tested on its stored values
    out = scipy.sparse.csr_matrix(matrix, dtype=np.uint8) if is_dense else matrix
    if np.any((out.data != 0) & (out.data != 1)):
        raise ValueError("Input matrix must be a binary matrix.")
    if out.dtype == float:
        out = out.astype(np.uint8)
    out.eliminate_zeros()
    return out
