"""Parity-check-matrix constructors.

``hamming_code`` / ``rep_code`` / ``ring_code`` mirror the reference's ``ldpc.codes`` surface
(reference: src_python/ldpc/codes/hamming_code.py:5-63, rep_code.py:5-40, rep_code.py:43-84) so that
BASELINE.json config 1 (``ldpc.codes.hamming_code(5)``) reads the same here.

``regular_ldpc_code`` / ``irregular_ldpc_code`` / ``rotated_surface_code_x`` / ``bivariate_bicycle_hx`` / ``hypergraph_product_hx`` are new: the reference
ships no generator for the codes BASELINE.json's GPU configs are quoted on.
"""
from ldpc_amd.codes.classical import hamming_code, rep_code, ring_code
from ldpc_amd.codes.synthetic import (
    regular_ldpc_code,
    irregular_ldpc_code,
    rotated_surface_code_x,
    bivariate_bicycle_hx,
    hypergraph_product_hx,
)

__all__ = [
    "hamming_code",
    "rep_code",
    "ring_code",
    "regular_ldpc_code",
    "irregular_ldpc_code",
    "rotated_surface_code_x",
    "bivariate_bicycle_hx",
    "hypergraph_product_hx",
]
