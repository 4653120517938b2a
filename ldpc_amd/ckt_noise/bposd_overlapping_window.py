"""BP+OSD inside each window (reference: ckt_noise/bposd_overlapping_window.py:10-58)."""
from __future__ import annotations

import numpy as np
from scipy.sparse import csr_matrix

from ldpc_amd.bposd_decoder import BpOsdDecoder
from ldpc_amd.ckt_noise.base_overlapping_window_decoder import BaseOverlappingWindowDecoder
from ldpc_amd.ckt_noise.config import DEFAULT_BPOSD_DECODER_ARGS


class _WindowBpOsd:
    """``BpOsdDecoder(round_dcm, error_channel=weights, ...)`` without the columns no row of the window touches.

    The reference hands every window the full-width matrix (:55-57), in which most columns are empty.  What BP + OSD-0
    do with an empty column does not depend on the syndrome: BP leaves its log-ratio at the prior (decision 1 iff the
    prior is >= 0.5, bp.hpp:276-298), OSD-0 never makes it a pivot and leaves it 0 (osd.hpp:110-117), and the other
    columns are ordered among themselves exactly as before.  So the device decodes the window's own columns and the
    empty ones are filled in here -- the matrix that must fit the OSD kernels' LDS shrinks from all errors to the
    window's.  Higher-order OSD enumerates candidates over ALL non-pivot columns in sorted order, empty ones included,
    so there the full matrix is kept.
    """

    def __init__(self, round_dcm, weights, config):
        round_dcm = csr_matrix(round_dcm)
        weights = np.asarray(weights, dtype=np.float64)
        self.n = round_dcm.shape[1]
        # order 0 takes the OSD-0 branch whatever the method (osd.hpp:114); the OSD_0 aliases are bposd_decoder.pyx:153-156
        zero_order = int(config.get("osd_order", 0)) == 0 or str(config.get("osd_method", 0)).lower() in ("osd_0", "0", "osd0")
        occupied = np.diff(round_dcm.tocsc().indptr) > 0
        self.cols = np.flatnonzero(occupied) if zero_order else np.arange(self.n)
        self.static_ones = np.flatnonzero(~occupied & (weights >= 0.5)) if zero_order else np.zeros(0, np.int64)
        # options that depend on n are resolved against the FULL width the reference's decoder sees, then restricted:
        # max_iter = 0 means n_full iterations (pyx:357); a serial_schedule_order lists all n_full bits, of which the
        # occupied ones keep their relative order (an empty column has no entry to update, bp.hpp:451-545)
        config = dict(config)
        if len(self.cols) != self.n:
            if int(config.get("max_iter", 0) or 0) == 0:
                config["max_iter"] = int(self.n)
            order = config.get("serial_schedule_order", None)
            if order is not None:
                order = np.asarray(order, dtype=np.int64)
                if len(order) != self.n:
                    raise ValueError("serial_schedule_order must have length equal to the block length of the code.")
                new_index = np.full(self.n, -1, np.int64)
                new_index[self.cols] = np.arange(len(self.cols))
                kept = new_index[order]
                config["serial_schedule_order"] = [int(v) for v in kept[kept >= 0]]  # (the decoder takes a list, pyx:620-623)
        self.inner = BpOsdDecoder(round_dcm[:, self.cols], error_channel=list(weights[self.cols]), **config)
        self._cols_dev = None

    def decode_batch(self, syndromes, want_log_prob_ratios: bool = False):
        """(shots, window detectors) on the device -> full-width corrections (shots, all errors) on the device."""
        import torch
        part = self.inner.decode_batch(syndromes, want_log_prob_ratios=False)
        if len(self.cols) == self.n:
            return part
        if self._cols_dev is None or self._cols_dev.device != part.device:
            self._cols_dev = torch.from_numpy(self.cols).to(part.device)
            self._ones_dev = torch.from_numpy(self.static_ones).to(part.device)
        full = torch.zeros((part.shape[0], self.n), dtype=torch.uint8, device=part.device)
        full[:, self._cols_dev] = part
        if len(self.static_ones):  # BP's decision where BP converged on a non-zero syndrome; OSD-0 and the zero shortcut give 0
            by_bp = self.inner.converge_batch & syndromes.any(dim=1)
            full[:, self._ones_dev] = by_bp.to(torch.uint8)[:, None]
        return full

    def mulvec_batch(self, vectors):
        """round_dcm @ v over GF(2) for every row (empty columns contribute nothing)."""
        v = vectors if len(self.cols) == self.n else vectors[:, self._cols_dev if self._cols_dev is not None else self.cols]
        return self.inner._get_engine().mulvec_batch(v.contiguous())


class BpOsdOverlappingWindowDecoder(BaseOverlappingWindowDecoder):
    def __init__(self, model, **kwargs):
        self.decoder_config = DEFAULT_BPOSD_DECODER_ARGS | kwargs.pop("decoder_config", {})
        super().__init__(model=model, **kwargs)

    def _get_dcm(self):
        return csr_matrix(self.dem_matrices.check_matrix)

    def _get_logical_observables_matrix(self):
        return self.dem_matrices.observables_matrix

    @property
    def _min_weight(self):
        return 0.0

    def _get_weights(self):
        return self.dem_matrices.priors

    def _init_decoder(self, round_dcm, weights: np.ndarray):
        return _WindowBpOsd(round_dcm, weights, self.decoder_config)
