"""CHECKER (test infrastructure, not product code): the reference's overlapping-window loop, shot by shot, in numpy.

Restates ``BaseOverlappingWindowDecoder._corr_multiple_rounds`` / ``decode`` / ``current_round_inds`` of
/root/reference/src_python/ldpc/ckt_noise/base_overlapping_window_decoder.py (:96-137, :68-94, :279-334) with the
``BpOsdOverlappingWindowDecoder`` hooks (bposd_overlapping_window.py:38-58: weights = the priors array itself, minimum
weight 0.0, one ``BpOsdDecoder(round_dcm, error_channel=list(weights), **config)`` per window, built on first use).

PINNED (round 3): tests/golden/window_*.npz are produced by the reference's OWN modules -- tests/golden/make_golden_window.py
imports base_overlapping_window_decoder.py and bposd_overlapping_window.py byte-identical from a scratch build of the
reference (behind a ``stim`` placeholder that raises on any use; the constructor's stim conversion is the only part bypassed)
and runs ``_corr_multiple_rounds_batch`` / ``decode_batch`` / ``decode`` around the reference's own ``BpOsdDecoder``.  This
restatement reproduces those fixtures byte for byte (tests/test_ckt_noise_host.py), and
tests/test_golden_generators.py re-runs the generator in --check mode wherever /root/reference is present.  The window
decodes inside the loop go through ``oracle.RefBpOsd`` (the real reference BP + OSD, oracle/_ref) when that library
exists, else through ``oracle.BpOracle`` (itself pinned to the reference by tests/golden/).
``BpOsdDecoder.decode``'s shortcut for an
all-zero syndrome (bposd_decoder.pyx:118-123) is part of the restatement.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

import oracle

_OSD = {"osd_0": (1, 0), "osd0": (1, 0), "osd_e": 2, "osd_cs": 3}
_BP = {"minimum_sum": "minimum_sum", "ms": "minimum_sum", "product_sum": "product_sum", "ps": "product_sum"}


def round_inds(dcm, decoding, window, commit, num_checks):
    """base_overlapping_window_decoder.py:279-334."""
    start = decoding * commit * num_checks
    end_commit = start + num_checks * commit
    end_decoding = start + num_checks * window
    min_index = dcm[start:end_commit, :].nonzero()[1].min()
    max_commit = dcm[start:end_commit, :].nonzero()[1].max()
    max_decoding = dcm[start:end_decoding, :].nonzero()[1].max()
    return slice(min_index, max_commit + 1), slice(min_index, max_decoding + 1), slice(start, end_commit), slice(start, end_decoding)


class WindowOracle:
    def __init__(self, check_matrix, observables_matrix, priors, *, decodings, window, commit, num_checks,
                 max_iter=30, bp_method="minimum_sum", ms_scaling_factor=1.0, osd_method="osd_0", osd_order=0, inner=None):
        self.dcm = sp.csr_matrix(check_matrix, dtype=np.uint8)
        self.obs = sp.csr_matrix(observables_matrix, dtype=np.uint8)
        self.weights = np.array(priors, dtype=np.float64)  # mutated as windows are first reached (:135), like the priors array
        self.decodings, self.window, self.commit, self.num_checks = decodings, window, commit, num_checks
        self.cfg = dict(max_iter=max_iter, bp_method=_BP[bp_method], ms_scaling_factor=ms_scaling_factor)
        code = _OSD[str(osd_method).lower()]
        self.osd = code if isinstance(code, tuple) else (code, int(osd_order))
        self.inner = inner if inner is not None else ("ref" if oracle.have_ref() else "oracle")
        self._decoders = {}

    def _decoder(self, decoding, round_dcm):
        if decoding not in self._decoders:  # :253-261
            if self.inner == "ref":
                d = oracle.RefBpOsd(round_dcm, error_channel=self.weights.copy(), osd_method=self.osd[0], osd_order=self.osd[1], **self.cfg)
                self._decoders[decoding] = lambda s, d=d: d.decode_batch(s[None, :], want_llr=False)[0][0]
            else:
                d = oracle.BpOracle(round_dcm, error_channel=self.weights.copy(), **self.cfg)
                self._decoders[decoding] = lambda s, d=d: d.bposd_decode_batch(s[None, :], self.osd[0], self.osd[1], want_llr=False)[0][0]
        return self._decoders[decoding]

    def corr(self, syndrome):
        """_corr_multiple_rounds (:96-137); ``syndrome`` (uint8) is updated in place as there."""
        total = np.zeros(self.dcm.shape[1], dtype=np.uint8)
        for decoding in range(self.decodings):
            commit_inds, dec_inds, _, synd_dec = round_inds(self.dcm, decoding, self.window, self.commit, self.num_checks)
            round_dcm = self.dcm[synd_dec, :]
            dec = self._decoder(decoding, round_dcm)
            s = np.ascontiguousarray(syndrome[synd_dec], dtype=np.uint8)
            c = dec(s) if s.any() else np.zeros(self.dcm.shape[1], np.uint8)  # bposd_decoder.pyx:118-123
            if decoding != self.decodings - 1:
                total[commit_inds] += c[commit_inds]
                syndrome[synd_dec] ^= (round_dcm @ total % 2).astype(np.uint8)
            else:
                total[dec_inds] += c[dec_inds]
            self.weights[commit_inds] = 0.0
        return total

    def decode(self, syndrome):
        return (self.obs @ self.corr(syndrome)) % 2

    def decode_batch(self, shots):
        shots = np.array(shots, dtype=np.uint8)
        corrs = np.stack([self.corr(shots[i]) for i in range(shots.shape[0])]) if len(shots) else np.zeros((0, self.dcm.shape[1]), np.uint8)
        preds = np.stack([(self.obs @ c) % 2 for c in corrs]).astype(bool) if len(shots) else np.zeros((0, self.obs.shape[0]), bool)
        return preds, corrs, shots
