"""Overlapping-window decoding of a circuit-level detector error model, every shot of a batch at once.

Mirror of the reference's ``BaseOverlappingWindowDecoder`` (ckt_noise/base_overlapping_window_decoder.py:7-276): same
constructor, ``decode``, ``decode_batch(shots, bit_packed_shots=..., bit_packed_predictions=...)``, the same hooks for
subclasses (``_get_dcm``, ``_get_logical_observables_matrix``, ``_get_weights``, ``_min_weight``, ``_init_decoder``) and
``current_round_inds``.  The reference decodes shot by shot inside each window (:203-214); the windows depend on each
other (a window's commit changes the next window's syndrome) but the shots do not, so here a window is ONE
``decode_batch`` call over all shots on the device, the commit and the syndrome update (:206-208) are one device
``H e`` product and two slice updates, and only the predictions come back to the host.

The arithmetic is the reference's, including the parts one might not expect: the update XORs ``round_dcm @ total_corr``
with the WHOLE correction committed so far (:208), corrections are accumulated with ``+=`` on uint8 (:206, :212), a
window decoder is built once -- with the weights as they were when its window was first reached -- and then reused
(:253-261), and the caller's ``shots`` array is modified in place (:208).

``model`` may be DEM text, a path, or a ``stim.DetectorErrorModel`` (see ``dem_matrices``); ``stim`` is not required.
"""
from __future__ import annotations

import numpy as np
from scipy.sparse import csr_matrix

from ldpc_amd.ckt_noise.dem_matrices import detector_error_model_to_check_matrices


class BaseOverlappingWindowDecoder:
    def __init__(self, model, decodings: int, window: int, commit: int, num_checks: int, **decoder_kwargs) -> None:
        self.decodings = decodings
        self.window = window
        self.commit = commit
        self.num_checks = num_checks

        self.dem_matrices = detector_error_model_to_check_matrices(model, allow_undecomposed_hyperedges=True)
        self.num_detectors = self.dem_matrices.check_matrix.shape[0]

        rounds = (self.window - self.commit) + self.decodings * self.commit
        if not self.num_detectors % rounds == 0:  # :41-48
            raise ValueError(
                f"The number of detectors must be a multiple of the number of rounds. There are {self.num_detectors} detectors and "
                f"{rounds} rounds."
                "Dem matrices must be decomposed into a number of rounds that is a multiple of the number of detectors."
                f"You expected {self.num_checks * rounds}")

        self.dcm = self._get_dcm()
        self.logical_observables_matrix = self._get_logical_observables_matrix()

    # ---- hooks (:53-66, 216-276) ----------------------------------------------------------------------------------
    def _get_dcm(self) -> csr_matrix:
        raise NotImplementedError("This method must be implemented by the subclass.")

    def _get_logical_observables_matrix(self):
        raise NotImplementedError("This method must be implemented by the subclass.")

    def _get_weights(self) -> np.ndarray:
        raise NotImplementedError("This method must be implemented by the subclass.")

    @property
    def _min_weight(self) -> float:
        raise NotImplementedError("This method must be implemented by the subclass.")

    def _init_decoder(self, round_dcm, weights: np.ndarray):
        raise NotImplementedError("This method must be implemented by the subclass.")

    def _get_decoder(self, decoding: int, round_dcm, weights: np.ndarray):
        if not hasattr(self, "_decoders"):
            self._decoders = {}
        if decoding not in self._decoders:
            self._decoders[decoding] = self._init_decoder(round_dcm, weights)
        return self._decoders[decoding]

    # ---- one shot (:68-137) ---------------------------------------------------------------------------------------
    def decode(self, syndrome: np.ndarray) -> np.ndarray:
        corr = self._corr_multiple_rounds(syndrome)
        return (self.logical_observables_matrix @ corr) % 2

    def _corr_multiple_rounds(self, syndrome: np.ndarray) -> np.ndarray:
        """One shot = a batch of one: same windows, same decoders; ``syndrome`` is updated in place as in the reference."""
        shots = np.asarray(syndrome)[None, :]
        return self._corr_multiple_rounds_batch(shots)[0]

    # ---- a batch of shots (:139-214) ------------------------------------------------------------------------------
    def decode_batch(self, shots: np.ndarray, *, bit_packed_shots: bool = False, bit_packed_predictions: bool = False) -> np.ndarray:
        if bit_packed_shots:  # the packed rows cross PCIe (1/8 of the bytes) and are unpacked on the device
            import torch
            packed = np.ascontiguousarray(shots, dtype=np.uint8)
            if packed.ndim != 2 or packed.shape[1] != (self.num_detectors + 7) // 8:
                raise ValueError(f"bit-packed shots must have shape (num_shots, {(self.num_detectors + 7) // 8})")
            if packed.shape[0]:
                dev = torch.device("cuda", torch.cuda.current_device())
                shots = self._observables_engine().unpack_b8(torch.from_numpy(packed).to(dev), self.num_detectors)
            else:
                shots = np.zeros((0, self.num_detectors), np.uint8)
        total, synd = self._corr_on_device(shots)
        if not bit_packed_shots and total is not None and isinstance(shots, np.ndarray) and shots.dtype == np.uint8:
            shots[...] = synd.cpu().numpy()  # as in the reference, the caller's unpacked shots end up updated (:208)
        if total is None:
            predictions = np.zeros((0, self.logical_observables_matrix.shape[0]), dtype=bool)
        else:  # (L @ corr) % 2 per shot (:169-171) as one device product; counts above one reduce to their parity first
            predictions = self._observables_engine().mulvec_batch((total & 1).contiguous()).cpu().numpy().astype(bool)
        if bit_packed_predictions:
            predictions = np.packbits(predictions, axis=1, bitorder="little")
        return predictions

    def _observables_engine(self):
        """A handle on the logical-observables matrix, used only for its ``L e`` product (gf2sparse.hpp:177-214)."""
        if getattr(self, "_obs_engine", None) is None:
            from ldpc_amd.engine import HipBpEngine
            lom = csr_matrix(self.logical_observables_matrix, dtype=np.uint8)
            lom.eliminate_zeros()
            lom.sort_indices()
            self._obs_engine = HipBpEngine(lom.indptr, lom.indices, lom.shape[1], np.full(lom.shape[1], 0.25), 1, 1, 1.0)
        return self._obs_engine

    def _window_plan(self):
        """Per window: the index slices, the window's rows of the check matrix, its decoder -- built on first use."""
        if getattr(self, "_plan", None) is None:
            weights = self._get_weights()
            plan = []
            for decoding in range(self.decodings):
                commit_inds, dec_inds, synd_commit_inds, synd_dec_inds = current_round_inds(
                    dcm=self.dcm, decoding=decoding, window=self.window, commit=self.commit, num_checks=self.num_checks)
                round_dcm = self.dcm[synd_dec_inds, :]
                decoder = self._get_decoder(decoding, round_dcm, weights)
                plan.append((commit_inds, dec_inds, synd_dec_inds, decoder))
                weights[commit_inds] = self._min_weight  # :135 / :214
            self._plan = plan
        return self._plan

    def _corr_multiple_rounds_batch(self, shots: np.ndarray) -> np.ndarray:
        """``corrs[i] == _corr_multiple_rounds(shots[i])`` of the reference; all shots move through a window together."""
        total, synd = self._corr_on_device(shots)
        if total is None:
            return np.zeros((0, self.dcm.shape[1]), dtype=np.uint8)
        if isinstance(shots, np.ndarray) and shots.dtype == np.uint8:
            shots[...] = synd.cpu().numpy()  # the reference leaves the updated syndromes in the caller's array (:208)
        return total.cpu().numpy()

    def _corr_on_device(self, shots):
        """The window loop (:189-214) with every array resident on the GPU: (corrections, updated syndromes) as tensors."""
        import torch
        plan = self._window_plan()
        if tuple(shots.shape[1:]) != (self.num_detectors,):
            raise ValueError(f"shots must have shape (num_shots, {self.num_detectors})")
        num_shots, num_errors = shots.shape[0], self.dcm.shape[1]
        if num_shots == 0:
            return None, None
        if isinstance(shots, torch.Tensor):  # already on the device (bit-packed input): worked on in place
            synd, dev = shots, shots.device
        else:
            dev = torch.device("cuda", torch.cuda.current_device())
            synd = torch.from_numpy(np.ascontiguousarray(np.asarray(shots).astype(np.uint8, copy=False))).to(dev)
        total = torch.zeros((num_shots, num_errors), dtype=torch.uint8, device=dev)
        for decoding, (commit_inds, dec_inds, synd_dec_inds, decoder) in enumerate(plan):
            corr = decoder.decode_batch(synd[:, synd_dec_inds].contiguous(), want_log_prob_ratios=False)
            if decoding != self.decodings - 1:
                total[:, commit_inds] += corr[:, commit_inds]
                # round_dcm @ total_corr % 2 (:208): the device product takes bits, and H t = H (t mod 2) over GF(2)
                bits = (total & 1).contiguous()
                flips = decoder.mulvec_batch(bits) if hasattr(decoder, "mulvec_batch") else decoder._get_engine().mulvec_batch(bits)
                synd[:, synd_dec_inds] ^= flips
            else:
                total[:, dec_inds] += corr[:, dec_inds]
        return total, synd


def current_round_inds(dcm: csr_matrix, decoding: int, window: int, commit: int, num_checks: int) -> tuple:
    """Column (error) and row (detector) slices of one window (reference :279-334).

    Rows: ``num_checks`` detectors per round, the window starts at round ``decoding * commit``.  Columns: from the
    smallest column met by the committed rows to the largest column met by the committed rows / by all rows of the window.
    """
    start = decoding * commit * num_checks
    end_commit = start + num_checks * commit
    end_decoding = start + num_checks * window
    cols_commit = dcm[slice(start, end_commit), :].nonzero()[1]
    cols_decoding = dcm[slice(start, end_decoding), :].nonzero()[1]
    min_index = cols_commit.min()
    commit_inds = slice(min_index, cols_commit.max() + 1)
    decoding_inds = slice(min_index, cols_decoding.max() + 1)
    return commit_inds, decoding_inds, slice(start, end_commit), slice(start, end_decoding)
