"""Batch form of the reference's ``SinterBpOsdDecoder`` (sinter_decoders/sinter_bposd_decoder.py:9-130).

The reference reads every shot of a ``b8`` file, calls ``BpOsdDecoder.decode`` once per shot in a Python loop and writes
``(observables_matrix @ correction) % 2`` per shot (:115-130).  Here the packed file goes to the GPU as it is: one
``ldpc_hip_bp_decode_b8`` call unpacks the detection events, runs BP (+ OSD) over all shots and packs the predicted
observables -- neither the syndromes nor the corrections are ever materialised on the host.

``decode_b8_files`` is the stim-free core (check matrix, priors and observables matrix given explicitly).
``SinterBpOsdDecoder.decode_via_files`` has the reference's signature; the detector error model at ``dem_path`` is
turned into matrices by ``ldpc_amd.ckt_noise.dem_matrices``, which reads the ``.dem`` text itself (no ``stim``).
"""
from __future__ import annotations

import pathlib

import numpy as np


def read_b8(path, bits_per_shot: int, num_shots: int | None = None) -> np.ndarray:
    """stim "b8" shot data -> ``(shots, ceil(bits / 8))`` uint8, still packed (bit i of a shot = bit i % 8 of byte i // 8)."""
    nb = (bits_per_shot + 7) // 8
    raw = np.fromfile(str(path), dtype=np.uint8)
    if nb == 0:
        return np.zeros((num_shots or 0, 0), np.uint8)
    if raw.size % nb:
        raise ValueError(f"{path}: {raw.size} bytes is not a whole number of {nb}-byte shots")
    shots = raw.reshape(-1, nb)
    if num_shots is not None and shots.shape[0] != num_shots:
        raise ValueError(f"{path}: holds {shots.shape[0]} shots, expected {num_shots}")
    return shots


def write_b8(path, packed: np.ndarray) -> None:
    np.ascontiguousarray(packed, np.uint8).tofile(str(path))


def _osd_code(method) -> int:
    key = str(method).lower()
    if key in ["osd_0", "0", "osd0"]:
        return 1
    if key in ["osd_e", "e", "exhaustive"]:
        return 2
    if key in ["osd_cs", "1", "cs", "combination_sweep"]:
        return 3
    raise ValueError(f"ERROR: OSD method '{method}' invalid. Please choose from the following methods: 'OSD_0', 'OSD_E' or 'OSD_CS'.")


def decode_b8_files(check_matrix, priors, observables_matrix, *, num_shots: int, dets_b8_in_path, obs_predictions_b8_out_path,
                    max_iter=0, bp_method="ms", ms_scaling_factor=0.625, osd_method="osd0", osd_order=0, device: int = -1) -> None:
    """Decode every shot of ``dets_b8_in_path`` and write the predicted observables to ``obs_predictions_b8_out_path``."""
    import scipy.sparse as sp
    from ldpc_amd.bp_decoder._bp_decoder import _bp_method_code  # same alias table as BpDecoder
    from ldpc_amd.engine import HipBpEngine

    h = sp.csr_matrix(check_matrix, dtype=np.uint8)
    h.eliminate_zeros()
    h.sort_indices()
    m, n = h.shape
    method = _bp_method_code(bp_method)
    eng = HipBpEngine(h.indptr, h.indices, n, np.asarray(priors, np.float64), max_iter if max_iter else n, method,
                      float(ms_scaling_factor), device=device)
    code = _osd_code(osd_method)
    eng.set_osd(code, 0 if code == 1 else int(osd_order))
    eng.set_observables(observables_matrix)
    dets = read_b8(dets_b8_in_path, m, num_shots)
    obs = eng.decode_b8(dets, with_osd=True)[0]
    write_b8(obs_predictions_b8_out_path, obs)
    eng.close()


try:  # pragma: no cover - sinter is not part of this image
    from sinter import Decoder as _SinterDecoder
except ImportError:
    _SinterDecoder = object


class SinterBpOsdDecoder(_SinterDecoder):
    """Constructor keywords and defaults of the reference class (sinter_bposd_decoder.py:37-56); a ``sinter.Decoder``
    when sinter is installed, a plain class with the same methods otherwise."""

    def __init__(self, max_iter=0, bp_method="ms", ms_scaling_factor=0.625, schedule="parallel", omp_thread_count=1,
                 serial_schedule_order=None, osd_method="osd0", osd_order=0):
        self.max_iter = max_iter
        self.bp_method = bp_method
        self.ms_scaling_factor = ms_scaling_factor
        self.schedule = schedule
        self.omp_thread_count = omp_thread_count
        self.serial_schedule_order = serial_schedule_order
        self.osd_method = osd_method
        self.osd_order = osd_order

    def _configure(self, dem_path):
        """``self.matrices`` and ``self.bposd`` as the reference sets them (:97-113); the model text is read without stim."""
        from ldpc_amd.bposd_decoder import BpOsdDecoder
        from ldpc_amd.ckt_noise.dem_matrices import detector_error_model_to_check_matrices
        self.matrices = detector_error_model_to_check_matrices(pathlib.Path(dem_path), allow_undecomposed_hyperedges=True)
        self.bposd = BpOsdDecoder(self.matrices.check_matrix, error_channel=list(self.matrices.priors), max_iter=self.max_iter,
                                  bp_method=self.bp_method, ms_scaling_factor=self.ms_scaling_factor, schedule=self.schedule,
                                  omp_thread_count=self.omp_thread_count, serial_schedule_order=self.serial_schedule_order,
                                  osd_method=self.osd_method, osd_order=self.osd_order)

    def decode_via_files(self, *, num_shots: int, num_dets: int, num_obs: int, dem_path: pathlib.Path,
                         dets_b8_in_path: pathlib.Path, obs_predictions_b8_out_path: pathlib.Path, tmp_dir: pathlib.Path) -> None:
        self._configure(dem_path)
        mats = self.matrices
        if mats.check_matrix.shape[0] != num_dets or mats.observables_matrix.shape[0] != num_obs:
            raise ValueError("detector error model does not match num_dets / num_obs")
        if self.schedule == "parallel":  # the packed file goes to the device as it is
            decode_b8_files(mats.check_matrix, list(mats.priors), mats.observables_matrix, num_shots=num_shots,
                            dets_b8_in_path=dets_b8_in_path, obs_predictions_b8_out_path=obs_predictions_b8_out_path,
                            max_iter=self.max_iter, bp_method=self.bp_method, ms_scaling_factor=self.ms_scaling_factor,
                            osd_method=self.osd_method, osd_order=self.osd_order)
            return
        # other schedules: unpack on the host, one decode_batch, pack the predictions (reference :115-126, shot by shot there)
        import scipy.sparse as sp
        shots = np.unpackbits(read_b8(dets_b8_in_path, num_dets, num_shots), axis=1, bitorder="little", count=num_dets)
        corr = self.bposd.decode_batch(np.ascontiguousarray(shots), want_log_prob_ratios=False)
        predictions = np.asarray((sp.csr_matrix(mats.observables_matrix) @ corr.T.astype(np.int64)) % 2).T.astype(np.uint8)
        write_b8(obs_predictions_b8_out_path, np.packbits(predictions, axis=1, bitorder="little"))

    def decode(self, syndrome: np.ndarray) -> np.ndarray:
        """One shot, after ``decode_via_files`` has configured the decoder (reference :128-130)."""
        corr = self.bposd.decode(syndrome)
        return (self.matrices.observables_matrix @ corr) % 2
