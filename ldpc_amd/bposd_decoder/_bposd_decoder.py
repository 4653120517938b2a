"""Host-side mirror of the reference's ``BpOsdDecoder`` (src_python/ldpc/bposd_decoder/_bposd_decoder.pyx:8-299).

BP runs in the HIP kernels; rows BP leaves unconverged get ordered-statistics decoding on the device as well
(``ldpc_hip_bposd_decode_batch``): OSD_0 (``osd.hpp:110-117``), OSD_E and OSD_CS (``osd.hpp:119-187``; OSD_E up to
order 24, OSD_CS at any order).  There is no CPU fallback.
"""
from __future__ import annotations

import warnings

import numpy as np

from ldpc_amd.bp_decoder._bp_decoder import BpDecoderBase, PARALLEL, _UNSET, _check_pcm_type, _typed

OSD_OFF, OSD_0, EXHAUSTIVE, COMBINATION_SWEEP = 0, 1, 2, 3  # ldpc::osd::OsdMethod, osd.hpp:18-23


class BpOsdDecoder(BpDecoderBase):
    """Belief propagation + OSD decoder (constructor keywords as in the reference, pyx:54-58)."""

    def __init__(self, pcm, *, error_rate=_UNSET, error_channel=_UNSET, max_iter=_UNSET, bp_method=_UNSET,
                 ms_scaling_factor=_UNSET, schedule=_UNSET, omp_thread_count=_UNSET, random_schedule_seed=_UNSET,
                 serial_schedule_order=_UNSET, osd_method=0, osd_order: int = 0, input_vector_type: str = "syndrome",
                 **kwargs):
        _check_pcm_type(pcm)
        given = dict(error_rate=error_rate, error_channel=error_channel, max_iter=max_iter, bp_method=bp_method,
                     ms_scaling_factor=ms_scaling_factor, schedule=schedule, omp_thread_count=omp_thread_count,
                     random_schedule_seed=random_schedule_seed, serial_schedule_order=serial_schedule_order)
        passed = dict(kwargs)
        passed.update({k: v for k, v in given.items() if v is not _UNSET})
        super().__init__(pcm, **passed)  # the base class's __cinit__ runs first; then this signature's checks (pyx:51-55)
        _typed("error_rate", passed.get("error_rate"), float)
        _typed("error_channel", passed.get("error_channel"), list)
        _typed("max_iter", passed.get("max_iter", 0), int)
        _typed("bp_method", passed.get("bp_method", "minimum_sum"), str)
        _typed("schedule", passed.get("schedule", "parallel"), str)
        _typed("omp_thread_count", passed.get("omp_thread_count", 1), int)
        _typed("random_schedule_seed", passed.get("random_schedule_seed", 0), int)
        _typed("serial_schedule_order", passed.get("serial_schedule_order"), list)
        # `osd_order: int` of this signature is a C int: numbers are truncated to one, anything else is refused (pyx:54)
        if isinstance(osd_order, (int, float, np.integer, np.floating)):
            osd_order = int(osd_order)
        else:
            _typed("osd_order", osd_order, int, optional=False)
        _typed("input_vector_type", input_vector_type, str, optional=False)
        for key in kwargs.keys():  # pyx:57-59 (the reference's message says BpDecoder here too)
            if key not in ["channel_probs", "_device", "_backend", "device_ids"]:
                raise ValueError(f"Unknown parameter '{key}' passed to the BpDecoder constructor.")
        self._osd_method = OSD_OFF  # `new OsdDecoderCpp(pcm, OSD_OFF, 0, ...)`, pyx:67
        self._osd_order = 0
        self.osd_method = osd_method
        self.osd_order = osd_order
        self.input_vector_type = "syndrome"  # pyx:72
        self._bp_decoding = np.zeros(self.n, np.uint8)
        self._osd0_decoding = np.zeros(self.n, np.uint8)
        self._osdw_decoding = np.zeros(self.n, np.uint8)
        self._last_syndrome = np.zeros(self.m, np.uint8)
        self.bp_decoding_batch = None
        # additive: per row of the last decode_batch, 0 = BP converged, 1 = OSD solved H x = s, 2 = the syndrome lies outside
        # the image of H (no x solves it; the returned vector is flagged, see include/ldpc_hip.h: ldpc_hip_bposd_get_status)
        self.osd_status_batch = None

    # ---- OSD parameters (pyx:139-234) -------------------------------------------------------------
    @property
    def osd_method(self):
        return {OSD_0: "OSD_0", EXHAUSTIVE: "OSD_E", COMBINATION_SWEEP: "OSD_CS", OSD_OFF: "OSD_OFF"}.get(self._osd_method)

    @osd_method.setter
    def osd_method(self, method) -> None:
        key = str(method).lower()
        if key in ["osd_0", "0", "osd0"]:
            self._osd_method = OSD_0
            self._osd_order = 0
        elif key in ["osd_e", "e", "exhaustive"]:
            self._osd_method = EXHAUSTIVE
        elif key in ["osd_cs", "1", "cs", "combination_sweep"]:
            self._osd_method = COMBINATION_SWEEP
        elif key in ["off", "osd_off", "deactivated", -1]:
            self._osd_method = OSD_OFF
        else:
            raise ValueError(f"ERROR: OSD method '{method}' invalid. Please choose from the following methods:\
                'OSD_0', 'OSD_E' or 'OSD_CS'.")

    @property
    def osd_order(self) -> int:
        return self._osd_order

    @osd_order.setter
    def osd_order(self, order: int) -> None:
        _typed("order", order, int, optional=False)
        if order < 0:
            raise ValueError(f"ERROR: OSD order '{order}' invalid. Please choose a positive integer.")
        if self._osd_method == OSD_0 and order != 0:
            raise ValueError(f"ERROR: OSD order '{order}' invalid. The 'osd_method' is set to 'OSD_0'. The osd order must therefore be set to 0.")
        if self._osd_method == EXHAUSTIVE and order > 15:
            warnings.warn("WARNING: Running the 'OSD_E' (Exhaustive method) with search depth greater than 15 is not "
                          "recommended. Use the 'osd_cs' method instead.")
        self._osd_order = order

    def _require_supported(self):
        self._require_parallel()
        if self._osd_method == OSD_OFF:
            raise NotImplementedError("osd_method='OSD_OFF': the reference dereferences an unset LU object here (osd.hpp:63, 110); "
                                      "choose OSD_0, OSD_E or OSD_CS.")
        if self._osd_method == EXHAUSTIVE and self._osd_order > 24:
            raise NotImplementedError(
                f"osd_method={self.osd_method} with osd_order={self._osd_order} is not available on the MI355X path "
                "(OSD_E up to order 24 = 16.7 million candidates per syndrome; OSD_CS takes any order); there is no CPU fallback.")

    def _decode_osd(self, synd2d, want_llr=True, force_osd0=False):
        """BP + OSD through the active backend with this decoder's osd_method / osd_order."""
        method, order = (OSD_0, 0) if force_osd0 else (self._osd_method, self._osd_order)
        cy = self._get_cy() if self._schedule == PARALLEL else None  # the schedule setters live on the ctypes engine
        if cy is not None:
            cy.osd_method, cy.osd_order = method, order
            out = cy.decode_batch(np.ascontiguousarray(synd2d, np.uint8), want_llr, True)
            self._last_status = cy.osd_status
            return out
        eng = self._get_engine()
        eng.set_osd(method, order)
        out = eng.decode_batch(synd2d, want_llr=want_llr, osd=True)
        self._last_status = eng.osd_status(len(synd2d)) if hasattr(eng, "osd_status") else None
        return out

    # ---- decode (pyx:78-136) ----------------------------------------------------------------------
    def decode(self, syndrome: np.ndarray) -> np.ndarray:
        _typed("syndrome", syndrome, np.ndarray, optional=False)
        if not len(syndrome) == self.m:
            raise ValueError(f"The syndrome must have length {self.m}. Not {len(syndrome)}.")
        vec = np.asarray(syndrome).astype(np.uint8)
        if not vec.any():  # pyx:118-123
            self._converge = True
            return np.zeros(self.n, dtype=syndrome.dtype)
        self._require_supported()
        dec, llr, it, cv = self._decode_osd(vec[None, :])
        self._remember(vec, dec[0], llr[0], it[0], cv[0])
        return dec[0].astype(syndrome.dtype)

    def _remember(self, synd_row, dec_row, llr_row, it, cv):
        """Scalar state after a row that ran BP (bpd.decoding / log_prob_ratios / iterations / converge and the osdD results,
        pyx:125-136, 236-299).  ``bp_decoding`` is BP's own hard decision whether or not it converged: the device hands back
        the OSD solution for an unconverged row, BP's decision is ``llr <= 0`` (bp.hpp:290) of the log-ratios it returns."""
        self._iterations = int(it)
        self._converge = bool(cv)
        if llr_row is not None:
            self._log_prob_ratios = np.asarray(llr_row).copy()
            self._bp_decoding = (self._log_prob_ratios <= 0).astype(np.uint8)
        elif self._converge:
            self._bp_decoding = np.asarray(dec_row, np.uint8).copy()
        if self._converge:
            self._bp_decoding = np.asarray(dec_row, np.uint8).copy()
            self._decoding = self._bp_decoding.copy()
        else:
            self._osdw_decoding = np.asarray(dec_row, np.uint8).copy()
            self._osd0_decoding = None  # evaluated on demand (osd0_decoding) when the order is > 0
            self._last_syndrome = np.asarray(synd_row, np.uint8).copy()

    def decode_batch(self, syndromes, want_log_prob_ratios: bool = True):
        """Every row through BP (+ OSD-0 where BP does not converge) in one call; row b == ``decode(syndromes[b])``."""
        if syndromes.ndim != 2 or syndromes.shape[1] != self.m:
            raise ValueError(f"The syndrome must have length {self.m}. Not {syndromes.shape[-1]}.")
        self._require_supported()
        from ldpc_amd.engine import _is_torch
        if _is_torch(syndromes):  # device tensors in, device tensors out (nothing crosses PCIe)
            eng = self._get_engine()
            eng.set_osd(self._osd_method, self._osd_order)
            dec, llr, it, cv = eng.decode_batch(syndromes, want_llr=want_log_prob_ratios, osd=True)
            self.osd_status_batch = eng.osd_status(int(syndromes.shape[0])) if hasattr(eng, "osd_status") else None
            zero = syndromes.any(dim=1).logical_not()
            if bool(zero.any()):
                dec[zero] = 0
                cv[zero] = 1
                it[zero] = 0
                if llr is not None:
                    llr[zero] = 0
            self.converge_batch, self.iter_batch, self.log_prob_ratios_batch = cv.bool(), it, llr
            return dec
        dtype = syndromes.dtype
        vec = np.ascontiguousarray(np.asarray(syndromes).astype(np.uint8))
        dec, llr, it, cv = self._decode_osd(vec, want_llr=want_log_prob_ratios)
        self.osd_status_batch = self._last_status
        zero = ~vec.any(axis=1)
        dec[zero] = 0
        cv[zero] = True
        it[zero] = 0
        if llr is not None:
            llr[zero] = 0.0
        self.converge_batch, self.iter_batch, self.log_prob_ratios_batch = cv, it, llr
        ran = np.flatnonzero(~zero)
        if len(ran):  # scalar members describe the last row that ran BP, as after the reference's per-row loop
            last = int(ran[-1])
            self._remember(vec[last], dec[last], None if llr is None else llr[last], it[last], cv[last])
        if len(vec):
            self._converge = bool(cv[-1])
        return dec.astype(dtype)

    # ---- results (pyx:236-299) --------------------------------------------------------------------
    @property
    def bp_decoding(self) -> np.ndarray:
        return np.array(self._bp_decoding).astype(int)

    @property
    def osd0_decoding(self) -> np.ndarray:
        if self._converge:
            return np.array(self._bp_decoding).astype(int)
        if self._osd0_decoding is None:
            higher = self._osd_method in (EXHAUSTIVE, COMBINATION_SWEEP) and self._osd_order > 0
            if not higher:  # order 0: osdw_decoding == osd0_decoding (osd.hpp:113)
                self._osd0_decoding = self._osdw_decoding.copy()
            else:  # the device returns the swept solution only; the OSD-0 one is recomputed when somebody asks
                self._osd0_decoding = self._decode_osd(self._last_syndrome[None, :], want_llr=False, force_osd0=True)[0][0].copy()
        return np.array(self._osd0_decoding).astype(int)

    @property
    def osdw_decoding(self) -> np.ndarray:
        return np.array(self._bp_decoding if self._converge else self._osdw_decoding).astype(int)

    @property
    def decoding(self) -> np.ndarray:
        # the reference's property reads a misspelt member (`self.osD`, pyx:247) and raises AttributeError;
        # the intended value is osdw_decoding
        return self.osdw_decoding
