# cython: language_level=3, boundscheck=False, wraparound=False
# distutils: language = c++
"""Cython binding of the C++ host class ``ldpc_hip::BpDecoder`` (include/ldpc_hip.hpp) -- the same kind of
binding the reference uses for ``ldpc::bp::BpDecoder`` (src_python/ldpc/bp_decoder/_bp_decoder.pxd:47-83,
_bp_decoder.pyx:132): a ``cdef cppclass`` declaration with the reference's member names and a ``cdef class``
that reads/writes those members.  ``CyBpCore`` is deliberately thin (construction, member access, decode,
decode_batch); the user-facing validation/alias layer is ``ldpc_amd.bp_decoder.BpDecoder``, which can run on
either this binding or the ctypes engine (``ldpc_amd/engine.py``) -- both drive the same C ABI.
"""
from libc.stdint cimport int32_t, int64_t, uint8_t
from libcpp cimport bool as cbool
from libcpp.vector cimport vector
from libcpp.string cimport string

import numpy as np

cdef extern from "ldpc_hip.hpp" namespace "ldpc_hip":
    cdef enum BpMethod:
        PRODUCT_SUM
        MINIMUM_SUM

    cdef cppclass BpDecoderCpp "ldpc_hip::BpDecoder":
        BpDecoderCpp(int m, int n, vector[int32_t]& csr_row_ptr, vector[int32_t]& csr_col_idx,
                     vector[double] channel_probs, int max_iter, BpMethod method,
                     double min_sum_scaling_factor, int device) except +
        vector[double] channel_probabilities
        int check_count
        int bit_count
        int maximum_iterations
        BpMethod bp_method
        double ms_scaling_factor
        vector[uint8_t] decoding
        vector[double] log_prob_ratios
        int iterations
        cbool converge
        vector[uint8_t] decoding_batch
        vector[double] log_prob_ratios_batch
        vector[int32_t] iterations_batch
        vector[uint8_t] converge_batch
        vector[uint8_t] osd_status_batch
        int last_status
        string last_error
        int osd_method
        int osd_order
        vector[uint8_t]& decode(vector[uint8_t]& syndrome)
        cbool decode_batch(const uint8_t *syndromes, int64_t batch, cbool want_llr, cbool osd0) nogil
        cbool decode_batch_into(const uint8_t *syndromes, int64_t batch, uint8_t *dec, double *llr, int32_t *iters, uint8_t *conv, cbool osd0) nogil


cdef class CyBpCore:
    cdef BpDecoderCpp *bpd
    cdef int m, n
    cdef public object osd_status  # (B,) uint8 after decode_batch(..., osd0=True): see ldpc_hip_bposd_get_status

    def __cinit__(self, row_ptr, col_idx, int n, channel_probs, int max_iter, int bp_method, double ms_scaling_factor,
                  int device=-1):
        cdef vector[int32_t] rp = np.ascontiguousarray(row_ptr, dtype=np.int32).tolist()
        cdef vector[int32_t] ci = np.ascontiguousarray(col_idx, dtype=np.int32).tolist()
        cdef vector[double] probs = np.ascontiguousarray(channel_probs, dtype=np.float64).tolist()
        self.m = <int>rp.size() - 1
        self.n = n
        self.bpd = new BpDecoderCpp(self.m, n, rp, ci, probs, max_iter,
                                    MINIMUM_SUM if bp_method == 1 else PRODUCT_SUM, ms_scaling_factor, device)

    def __dealloc__(self):
        if self.bpd != NULL:
            del self.bpd

    # ---- members, as the reference's property layer accesses them (pyx:167-357) ----
    @property
    def channel_probabilities(self):
        return np.array(self.bpd.channel_probabilities)

    @channel_probabilities.setter
    def channel_probabilities(self, value):
        cdef int i
        if len(value) != self.n:
            raise ValueError(f"The error channel vector must have length {self.n}, not {len(value)}.")
        for i in range(self.n):
            self.bpd.channel_probabilities[i] = value[i]

    @property
    def maximum_iterations(self):
        return self.bpd.maximum_iterations

    @maximum_iterations.setter
    def maximum_iterations(self, int value):
        self.bpd.maximum_iterations = value

    @property
    def bp_method(self):
        return <int>self.bpd.bp_method

    @bp_method.setter
    def bp_method(self, int value):
        self.bpd.bp_method = MINIMUM_SUM if value == 1 else PRODUCT_SUM

    @property
    def ms_scaling_factor(self):
        return self.bpd.ms_scaling_factor

    @ms_scaling_factor.setter
    def ms_scaling_factor(self, double value):
        self.bpd.ms_scaling_factor = value

    @property
    def converge(self):
        return self.bpd.converge

    @property
    def iterations(self):
        return self.bpd.iterations

    @property
    def log_prob_ratios(self):
        return np.array(self.bpd.log_prob_ratios)

    @property
    def decoding(self):
        return np.array(self.bpd.decoding, dtype=np.uint8)

    @property
    def osd_method(self):
        return self.bpd.osd_method

    @osd_method.setter
    def osd_method(self, int value):
        self.bpd.osd_method = value

    @property
    def osd_order(self):
        return self.bpd.osd_order

    @osd_order.setter
    def osd_order(self, int value):
        self.bpd.osd_order = value

    # ---- data path ----
    def decode(self, syndrome):
        """One syndrome (``BpDecoderCpp.decode``, reference pyx:682 / bp.hpp:159-190)."""
        cdef vector[uint8_t] s = np.ascontiguousarray(syndrome, dtype=np.uint8).tolist()
        self.bpd.decode(s)
        if self.bpd.last_status != 0:
            raise RuntimeError(self.bpd.last_error.decode("utf-8", "replace"))
        return np.array(self.bpd.decoding, dtype=np.uint8)

    def decode_batch(self, const uint8_t[:, ::1] syndromes, bint want_llr=True, bint osd0=False, llr_out=None):
        """``(B, m)`` uint8 -> ``(decoding (B, n), llr (B, n) | None, iterations (B,), converge (B,) bool)``."""
        cdef int64_t b = syndromes.shape[0]
        cdef cbool ok
        cdef cbool c_llr = want_llr, c_osd = osd0
        if syndromes.shape[1] != self.m:
            raise ValueError(f"The input_vector must have length {self.m} (for syndrome decoding). Not length {syndromes.shape[1]}.")
        if b == 0:
            return (np.zeros((0, self.n), np.uint8), np.zeros((0, self.n)) if want_llr else None,
                    np.zeros(0, np.int32), np.zeros(0, bool))
        # the results go straight into the arrays that are handed out (no intermediate C++ vectors: at 65 536 x 10 000 those were
        # 6 GB zero-filled and copied once more)
        dec = np.empty((b, self.n), np.uint8)
        if want_llr and llr_out is not None:  # the caller's array for the log-ratios (float64, C-contiguous, (B, n))
            if llr_out.dtype != np.float64 or llr_out.shape != (b, self.n) or not llr_out.flags.c_contiguous:
                raise ValueError(f"llr_out must be a C-contiguous float64 array of shape ({b}, {self.n})")
            llr = llr_out
        else:
            llr = np.empty((b, self.n), np.float64) if want_llr else None
        it = np.empty(b, np.int32)
        cv = np.empty(b, np.uint8)
        cdef uint8_t[:, ::1] dec_view = dec
        cdef double[:, ::1] llr_view
        cdef int32_t[::1] it_view = it
        cdef uint8_t[::1] cv_view = cv
        cdef uint8_t dummy = 0
        cdef uint8_t *dec_p = &dummy  # (n = 0: the C ABI wants a non-null pointer; nothing is written)
        cdef double *llr_p = NULL
        if self.n > 0:
            dec_p = &dec_view[0, 0]
            if want_llr:
                llr_view = llr
                llr_p = &llr_view[0, 0]
        with nogil:
            ok = self.bpd.decode_batch_into(&syndromes[0, 0], b, dec_p, llr_p, &it_view[0], &cv_view[0], c_osd)
        if not ok:
            raise RuntimeError(self.bpd.last_error.decode("utf-8", "replace"))
        self.osd_status = None
        if osd0 and self.bpd.osd_status_batch.size() == <size_t>b:
            self.osd_status = np.empty(b, np.uint8)
            cv_view = <uint8_t[:b]> &self.bpd.osd_status_batch[0]
            self.osd_status[:] = cv_view
        return dec, llr, it, cv.astype(bool)
