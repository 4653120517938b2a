"""ctypes front-ends for the CHECKERS under oracle/ -- test infrastructure, not product code.

* ``BpOracle``  -> oracle/libbp_oracle.so  (C restatement of bp.hpp:192-325, oracle/bp_oracle.c)
* ``RefBp``     -> oracle/_ref/libref_bp.so (the real reference headers behind oracle/ref_harness.cpp;
  exists only where /root/reference was present at build time, or where the prebuilt .so travelled)

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this module.
Nothing under ``ldpc_amd/`` does (tests/test_layout.py enforces it).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libbp_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libref_bp.so")
REFERENCE_ROOT = "/root/reference"

_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(ref: bool | None = None) -> None:
    """(Re)build the checker libraries with oracle/Makefile; ``ref`` defaults to 'if the reference is here'."""
    targets = ["all"]
    if ref is None:
        ref = os.path.isdir(os.path.join(REFERENCE_ROOT, "src_cpp"))
    if ref:
        targets.append("ref")
    subprocess.run(["make", "-C", _HERE, *targets], check=True, capture_output=True)


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def csr_arrays(h):
    """(m, n, row_ptr int32, col_idx int32) of a binary matrix, columns sorted in each row."""
    h = sp.csr_matrix(h)
    h = h.copy()
    h.eliminate_zeros()
    h.sum_duplicates()
    h.sort_indices()
    m, n = h.shape
    return m, n, np.ascontiguousarray(h.indptr, np.int32), np.ascontiguousarray(h.indices, np.int32)


def _method_id(bp_method) -> int:
    s = str(bp_method).lower()
    if s in ("product_sum", "ps", "0", "prod_sum"):
        return 0
    if s in ("minimum_sum", "ms", "1", "min_sum"):
        return 1
    raise ValueError(bp_method)


def _probs(n, error_rate=None, error_channel=None):
    if error_channel is not None:
        p = np.ascontiguousarray(error_channel, np.float64)
        assert p.shape == (n,)
        return p
    return np.full(n, float(error_rate), np.float64)


class _OracleLib:
    _lib = None

    @classmethod
    def get(cls):
        if cls._lib is None:
            if not os.path.exists(ORACLE_SO):
                build(ref=False)
            lib = C.CDLL(ORACLE_SO)
            lib.bp_oracle_new.restype = C.c_void_p
            lib.bp_oracle_new.argtypes = [C.c_int, C.c_int, _i32p, _i32p]
            lib.bp_oracle_free.argtypes = [C.c_void_p]
            lib.bp_oracle_decode_batch.argtypes = [
                C.c_void_p, _f64p, C.c_int, C.c_int, C.c_double, _u8p, C.c_int64, _u8p,
                C.c_void_p, _i32p, _u8p]
            lib.oracle_gen_bsc_syndromes.argtypes = [
                C.c_int, C.c_int, _i32p, _i32p, C.c_uint64, C.c_uint64, C.c_int64, C.c_int64,
                _u8p, C.c_void_p]
            lib.bp_oracle_decode_serial_batch.argtypes = [
                C.c_void_p, _f64p, C.c_int, C.c_int, C.c_double, C.c_void_p, _u8p, C.c_int64, _u8p,
                C.c_void_p, _i32p, _u8p]
            lib.bp_oracle_decode_serial_relative_batch.argtypes = [
                C.c_void_p, _f64p, C.c_int, C.c_int, C.c_double, _i32p, C.c_int, _u8p, C.c_int64, _u8p, _f64p, _i32p, _u8p]
            lib.bp_oracle_decode_serial_orders_batch.argtypes = [
                C.c_void_p, _f64p, C.c_int, C.c_int, C.c_double, _i32p, C.c_int, _u8p, C.c_int64, _u8p, _f64p, _i32p, _u8p]
            lib.osd0_oracle.argtypes = [C.c_int, C.c_int, _i32p, _i32p, _f64p, _u8p, _u8p]
            lib.bposd0_oracle_decode_batch.argtypes = lib.bp_oracle_decode_batch.argtypes
            lib.bp_oracle_soft_info_decode_batch.argtypes = [
                C.c_void_p, _f64p, C.c_int, C.c_double, C.c_void_p, _f64p, C.c_int64, C.c_double, C.c_double, _u8p,
                C.c_void_p, _i32p, _u8p, C.c_void_p]
            lib.osdw_oracle.argtypes = [C.c_int, C.c_int, _i32p, _i32p, _f64p, _u8p, _f64p, C.c_int, C.c_int, _u8p, C.c_void_p]
            lib.bposdw_oracle_decode_batch.argtypes = [
                C.c_void_p, _f64p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, _u8p, C.c_int64, _u8p,
                C.c_void_p, _i32p, _u8p]
            lib.oracle_sm64.restype = C.c_uint64
            lib.oracle_sm64.argtypes = [C.c_uint64, C.c_uint64]
            cls._lib = lib
        return cls._lib


class BpOracle:
    """CPU restatement of the reference's parallel-schedule BP (one syndrome at a time)."""

    def __init__(self, h, error_rate=None, error_channel=None, max_iter=0, bp_method="product_sum",
                 ms_scaling_factor=1.0):
        self.lib = _OracleLib.get()
        self.m, self.n, self.row_ptr, self.col_idx = csr_arrays(h)
        self.channel_probs = _probs(self.n, error_rate, error_channel)
        self.max_iter = int(max_iter) if max_iter else self.n  # _bp_decoder.pyx:357
        self.method = _method_id(bp_method)
        self.alpha = float(ms_scaling_factor)
        self._h = self.lib.bp_oracle_new(self.m, self.n, self.row_ptr, self.col_idx)
        if not self._h:
            raise ValueError("bad CSR input")

    def __del__(self):
        if getattr(self, "_h", None):
            self.lib.bp_oracle_free(self._h)
            self._h = None

    def decode_batch(self, syndromes, want_llr=True):
        s = np.ascontiguousarray(syndromes, np.uint8)
        if s.ndim == 1:
            s = s[None, :]
        assert s.shape[1] == self.m
        b = s.shape[0]
        dec = np.zeros((b, self.n), np.uint8)
        llr = np.zeros((b, self.n), np.float64) if want_llr else None
        it = np.zeros(b, np.int32)
        conv = np.zeros(b, np.uint8)
        self.lib.bp_oracle_decode_batch(
            self._h, self.channel_probs, self.max_iter, self.method, self.alpha, s, b, dec,
            llr.ctypes.data if want_llr else None, it, conv)
        return dec, llr, it, conv.astype(bool)

    def decode_serial_batch(self, syndromes, order=None, want_llr=True):
        """Serial schedule with a fixed bit order (bp.hpp:451-545); ``order`` None = 0..n-1."""
        s = np.ascontiguousarray(syndromes, np.uint8)
        b = s.shape[0]
        dec = np.zeros((b, self.n), np.uint8)
        llr = np.zeros((b, self.n), np.float64) if want_llr else None
        it = np.zeros(b, np.int32)
        conv = np.zeros(b, np.uint8)
        od = None if order is None else np.ascontiguousarray(order, np.int32)
        self.lib.bp_oracle_decode_serial_batch(self._h, self.channel_probs, self.max_iter, self.method, self.alpha,
                                               od.ctypes.data if od is not None else None, s, b, dec,
                                               llr.ctypes.data if want_llr else None, it, conv)
        return dec, llr, it, conv.astype(bool)

    def decode_serial_relative_batch(self, syndromes, order_state=None, fresh=True):
        """schedule='serial_relative' (bp.hpp:469-483).  ``fresh``: every row on a new decoder object (order 0..n-1 or
        ``order_state``); else the rows run one after the other on one object.  Returns (..., final order)."""
        s = np.ascontiguousarray(syndromes, np.uint8)
        b = s.shape[0]
        dec = np.zeros((b, self.n), np.uint8)
        llr = np.zeros((b, self.n), np.float64)
        it = np.zeros(b, np.int32)
        conv = np.zeros(b, np.uint8)
        st = np.arange(self.n, dtype=np.int32) if order_state is None else np.ascontiguousarray(order_state, np.int32).copy()
        self.lib.bp_oracle_decode_serial_relative_batch(self._h, self.channel_probs, self.max_iter, self.method, self.alpha, st,
                                                        1 if fresh else 0, s, b, dec, llr, it, conv)
        return dec, llr, it, conv.astype(bool), st

    def decode_random_serial_batch(self, syndromes, seed):
        """random_serial_schedule=True with random_schedule_seed=``seed`` (bp.hpp:467-468), a new decoder object per row:
        iteration t of every row walks the order after t shuffles of 0..n-1 (std::shuffle on std::mt19937, shuffle_helper.cpp)."""
        s = np.ascontiguousarray(syndromes, np.uint8)
        b = s.shape[0]
        orders = shuffle_orders(seed, self.n, self.max_iter)[0]
        dec = np.zeros((b, self.n), np.uint8)
        llr = np.zeros((b, self.n), np.float64)
        it = np.zeros(b, np.int32)
        conv = np.zeros(b, np.uint8)
        self.lib.bp_oracle_decode_serial_orders_batch(self._h, self.channel_probs, self.max_iter, self.method, self.alpha,
                                                      np.ascontiguousarray(orders.reshape(-1)), self.max_iter, s, b, dec, llr, it, conv)
        return dec, llr, it, conv.astype(bool)

    def soft_info_decode_random_batch(self, soft_syndromes, cutoff, sigma, seed, order=None):
        """soft_info_decode_serial with random_serial_schedule (bp.hpp:573-577), every row from the same starting ``order``
        (default 0..n-1: a new decoder object per row): iteration t walks the order after t reseeded shuffles."""
        s = np.ascontiguousarray(soft_syndromes, np.float64)
        b = s.shape[0]
        orders = shuffle_orders_reseeded(seed, self.n, self.max_iter, order)[0]
        dec = np.zeros((b, self.n), np.uint8)
        llr = np.zeros((b, self.n), np.float64)
        it = np.zeros(b, np.int32)
        conv = np.zeros(b, np.uint8)
        soft = np.zeros((b, self.m), np.float64)
        self.lib.bp_oracle_soft_info_decode_orders_batch.argtypes = [C.c_void_p, _f64p, C.c_int, C.c_double, _i32p, C.c_int, _f64p,
                                                                     C.c_int64, C.c_double, C.c_double, _u8p, C.c_void_p, _i32p,
                                                                     _u8p, C.c_void_p]
        self.lib.bp_oracle_soft_info_decode_orders_batch(self._h, self.channel_probs, self.max_iter, self.alpha,
                                                         np.ascontiguousarray(orders.reshape(-1)), self.max_iter, s, b,
                                                         float(cutoff), float(sigma), dec, llr.ctypes.data, it, conv, soft.ctypes.data)
        return dec, llr, it, conv.astype(bool), soft

    def osd0(self, syndrome, llr):
        """OSD-0 alone (osd.hpp:110-117 restated) on one syndrome and one vector of log-ratios."""
        out = np.zeros(self.n, np.uint8)
        self.lib.osd0_oracle(self.m, self.n, self.row_ptr, self.col_idx, np.ascontiguousarray(llr, np.float64),
                             np.ascontiguousarray(syndrome, np.uint8), out)
        return out

    def bposd0_decode_batch(self, syndromes, want_llr=True):
        """BpOsdDecoder.decode per row (_bposd_decoder.pyx:125-134): BP, then OSD-0 where BP did not converge."""
        s = np.ascontiguousarray(syndromes, np.uint8)
        b = s.shape[0]
        dec = np.zeros((b, self.n), np.uint8)
        llr = np.zeros((b, self.n), np.float64) if want_llr else None
        it = np.zeros(b, np.int32)
        conv = np.zeros(b, np.uint8)
        self.lib.bposd0_oracle_decode_batch(self._h, self.channel_probs, self.max_iter, self.method, self.alpha, s, b,
                                            dec, llr.ctypes.data if want_llr else None, it, conv)
        return dec, llr, it, conv.astype(bool)

    def soft_info_decode_batch(self, soft_syndromes, cutoff, sigma, order=None):
        """soft_info_decode_serial (bp.hpp:547-660) per row -> (decoding, llr, iterations, converge, soft_syndrome)."""
        s = np.ascontiguousarray(soft_syndromes, np.float64)
        b = s.shape[0]
        dec = np.zeros((b, self.n), np.uint8)
        llr = np.zeros((b, self.n), np.float64)
        it = np.zeros(b, np.int32)
        conv = np.zeros(b, np.uint8)
        soft = np.zeros((b, self.m), np.float64)
        od = None if order is None else np.ascontiguousarray(order, np.int32)
        self.lib.bp_oracle_soft_info_decode_batch(self._h, self.channel_probs, self.max_iter, self.alpha,
                                                  od.ctypes.data if od is not None else None, s, b, float(cutoff), float(sigma),
                                                  dec, llr.ctypes.data, it, conv, soft.ctypes.data)
        return dec, llr, it, conv.astype(bool), soft

    def osdw(self, syndrome, llr, osd_method, osd_order, channel_probs=None):
        """OSD of any order alone (osd.hpp:103-187 restated): ``osd_method`` 1 OSD_0, 2 OSD_E, 3 OSD_CS.
        Returns (osdw_decoding, osd0_decoding)."""
        out = np.zeros(self.n, np.uint8)
        out0 = np.zeros(self.n, np.uint8)
        probs = self.channel_probs if channel_probs is None else np.ascontiguousarray(channel_probs, np.float64)
        self.lib.osdw_oracle(self.m, self.n, self.row_ptr, self.col_idx, np.ascontiguousarray(llr, np.float64),
                             np.ascontiguousarray(syndrome, np.uint8), probs, int(osd_method), int(osd_order), out,
                             out0.ctypes.data)
        return out, out0

    def bposd_decode_batch(self, syndromes, osd_method, osd_order, want_llr=True):
        """BpOsdDecoder.decode per row with any OSD method / order."""
        s = np.ascontiguousarray(syndromes, np.uint8)
        b = s.shape[0]
        dec = np.zeros((b, self.n), np.uint8)
        llr = np.zeros((b, self.n), np.float64) if want_llr else None
        it = np.zeros(b, np.int32)
        conv = np.zeros(b, np.uint8)
        self.lib.bposdw_oracle_decode_batch(self._h, self.channel_probs, self.max_iter, self.method, self.alpha,
                                            int(osd_method), int(osd_order), s, b, dec,
                                            llr.ctypes.data if want_llr else None, it, conv)
        return dec, llr, it, conv.astype(bool)

    def gen_bsc_syndromes(self, seed, p, shot0, shots, want_errors=False):
        from ldpc_amd.prng import bernoulli_threshold
        synd = np.zeros((shots, self.m), np.uint8)
        err = np.zeros((shots, self.n), np.uint8) if want_errors else None
        self.lib.oracle_gen_bsc_syndromes(
            self.m, self.n, self.row_ptr, self.col_idx, seed, bernoulli_threshold(p), shot0, shots,
            synd, err.ctypes.data if want_errors else None)
        return (synd, err) if want_errors else synd


def shuffle_orders(seed, n, count, order=None, state=b""):
    """``count`` successive std::shuffle results of ``order`` (default 0..n-1) on std::mt19937(seed) -> (orders [count][n],
    final order, generator state) through oracle/liboracle_shuffle.so (the host's own C++ standard library)."""
    so = os.path.join(_HERE, "liboracle_shuffle.so")
    if not os.path.exists(so):
        build(ref=False)
    lib = C.CDLL(so)
    lib.oracle_shuffle_orders.argtypes = [C.c_uint32, C.c_int32, _i32p, C.c_int32, _i32p, C.c_char_p]
    cur = np.arange(n, dtype=np.int32) if order is None else np.ascontiguousarray(order, np.int32).copy()
    out = np.zeros((max(count, 1), n), np.int32)
    buf = C.create_string_buffer(bytes(state), 16384)
    lib.oracle_shuffle_orders(int(seed), int(n), cur, int(count), out.reshape(-1), buf)
    return out[:count], cur, buf.value


def shuffle_orders_reseeded(seed, n, count, order=None):
    """The soft-syndrome routine's random schedule (bp.hpp:573-577): ``count`` successive
    ``std::shuffle(order, std::default_random_engine(seed))`` -- a new engine from the same seed each time -> (orders [count][n],
    final order).  Through oracle/liboracle_shuffle.so."""
    so = os.path.join(_HERE, "liboracle_shuffle.so")
    if not os.path.exists(so):
        build(ref=False)
    lib = C.CDLL(so)
    lib.oracle_shuffle_orders_reseeded.argtypes = [C.c_int32, C.c_int32, _i32p, C.c_int32, _i32p]
    cur = np.arange(n, dtype=np.int32) if order is None else np.ascontiguousarray(order, np.int32).copy()
    out = np.zeros((max(count, 1), n), np.int32)
    lib.oracle_shuffle_orders_reseeded(int(seed), int(n), cur, int(count), out.reshape(-1))
    return out[:count], cur


def ref_soft_random(h, soft_syndromes, cutoff, sigma, *, error_rate=None, error_channel=None, max_iter=0, ms_scaling_factor=1.0,
                    seed=0, fresh=True):
    """The REAL soft_info_decode_serial with random_serial_schedule: a new decoder object per row (``fresh``) or one for all
    rows.  Returns (decoding, llr, iterations, converge, soft syndrome, final order per row)."""
    if not have_ref():
        raise RuntimeError("oracle/_ref/libref_bp.so not built (needs /root/reference: make -C oracle ref)")
    lib = C.CDLL(REF_SO)
    fn = lib.ref_bp_soft_random_batch
    fn.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f64p, C.c_int, C.c_double, C.c_int, C.c_int, _f64p, C.c_int64,
                   C.c_double, C.c_double, _u8p, _f64p, _i32p, _u8p, _f64p, _i32p]
    m, n, row_ptr, col_idx = csr_arrays(h)
    rows = np.ascontiguousarray(np.repeat(np.arange(m, dtype=np.int32), np.diff(row_ptr)).astype(np.int32))
    probs = _probs(n, error_rate, error_channel)
    s = np.ascontiguousarray(soft_syndromes, np.float64)
    b = s.shape[0]
    dec = np.zeros((b, n), np.uint8)
    llr = np.zeros((b, n), np.float64)
    it = np.zeros(b, np.int32)
    conv = np.zeros(b, np.uint8)
    soft = np.zeros((b, m), np.float64)
    final = np.zeros((b, n), np.int32)
    fn(m, n, len(col_idx), rows, col_idx, probs, int(max_iter) if max_iter else n, float(ms_scaling_factor), int(seed),
       1 if fresh else 0, s, b, float(cutoff), float(sigma), dec, llr, it, conv, soft, final.reshape(-1))
    return dec, llr, it, conv.astype(bool), soft, final


def ref_decode_stateful(h, syndromes, *, schedule, error_rate=None, error_channel=None, max_iter=0, bp_method="product_sum",
                        ms_scaling_factor=1.0, random_serial=False, seed=0, fresh=True, order0=None):
    """The REAL reference with a schedule that keeps state in the decoder object: a new object per row (``fresh``) or one
    object for all rows; ``order0``: the ``serial_schedule_order`` handed to the constructor (any n bit numbers).
    Returns (decoding, llr, iterations, converge, final order per row)."""
    if not have_ref():
        raise RuntimeError("oracle/_ref/libref_bp.so not built (needs /root/reference: make -C oracle ref)")
    lib = C.CDLL(REF_SO)
    if order0 is not None:
        if random_serial:
            raise ValueError("the reference refuses a fixed order together with the random schedule (bp.hpp:112-114)")
        fo = lib.ref_bp_decode_batch_order
        fo.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f64p, C.c_int, C.c_int, C.c_int, C.c_double, _i32p, C.c_int,
                       _u8p, C.c_int64, _u8p, _f64p, _i32p, _u8p, _i32p]
        m, n, row_ptr, col_idx = csr_arrays(h)
        rows = np.ascontiguousarray(np.repeat(np.arange(m, dtype=np.int32), np.diff(row_ptr)).astype(np.int32))
        probs = _probs(n, error_rate, error_channel)
        s = np.ascontiguousarray(syndromes, np.uint8)
        b = s.shape[0]
        dec = np.zeros((b, n), np.uint8)
        llr = np.zeros((b, n), np.float64)
        it = np.zeros(b, np.int32)
        conv = np.zeros(b, np.uint8)
        final = np.zeros((b, n), np.int32)
        fo(m, n, len(col_idx), rows, col_idx, probs, int(max_iter) if max_iter else n, _method_id(bp_method), RefBp.SCHEDULE[schedule],
           float(ms_scaling_factor), np.ascontiguousarray(order0, np.int32), 0 if fresh else 1, s, b, dec, llr, it, conv, final.reshape(-1))
        return dec, llr, it, conv.astype(bool), final
    fn = lib.ref_bp_decode_fresh_batch if fresh else lib.ref_bp_decode_carried_batch
    fn.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f64p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                   _u8p, C.c_int64, _u8p, _f64p, _i32p, _u8p, _i32p]
    m, n, row_ptr, col_idx = csr_arrays(h)
    rows = np.ascontiguousarray(np.repeat(np.arange(m, dtype=np.int32), np.diff(row_ptr)).astype(np.int32))
    probs = _probs(n, error_rate, error_channel)
    s = np.ascontiguousarray(syndromes, np.uint8)
    b = s.shape[0]
    dec = np.zeros((b, n), np.uint8)
    llr = np.zeros((b, n), np.float64)
    it = np.zeros(b, np.int32)
    conv = np.zeros(b, np.uint8)
    final = np.zeros((b, n), np.int32)
    fn(m, n, len(col_idx), rows, col_idx, probs, int(max_iter) if max_iter else n, _method_id(bp_method),
       RefBp.SCHEDULE[schedule], float(ms_scaling_factor), 1 if random_serial else 0, int(seed), s, b, dec, llr, it, conv,
       final.reshape(-1))
    return dec, llr, it, conv.astype(bool), final


class RefBp:
    """The real reference ``ldpc::bp::BpDecoder`` (bp.hpp:51-666) behind oracle/ref_harness.cpp."""

    SCHEDULE = {"serial": 0, "parallel": 1, "serial_relative": 2}  # bp.hpp:28-32
    INPUT = {"syndrome": 0, "received_vector": 1, "auto": 2}  # bp.hpp:34-38

    def __init__(self, h, error_rate=None, error_channel=None, max_iter=0, bp_method="product_sum",
                 ms_scaling_factor=1.0, schedule="parallel", input_vector_type="syndrome"):
        if not have_ref():
            raise RuntimeError("oracle/_ref/libref_bp.so not built (needs /root/reference: make -C oracle ref)")
        lib = C.CDLL(REF_SO)
        lib.ref_bp_new.restype = C.c_void_p
        lib.ref_bp_new.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f64p, C.c_int, C.c_int,
                                   C.c_int, C.c_double, C.c_int]
        lib.ref_bp_free.argtypes = [C.c_void_p]
        lib.ref_bp_set_channel.argtypes = [C.c_void_p, _f64p]
        lib.ref_bp_decode_batch.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int64, _u8p, C.c_void_p,
                                            _i32p, _u8p]
        lib.ref_bp_mulvec.argtypes = [C.c_void_p, _u8p, _u8p]
        lib.ref_bp_set_serial_order.argtypes = [C.c_void_p, _i32p]
        self.lib = lib
        self.m, self.n, row_ptr, col_idx = csr_arrays(h)
        rows = np.repeat(np.arange(self.m, dtype=np.int32), np.diff(row_ptr)).astype(np.int32)
        self.channel_probs = _probs(self.n, error_rate, error_channel)
        self.max_iter = int(max_iter) if max_iter else self.n
        self._h = lib.ref_bp_new(self.m, self.n, len(col_idx), np.ascontiguousarray(rows), col_idx,
                                 self.channel_probs, self.max_iter, _method_id(bp_method),
                                 self.SCHEDULE[schedule], float(ms_scaling_factor),
                                 self.INPUT[input_vector_type])

    def __del__(self):
        if getattr(self, "_h", None):
            self.lib.ref_bp_free(self._h)
            self._h = None

    def set_channel(self, probs):
        self.channel_probs = np.ascontiguousarray(probs, np.float64)
        self.lib.ref_bp_set_channel(self._h, self.channel_probs)

    def set_serial_order(self, order):
        self.lib.ref_bp_set_serial_order(self._h, np.ascontiguousarray(order, np.int32))

    def decode_batch(self, inputs, want_llr=True):
        s = np.ascontiguousarray(inputs, np.uint8)
        if s.ndim == 1:
            s = s[None, :]
        b, ln = s.shape
        dec = np.zeros((b, self.n), np.uint8)
        llr = np.zeros((b, self.n), np.float64) if want_llr else None
        it = np.zeros(b, np.int32)
        conv = np.zeros(b, np.uint8)
        self.lib.ref_bp_decode_batch(self._h, s, ln, b, dec, llr.ctypes.data if want_llr else None, it, conv)
        return dec, llr, it, conv.astype(bool)

    def mulvec(self, v):
        out = np.zeros(self.m, np.uint8)
        self.lib.ref_bp_mulvec(self._h, np.ascontiguousarray(v, np.uint8), out)
        return out

    def soft_info_decode_batch(self, soft_syndromes, cutoff, sigma):
        """The real soft_info_decode_serial (bp.hpp:547-660); construct with schedule='serial', bp_method='minimum_sum'."""
        s = np.ascontiguousarray(soft_syndromes, np.float64)
        b = s.shape[0]
        dec = np.zeros((b, self.n), np.uint8)
        llr = np.zeros((b, self.n), np.float64)
        it = np.zeros(b, np.int32)
        conv = np.zeros(b, np.uint8)
        soft = np.zeros((b, self.m), np.float64)
        self.lib.ref_bp_soft_info_decode_batch.argtypes = [C.c_void_p, _f64p, C.c_int64, C.c_double, C.c_double, _u8p, _f64p,
                                                           _i32p, _u8p, _f64p]
        self.lib.ref_bp_soft_info_decode_batch(self._h, s, b, float(cutoff), float(sigma), dec, llr, it, conv, soft)
        return dec, llr, it, conv.astype(bool), soft


class RefBpOsd:
    """The real reference BP + ``ldpc::osd::OsdDecoder`` (OSD_0) behind oracle/ref_harness.cpp."""

    def __init__(self, h, error_rate=None, error_channel=None, max_iter=0, bp_method="product_sum", ms_scaling_factor=1.0,
                 osd_method=1, osd_order=0, schedule="parallel"):
        if not have_ref():
            raise RuntimeError("oracle/_ref/libref_bp.so not built (needs /root/reference: make -C oracle ref)")
        lib = C.CDLL(REF_SO)
        lib.ref_bposd_set_osd.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.ref_bposd_k.argtypes = [C.c_void_p]
        lib.ref_bposd_new.restype = C.c_void_p
        lib.ref_bposd_new.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f64p, C.c_int, C.c_int, C.c_double]
        lib.ref_bposd_free.argtypes = [C.c_void_p]
        lib.ref_bposd_decode_batch.argtypes = [C.c_void_p, _u8p, C.c_int64, _u8p, C.c_void_p, _i32p, _u8p]
        lib.ref_osd0.argtypes = [C.c_void_p, _u8p, _f64p, _u8p]
        self.lib = lib
        self.m, self.n, row_ptr, col_idx = csr_arrays(h)
        rows = np.repeat(np.arange(self.m, dtype=np.int32), np.diff(row_ptr)).astype(np.int32)
        self.channel_probs = _probs(self.n, error_rate, error_channel)
        self.max_iter = int(max_iter) if max_iter else self.n
        lib.ref_bposd_new2.restype = C.c_void_p
        lib.ref_bposd_new2.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f64p, C.c_int, C.c_int, C.c_double, C.c_int]
        self._h = lib.ref_bposd_new2(self.m, self.n, len(col_idx), np.ascontiguousarray(rows), col_idx, self.channel_probs,
                                     self.max_iter, _method_id(bp_method), float(ms_scaling_factor),
                                     {"parallel": 1, "serial": 0}[schedule])
        if (int(osd_method), int(osd_order)) != (1, 0):
            lib.ref_bposd_set_osd(self._h, int(osd_method), int(osd_order))
        self.k = int(lib.ref_bposd_k(self._h))

    def __del__(self):
        if getattr(self, "_h", None):
            self.lib.ref_bposd_free(self._h)
            self._h = None

    def decode_batch(self, syndromes, want_llr=True):
        s = np.ascontiguousarray(syndromes, np.uint8)
        b = s.shape[0]
        dec = np.zeros((b, self.n), np.uint8)
        llr = np.zeros((b, self.n), np.float64) if want_llr else None
        it = np.zeros(b, np.int32)
        conv = np.zeros(b, np.uint8)
        self.lib.ref_bposd_decode_batch(self._h, s, b, dec, llr.ctypes.data if want_llr else None, it, conv)
        return dec, llr, it, conv.astype(bool)

    def osd0(self, syndrome, llr):
        out = np.zeros(self.n, np.uint8)
        self.lib.ref_osd0(self._h, np.ascontiguousarray(syndrome, np.uint8), np.ascontiguousarray(llr, np.float64), out)
        return out


def llr_close(got, want, rtol=1e-5) -> bool:
    """north_star tolerance: LLRs within ``rtol`` RELATIVE of the reference; non-finite entries must
    match in kind (NaN with NaN, +inf with +inf, -inf with -inf)."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    fin = np.isfinite(want)
    if not np.array_equal(fin, np.isfinite(got)) or not np.array_equal(np.isnan(want), np.isnan(got)):
        return False
    inf = np.isinf(want)
    if not np.array_equal(np.sign(want[inf]), np.sign(got[inf])):
        return False
    return bool(np.all(np.abs(got[fin] - want[fin]) <= rtol * np.abs(want[fin])))


def bits_equal(a, b) -> bool:
    """Bit-identical float64 arrays, except that any NaN matches any NaN (the sign/payload of a NaN
    is not a result: it depends on operand order inside inf - inf)."""
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    if a.shape != b.shape:
        return False
    same = a.view(np.uint64) == b.view(np.uint64)
    return bool(np.all(same | (np.isnan(a) & np.isnan(b))))
