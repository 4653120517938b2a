"""``HipBpEngine``: Python owner of one ``ldpc_hip_bp`` handle (include/ldpc_hip.h).

It plays the role ``BpDecoderCpp *bpd`` plays inside the reference's Cython class
(_bp_decoder.pxd:85-93): the object the user-facing ``BpDecoder`` forwards to.  Arrays cross the
boundary as raw pointers: NumPy arrays as host pointers (the library stages them through HBM),
torch CUDA tensors as device pointers (no copy; outputs stay resident in HBM).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ldpc_amd import _lib

PRODUCT_SUM = 0  # ldpc::bp::BpMethod, bp.hpp:23-26
MINIMUM_SUM = 1


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class HipBpEngine:
    def __init__(self, row_ptr, col_idx, n, channel_probs, max_iter, bp_method, ms_scaling_factor,
                 device: int = -1):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.device = int(device)
        row_ptr = np.ascontiguousarray(row_ptr, np.int32)
        col_idx = np.ascontiguousarray(col_idx, np.int32)
        probs = np.ascontiguousarray(channel_probs, np.float64)
        self.m = int(len(row_ptr) - 1)
        self.n = int(n)
        self.nnz = int(len(col_idx))
        if probs.shape != (self.n,):
            raise ValueError("Channel probabilities vector must have length equal to the number of bits")
        desc = _lib.BpDesc(
            m=self.m, n=self.n, nnz=self.nnz,
            csr_row_ptr=row_ptr.ctypes.data_as(C.POINTER(C.c_int32)),
            csr_col_idx=col_idx.ctypes.data_as(C.POINTER(C.c_int32)),
            channel_probs=probs.ctypes.data_as(C.POINTER(C.c_double)),
            max_iter=int(max_iter), bp_method=int(bp_method),
            ms_scaling_factor=float(ms_scaling_factor), device=int(device))
        _lib.check(self._lib.ldpc_hip_bp_create(C.byref(desc), C.byref(self._h)))

    @classmethod
    def _view(cls, handle, m, n, nnz, device):
        """Non-owning engine over a handle that belongs to an ``ldpc_hip_bp_multi`` (``HipBpMultiEngine``)."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self._h = C.c_void_p(handle)
        self._borrowed = True
        self.device, self.m, self.n, self.nnz = int(device), int(m), int(n), int(nnz)
        return self

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            if not getattr(self, "_borrowed", False):
                self._lib.ldpc_hip_bp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- parameters -------------------------------------------------------------------------------
    def set_channel(self, channel_probs):
        p = np.ascontiguousarray(channel_probs, np.float64)
        _lib.check(self._lib.ldpc_hip_bp_set_channel(self._h, p.ctypes.data_as(C.POINTER(C.c_double)), len(p)))

    def set_params(self, max_iter, bp_method, ms_scaling_factor):
        _lib.check(self._lib.ldpc_hip_bp_set_params(self._h, int(max_iter), int(bp_method), float(ms_scaling_factor)))

    def set_schedule(self, schedule, serial_schedule_order=None):
        """``'parallel'`` (1) or ``'serial'`` (0, fixed order; ``serial_schedule_order`` of n ints or None)."""
        code = {"parallel": 1, "serial": 0, "serial_relative": 2, 0: 0, 1: 1, 2: 2}[schedule]
        if serial_schedule_order is None:
            _lib.check(self._lib.ldpc_hip_bp_set_schedule(self._h, code, None))
        else:
            order = np.ascontiguousarray(serial_schedule_order, np.int32)
            if order.shape != (self.n,):
                raise ValueError("serial_schedule_order must have length n")
            _lib.check(self._lib.ldpc_hip_bp_set_schedule(self._h, code, order.ctypes.data))

    def set_random_serial(self, enable, seed=0):
        """Random serial schedule (``random_serial_schedule`` / ``random_schedule_seed``): shuffle the order before every iteration."""
        _lib.check(self._lib.ldpc_hip_bp_set_random_serial(self._h, int(bool(enable)), int(seed) & 0xFFFFFFFF))

    def schedule_order(self):
        """The handle's current ``serial_schedule_order`` (what serial_relative / the random schedule left behind)."""
        out = np.zeros(self.n, np.int32)
        _lib.check(self._lib.ldpc_hip_bp_get_schedule_order(self._h, out.ctypes.data))
        return out

    def set_stream(self, stream_ptr):
        """``None`` -> the handle's own stream; ``0`` -> the legacy default stream (torch's stream 0); else a hipStream_t."""
        if stream_ptr is None:
            ptr = 0
        elif stream_ptr == 0:
            ptr = 1  # LDPC_HIP_STREAM_LEGACY_DEFAULT
        else:
            ptr = stream_ptr
        _lib.check(self._lib.ldpc_hip_bp_set_stream(self._h, C.c_void_p(ptr)))

    def set_tuning(self, waves_per_workgroup=0, max_chunk_tiles=0):
        _lib.check(self._lib.ldpc_hip_bp_set_tuning(self._h, int(waves_per_workgroup), int(max_chunk_tiles)))

    def set_math(self, mode):
        """'libm_exact' (default, bit-identical LLRs) or 'fast' (~1 ulp); see include/ldpc_hip.h."""
        code = {"libm_exact": 0, "exact": 0, 0: 0, "fast": 1, 1: 1}[mode]
        _lib.check(self._lib.ldpc_hip_bp_set_math(self._h, code))

    def set_ring(self, depth):
        """LDS-DMA ring for regular-degree matrices: False/0 = off, True/1 = default depth, 2 or 3 = slots per wave."""
        _lib.check(self._lib.ldpc_hip_bp_set_ring(self._h, int(depth)))

    def set_handoff(self, threshold_tiles):
        """Straggler hand-off of the streaming kernel: -1 automatic (default: 256 tiles; product-sum batches on matrices without a ring variant
        run as per-pass launches from the first iteration), 0 off, k = park when <= k tiles run."""
        _lib.check(self._lib.ldpc_hip_bp_set_handoff(self._h, int(threshold_tiles)))

    def set_osd(self, osd_method, osd_order):
        """OSD method / order for ``decode_batch(osd=True)``: 0 off, 1 OSD_0, 2 OSD_E, 3 OSD_CS (osd.hpp:18-23)."""
        _lib.check(self._lib.ldpc_hip_bp_set_osd(self._h, int(osd_method), int(osd_order)))

    def osd_status(self, batch):
        """Per row of the last BP + OSD decode: 0 BP converged, 1 OSD solved H x = s, 2 syndrome outside the image of H
        (``ldpc_hip_bposd_get_status``)."""
        out = np.zeros(int(batch), np.uint8)
        _lib.check(self._lib.ldpc_hip_bposd_get_status(self._h, out.ctypes.data, int(batch)))
        return out

    def set_repack(self, first_pass_iters):
        """Serial schedule: iterations of the first pass before unconverged rows are repacked (-1 auto, 0 off)."""
        _lib.check(self._lib.ldpc_hip_bp_set_repack(self._h, int(first_pass_iters)))

    def set_serial_kernel(self, mode):
        """Serial schedule: -1 automatic, 0 one wavefront per tile (bit by bit), 1 level-parallel workgroup per tile."""
        _lib.check(self._lib.ldpc_hip_bp_set_serial_kernel(self._h, int(mode)))

    def set_osd_kernel(self, mode):
        """OSD elimination: -1 automatic (registers / LDS / HBM by size), 0 the LDS kernels, 2 OSD-0 through the HBM kernel."""
        _lib.check(self._lib.ldpc_hip_bp_set_osd_kernel(self._h, int(mode)))

    def set_small_code_kernel(self, mode):
        """On-chip kernels for small codes: -1 automatic (default), 0 never, 1 whenever a syndrome fits in LDS, 2 slot kernel only, 3 lane = node wavefront kernel only (4 / 5: one wavefront / a workgroup per syndrome), 6 lane = edge kernel where it applies."""
        _lib.check(self._lib.ldpc_hip_bp_set_small_code_kernel(self._h, int(mode)))

    def set_debug_switch(self, name, value=1):
        """Measurement / test switch of the handle (``ldpc_hip_bp_set_debug_switch``; never changes a result).  ``value < 0`` unsets."""
        _lib.check(self._lib.ldpc_hip_bp_set_debug_switch(self._h, str(name).encode(), int(value)))

    def workspace_bytes(self, batch):
        return int(self._lib.ldpc_hip_bp_workspace_bytes(self._h, int(batch)))

    def last_kernel_ms(self) -> float:
        ms = C.c_float(0.0)
        _lib.check(self._lib.ldpc_hip_bp_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def last_phase_ms(self):
        """(persistent kernel ms, per-pass kernels ms) of the last streaming decode; (0, total) or (0, 0) otherwise."""
        a, b = C.c_float(0.0), C.c_float(0.0)
        _lib.check(self._lib.ldpc_hip_bp_last_phase_ms(self._h, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def clock_probe(self):
        """(shader cycles, constant-rate ticks, tick rate in Hz) accumulated by the workgroups of this handle's long-running BP kernels
        so far (``ldpc_hip_bp_clock_probe``; waits for the stream).  Take differences around a timed region: see ``clock_ghz``."""
        c, t, hz = C.c_uint64(0), C.c_uint64(0), C.c_double(0.0)
        _lib.check(self._lib.ldpc_hip_bp_clock_probe(self._h, C.byref(c), C.byref(t), C.byref(hz)))
        return int(c.value), int(t.value), float(hz.value)

    def copy_probe(self, tiles: int, segments_per_tile: int | None = None, passes: int = 2):
        """(ms, GB/s read + written) of a bare copy of ``tiles`` x ``segments_per_tile`` 512-byte message segments between the handle's
        two message arrays (``ldpc_hip_bp_copy_probe``): what this box gives the streamed kernels' traffic with no arithmetic."""
        ms, rate = C.c_float(0.0), C.c_double(0.0)
        _lib.check(self._lib.ldpc_hip_bp_copy_probe(self._h, int(tiles), int(segments_per_tile or self.nnz), int(passes), C.byref(ms), C.byref(rate)))
        return float(ms.value), float(rate.value)

    @staticmethod
    def clock_ghz(before, after):
        """Average shader clock (GHz) of the BP kernels that ran between two ``clock_probe()`` readings, or None if none did."""
        dc, dt = after[0] - before[0], after[1] - before[1]
        return dc / dt * after[2] / 1e9 if dt > 0 and dc > 0 else None

    # -- data path --------------------------------------------------------------------------------
    def decode_batch(self, syndromes, want_llr=True, out=None, asynchronous=False, osd0=False, osd=False, llr_out=None):
        """Decode ``(B, m)`` uint8 syndromes.  Returns ``(decoding, llr|None, iterations, converge)``.

        ``osd0=True`` runs BP + OSD-0 (``ldpc_hip_bposd0_decode_batch``): ``decoding`` holds the OSD-0 solution for
        rows BP left unconverged; llr / iterations / converge remain BP's.  ``osd=True`` does the same with the
        method and order given to ``set_osd`` (``ldpc_hip_bposd_decode_batch``: OSD_0, OSD_E or OSD_CS).

        NumPy in -> NumPy out (host pointers); torch CUDA tensor in -> torch CUDA tensors out.
        ``out`` may carry preallocated torch outputs ``(decoding, llr, iterations, converge)``.
        """
        if _is_torch(syndromes):
            import torch
            s = syndromes
            if s.dtype != torch.uint8 or s.dim() != 2 or s.shape[1] != self.m or not s.is_cuda:
                raise ValueError(f"syndromes must be a CUDA uint8 tensor of shape (B, {self.m})")
            s = s.contiguous()
            b = int(s.shape[0])
            # launch on torch's current stream so the call is ordered after whatever produced `s`
            self.set_stream(torch.cuda.current_stream(s.device).cuda_stream)
            if out is not None:
                dec, llr, it, cv = out
            else:
                dec = torch.empty((b, self.n), dtype=torch.uint8, device=s.device)
                llr = torch.empty((b, self.n), dtype=torch.float64, device=s.device) if want_llr else None
                it = torch.empty((b,), dtype=torch.int32, device=s.device)
                cv = torch.empty((b,), dtype=torch.uint8, device=s.device)
            if osd:
                fn = self._lib.ldpc_hip_bposd_decode_batch_async if asynchronous else self._lib.ldpc_hip_bposd_decode_batch
            elif osd0:
                fn = self._lib.ldpc_hip_bposd0_decode_batch_async if asynchronous else self._lib.ldpc_hip_bposd0_decode_batch
            else:
                fn = self._lib.ldpc_hip_bp_decode_batch_async if asynchronous else self._lib.ldpc_hip_bp_decode_batch
            _lib.check(fn(self._h, s.data_ptr(), b, dec.data_ptr(), llr.data_ptr() if llr is not None else None,
                          it.data_ptr(), cv.data_ptr()))
            return dec, llr, it, cv
        s = np.ascontiguousarray(syndromes, np.uint8)
        if s.ndim != 2 or s.shape[1] != self.m:
            raise ValueError(f"syndromes must have shape (B, {self.m})")
        b = s.shape[0]
        dec = np.empty((b, self.n), np.uint8)  # (every element is written by the call; zero-filling 6 GB first costs as much as the decode)
        if want_llr and llr_out is not None:  # (NumPy path only: the caller's array for the log-ratios)
            if llr_out.dtype != np.float64 or llr_out.shape != (b, self.n) or not llr_out.flags.c_contiguous:
                raise ValueError(f"llr_out must be a C-contiguous float64 array of shape ({b}, {self.n})")
            llr = llr_out
        else:
            llr = np.empty((b, self.n), np.float64) if want_llr else None
        it = np.empty(b, np.int32)
        cv = np.empty(b, np.uint8)
        fn = (self._lib.ldpc_hip_bposd_decode_batch if osd else
              self._lib.ldpc_hip_bposd0_decode_batch if osd0 else self._lib.ldpc_hip_bp_decode_batch)
        _lib.check(fn(self._h, s.ctypes.data, b, dec.ctypes.data, llr.ctypes.data if want_llr else None,
                      it.ctypes.data, cv.ctypes.data))
        return dec, llr, it, cv.astype(bool)

    def soft_info_decode_batch(self, soft_syndromes, cutoff, sigma, want_llr=True):
        """``(B, m)`` float64 analog syndromes -> ``(decoding, llr|None, iterations, converge, soft_syndrome)``
        (``ldpc_hip_bp_soft_info_decode_batch``; NumPy or torch CUDA tensors)."""
        if _is_torch(soft_syndromes):
            import torch
            s = soft_syndromes.contiguous()
            if s.dtype != torch.float64 or s.dim() != 2 or s.shape[1] != self.m or not s.is_cuda:
                raise ValueError(f"soft_syndromes must be a CUDA float64 tensor of shape (B, {self.m})")
            b = int(s.shape[0])
            self.set_stream(torch.cuda.current_stream(s.device).cuda_stream)
            dec = torch.empty((b, self.n), dtype=torch.uint8, device=s.device)
            llr = torch.empty((b, self.n), dtype=torch.float64, device=s.device) if want_llr else None
            it = torch.empty((b,), dtype=torch.int32, device=s.device)
            cv = torch.empty((b,), dtype=torch.uint8, device=s.device)
            so = torch.empty((b, self.m), dtype=torch.float64, device=s.device)
            _lib.check(self._lib.ldpc_hip_bp_soft_info_decode_batch(
                self._h, s.data_ptr(), b, float(cutoff), float(sigma), dec.data_ptr(), llr.data_ptr() if want_llr else None,
                it.data_ptr(), cv.data_ptr(), so.data_ptr()))
            return dec, llr, it, cv, so
        s = np.ascontiguousarray(soft_syndromes, np.float64)
        if s.ndim != 2 or s.shape[1] != self.m:
            raise ValueError(f"soft_syndromes must have shape (B, {self.m})")
        b = s.shape[0]
        dec = np.zeros((b, self.n), np.uint8)
        llr = np.zeros((b, self.n), np.float64) if want_llr else None
        it = np.zeros(b, np.int32)
        cv = np.zeros(b, np.uint8)
        so = np.zeros((b, self.m), np.float64)
        _lib.check(self._lib.ldpc_hip_bp_soft_info_decode_batch(
            self._h, s.ctypes.data, b, float(cutoff), float(sigma), dec.ctypes.data, llr.ctypes.data if want_llr else None,
            it.ctypes.data, cv.ctypes.data, so.ctypes.data))
        return dec, llr, it, cv.astype(bool), so

    def pack_b8(self, bits_tensor):
        """``(B, bits)`` uint8 CUDA tensor (one byte per bit) -> ``(B, ceil(bits / 8))`` packed, on the tensor's stream."""
        import torch
        t = bits_tensor.contiguous()
        b, bits = int(t.shape[0]), int(t.shape[1])
        self.set_stream(torch.cuda.current_stream(t.device).cuda_stream)
        out = torch.empty((b, (bits + 7) // 8), dtype=torch.uint8, device=t.device)
        _lib.check(self._lib.ldpc_hip_pack_b8(self._h, t.data_ptr(), b, bits, out.data_ptr()))
        return out

    def unpack_b8(self, packed_tensor, bits):
        import torch
        t = packed_tensor.contiguous()
        b = int(t.shape[0])
        self.set_stream(torch.cuda.current_stream(t.device).cuda_stream)
        out = torch.empty((b, int(bits)), dtype=torch.uint8, device=t.device)
        _lib.check(self._lib.ldpc_hip_unpack_b8(self._h, t.data_ptr(), b, int(bits), out.data_ptr()))
        return out

    def set_observables(self, observables_matrix):
        """The k x n matrix whose product with a decoding gives the predicted observables (``decode_b8``)."""
        import scipy.sparse as sp
        obs = sp.csr_matrix(observables_matrix)
        obs.eliminate_zeros()
        obs.sort_indices()
        if obs.shape[1] != self.n:
            raise ValueError(f"observables_matrix must have {self.n} columns")
        rp = np.ascontiguousarray(obs.indptr, np.int32)
        ci = np.ascontiguousarray(obs.indices, np.int32)
        self.k = int(obs.shape[0])
        _lib.check(self._lib.ldpc_hip_bp_set_observables(self._h, self.k, rp.ctypes.data, ci.ctypes.data if len(ci) else None))

    def decode_b8(self, dets_b8, with_osd=False, want_decoding=False):
        """Bit-packed shots in (``(B, ceil(m/8))`` uint8, stim "b8"), bit-packed predictions out.

        Returns ``(obs_b8 (B, ceil(k/8)), decoding_b8 (B, ceil(n/8)) | None, iterations, converge)``; NumPy or torch
        CUDA tensors as for ``decode_batch``.  ``with_osd`` uses the method / order given to ``set_osd``."""
        mb, nb, kb = (self.m + 7) // 8, (self.n + 7) // 8, (getattr(self, "k", 0) + 7) // 8
        if not hasattr(self, "k"):
            raise ValueError("call set_observables first")
        if _is_torch(dets_b8):
            import torch
            d = dets_b8.contiguous()
            if d.dtype != torch.uint8 or d.dim() != 2 or d.shape[1] != mb or not d.is_cuda:
                raise ValueError(f"dets_b8 must be a CUDA uint8 tensor of shape (B, {mb})")
            b = int(d.shape[0])
            self.set_stream(torch.cuda.current_stream(d.device).cuda_stream)
            obs = torch.empty((b, kb), dtype=torch.uint8, device=d.device)
            dec = torch.empty((b, nb), dtype=torch.uint8, device=d.device) if want_decoding else None
            it = torch.empty((b,), dtype=torch.int32, device=d.device)
            cv = torch.empty((b,), dtype=torch.uint8, device=d.device)
            _lib.check(self._lib.ldpc_hip_bp_decode_b8(self._h, d.data_ptr(), b, int(bool(with_osd)), obs.data_ptr(),
                                                       dec.data_ptr() if want_decoding else None, it.data_ptr(), cv.data_ptr()))
            return obs, dec, it, cv
        d = np.ascontiguousarray(dets_b8, np.uint8)
        if d.ndim != 2 or d.shape[1] != mb:
            raise ValueError(f"dets_b8 must have shape (B, {mb})")
        b = d.shape[0]
        obs = np.zeros((b, kb), np.uint8)
        dec = np.zeros((b, nb), np.uint8) if want_decoding else None
        it = np.zeros(b, np.int32)
        cv = np.zeros(b, np.uint8)
        _lib.check(self._lib.ldpc_hip_bp_decode_b8(self._h, d.ctypes.data, b, int(bool(with_osd)), obs.ctypes.data,
                                                   dec.ctypes.data if want_decoding else None, it.ctypes.data, cv.ctypes.data))
        return obs, dec, it, cv.astype(bool)

    def mulvec_batch(self, vectors):
        """``GF2Sparse::mulvec`` (gf2sparse.hpp:177-214) for every row of ``vectors`` (B, n)."""
        if _is_torch(vectors):
            import torch
            v = vectors.contiguous()
            self.set_stream(torch.cuda.current_stream(v.device).cuda_stream)
            out = torch.empty((v.shape[0], self.m), dtype=torch.uint8, device=v.device)
            _lib.check(self._lib.ldpc_hip_gf2_mulvec_batch(self._h, v.data_ptr(), int(v.shape[0]), out.data_ptr()))
            return out
        v = np.ascontiguousarray(vectors, np.uint8)
        out = np.zeros((v.shape[0], self.m), np.uint8)
        _lib.check(self._lib.ldpc_hip_gf2_mulvec_batch(self._h, v.ctypes.data, v.shape[0], out.ctypes.data))
        return out

    def gen_bsc_syndromes(self, seed, error_rate, shot0, shots, device=None, want_errors=False):
        """Synthetic BSC shots generated on the GPU (twin of ``noise_models.generate_bsc_batch`` + H e)."""
        from ldpc_amd.prng import bernoulli_threshold
        thr = bernoulli_threshold(error_rate)
        if device is not None:
            import torch
            self.set_stream(torch.cuda.current_stream(device).cuda_stream)
            synd = torch.empty((shots, self.m), dtype=torch.uint8, device=device)
            err = torch.empty((shots, self.n), dtype=torch.uint8, device=device) if want_errors else None
            _lib.check(self._lib.ldpc_hip_gen_bsc_syndromes(
                self._h, seed, thr, shot0, shots, synd.data_ptr(), err.data_ptr() if want_errors else None))
        else:
            synd = np.zeros((shots, self.m), np.uint8)
            err = np.zeros((shots, self.n), np.uint8) if want_errors else None
            _lib.check(self._lib.ldpc_hip_gen_bsc_syndromes(
                self._h, seed, thr, shot0, shots, synd.ctypes.data, err.ctypes.data if want_errors else None))
        return (synd, err) if want_errors else synd


class HipBpMultiEngine:
    """One decoder over several GPUs of the node inside ONE process (``ldpc_hip_bp_multi``, include/ldpc_hip.h).

    ``decode_batch`` cuts the batch into contiguous row ranges, one per entry of ``device_ids``, decodes them
    concurrently and returns when every row is in place; results equal the single-GPU call's bit for bit.  Setters are
    those of ``HipBpEngine`` and apply to every GPU.  NumPy in -> NumPy out (each GPU stages its own rows over PCIe);
    torch CUDA tensors in -> torch CUDA tensors out on the same GPU (the others get their rows by peer copy over xGMI
    and return their decisions bit-packed).  For one process PER GPU use ``ldpc_amd.sharding`` instead.
    """

    _BROADCAST = ("set_debug_switch", "set_channel", "set_params", "set_schedule", "set_random_serial", "set_tuning", "set_math", "set_ring", "set_handoff", "set_osd",
                  "set_repack", "set_serial_kernel", "set_osd_kernel", "set_small_code_kernel")

    def __init__(self, row_ptr, col_idx, n, channel_probs, max_iter, bp_method, ms_scaling_factor, device_ids):
        self._lib = _lib.load()
        self._mh = C.c_void_p()
        row_ptr = np.ascontiguousarray(row_ptr, np.int32)
        col_idx = np.ascontiguousarray(col_idx, np.int32)
        probs = np.ascontiguousarray(channel_probs, np.float64)
        self.m, self.n, self.nnz = int(len(row_ptr) - 1), int(n), int(len(col_idx))
        if probs.shape != (self.n,):
            raise ValueError("Channel probabilities vector must have length equal to the number of bits")
        self.device_ids = [int(d) for d in device_ids]
        if not self.device_ids:
            raise ValueError("device_ids must name at least one GPU")
        ids = (C.c_int32 * len(self.device_ids))(*self.device_ids)
        desc = _lib.BpDesc(m=self.m, n=self.n, nnz=self.nnz, csr_row_ptr=row_ptr.ctypes.data_as(C.POINTER(C.c_int32)),
                           csr_col_idx=col_idx.ctypes.data_as(C.POINTER(C.c_int32)),
                           channel_probs=probs.ctypes.data_as(C.POINTER(C.c_double)), max_iter=int(max_iter),
                           bp_method=int(bp_method), ms_scaling_factor=float(ms_scaling_factor), device=-1)
        _lib.check(self._lib.ldpc_hip_bp_multi_create(C.byref(desc), ids, len(self.device_ids), C.byref(self._mh)))
        self.subs = [HipBpEngine._view(self._lib.ldpc_hip_bp_multi_handle(self._mh, i), self.m, self.n, self.nnz, d)
                     for i, d in enumerate(self.device_ids)]
        for name in self._BROADCAST:
            setattr(self, name, self._broadcast(name))

    def _broadcast(self, name):
        def apply(*a, **k):
            for sub in self.subs:
                getattr(sub, name)(*a, **k)
        apply.__name__ = name
        return apply

    def close(self):
        if getattr(self, "_mh", None) is not None and self._mh.value:
            for sub in self.subs:
                sub.close()
            self._lib.ldpc_hip_bp_multi_destroy(self._mh)
            self._mh = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def schedule_order(self):
        """The order serial_relative / the random schedule left behind (one state for all GPUs: every call ends by copying the
        state of the batch's last row to every handle, ``multi_device.h``)."""
        return self.subs[0].schedule_order()

    def soft_info_decode_batch(self, soft_syndromes, cutoff, sigma, want_llr=True):
        """Soft-syndrome decoding is a per-handle call in the C ABI (include/ldpc_hip.h): it runs on the first GPU of
        ``device_ids``, unsharded -- same results as the single-GPU engine."""
        if _is_torch(soft_syndromes) and soft_syndromes.is_cuda and soft_syndromes.device.index != self.subs[0].device:
            raise ValueError(f"soft_syndromes must live on cuda:{self.subs[0].device} (the first of device_ids)")
        return self.subs[0].soft_info_decode_batch(soft_syndromes, cutoff, sigma, want_llr=want_llr)

    def set_staging(self, force):
        """Testing aid (``ldpc_hip_bp_multi_set_staging``): route device tensors through the peer-copy path on their own GPU too."""
        _lib.check(self._lib.ldpc_hip_bp_multi_set_staging(self._mh, int(bool(force))))

    def last_kernel_ms(self):
        """BP kernel time of the last decode on every GPU."""
        ms = (C.c_float * len(self.subs))()
        _lib.check(self._lib.ldpc_hip_bp_multi_last_kernel_ms(self._mh, ms))
        return [float(v) for v in ms]

    def osd_status(self, batch):
        """Status of every row of the last BP + OSD decode, shard by shard (rows are cut on 64-row boundaries)."""
        batch = int(batch)
        nd, tiles = len(self.subs), (batch + 63) // 64
        base, rem = divmod(tiles, nd)
        parts = []
        for d, sub in enumerate(self.subs):
            t0 = d * base + min(d, rem)
            t1 = t0 + base + (1 if d < rem else 0)
            lo, hi = min(t0 * 64, batch), min(t1 * 64, batch)
            if hi > lo:
                parts.append(sub.osd_status(hi - lo))
        return np.concatenate(parts) if parts else np.zeros(0, np.uint8)

    def _sub_for(self, tensor=None):
        if tensor is not None and _is_torch(tensor) and tensor.is_cuda:
            for sub in self.subs:
                if sub.device == tensor.device.index:
                    return sub
        return self.subs[0]

    def mulvec_batch(self, vectors):
        return self._sub_for(vectors).mulvec_batch(vectors)

    def gen_bsc_syndromes(self, seed, error_rate, shot0, shots, device=None, want_errors=False):
        sub = self.subs[0]
        if device is not None:
            import torch
            idx = torch.device(device).index
            sub = next((s_ for s_ in self.subs if s_.device == idx), sub)
        return sub.gen_bsc_syndromes(seed, error_rate, shot0, shots, device=device, want_errors=want_errors)

    def decode_batch(self, syndromes, want_llr=True, out=None, osd0=False, osd=False, asynchronous=False, llr_out=None):
        """As ``HipBpEngine.decode_batch`` (always synchronous: the call returns when every GPU has delivered its rows)."""
        with_osd = 1 if osd else (0 if osd0 else -1)
        if _is_torch(syndromes):
            import torch
            s = syndromes
            if s.dtype != torch.uint8 or s.dim() != 2 or s.shape[1] != self.m or not s.is_cuda:
                raise ValueError(f"syndromes must be a CUDA uint8 tensor of shape (B, {self.m})")
            s = s.contiguous()
            b = int(s.shape[0])
            if out is not None:
                dec, llr, it, cv = out
            else:
                dec = torch.empty((b, self.n), dtype=torch.uint8, device=s.device)
                llr = torch.empty((b, self.n), dtype=torch.float64, device=s.device) if want_llr else None
                it = torch.empty((b,), dtype=torch.int32, device=s.device)
                cv = torch.empty((b,), dtype=torch.uint8, device=s.device)
            torch.cuda.current_stream(s.device).synchronize()  # the GPUs read `s` on their own streams
            _lib.check(self._lib.ldpc_hip_bp_multi_decode_batch(self._mh, with_osd, s.data_ptr(), b, dec.data_ptr(),
                                                                llr.data_ptr() if llr is not None else None, it.data_ptr(), cv.data_ptr()))
            return dec, llr, it, cv
        s = np.ascontiguousarray(syndromes, np.uint8)
        if s.ndim != 2 or s.shape[1] != self.m:
            raise ValueError(f"syndromes must have shape (B, {self.m})")
        b = s.shape[0]
        dec = np.empty((b, self.n), np.uint8)
        if want_llr and llr_out is not None:
            if llr_out.dtype != np.float64 or llr_out.shape != (b, self.n) or not llr_out.flags.c_contiguous:
                raise ValueError(f"llr_out must be a C-contiguous float64 array of shape ({b}, {self.n})")
            llr = llr_out
        else:
            llr = np.empty((b, self.n), np.float64) if want_llr else None
        it = np.empty(b, np.int32)
        cv = np.empty(b, np.uint8)
        _lib.check(self._lib.ldpc_hip_bp_multi_decode_batch(self._mh, with_osd, s.ctypes.data, b, dec.ctypes.data,
                                                            llr.ctypes.data if want_llr else None, it.ctypes.data, cv.ctypes.data))
        return dec, llr, it, cv.astype(bool)
