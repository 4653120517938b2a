"""ctypes binding of libldpc_hip.so (C ABI: include/ldpc_hip.h).

There is exactly one compute path: the HIP library.  If it is missing or does not load, importing
the decoder raises -- there is no CPU fallback (the CPU restatement under oracle/ is a test checker
and is never imported from here).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (LDPC_HIP_LIB: another build of the same library, for A/B measurements of kernel variants on one box)
LIB_PATH = os.environ.get("LDPC_HIP_LIB") or os.path.join(_HERE, "lib", "libldpc_hip.so")

# every symbol include/ldpc_hip.h declares (tests/test_cabi_symbols.py checks header <-> library)
SYMBOLS = (
    "ldpc_hip_bp_create", "ldpc_hip_bp_destroy", "ldpc_hip_bp_set_channel", "ldpc_hip_bp_set_params",
    "ldpc_hip_bp_set_stream", "ldpc_hip_bp_set_schedule", "ldpc_hip_bp_set_random_serial", "ldpc_hip_bp_get_schedule_order", "ldpc_hip_bp_decode_batch", "ldpc_hip_bp_decode_batch_async", "ldpc_hip_bposd0_decode_batch", "ldpc_hip_bposd0_decode_batch_async",
    "ldpc_hip_bp_set_osd", "ldpc_hip_bposd_get_status", "ldpc_hip_bp_set_osd_kernel", "ldpc_hip_bp_set_repack", "ldpc_hip_bp_set_serial_kernel", "ldpc_hip_bposd_decode_batch", "ldpc_hip_bposd_decode_batch_async",
    "ldpc_hip_bp_set_observables", "ldpc_hip_bp_decode_b8", "ldpc_hip_bp_soft_info_decode_batch", "ldpc_hip_pack_b8", "ldpc_hip_unpack_b8", "ldpc_hip_bp_last_phase_ms", "ldpc_hip_bp_clock_probe", "ldpc_hip_bp_copy_probe", "ldpc_hip_host_alloc", "ldpc_hip_host_free",
    "ldpc_hip_gf2_mulvec_batch", "ldpc_hip_gen_bsc_syndromes", "ldpc_hip_bp_last_kernel_ms",
    "ldpc_hip_bp_workspace_bytes", "ldpc_hip_bp_set_tuning", "ldpc_hip_bp_set_math", "ldpc_hip_bp_set_ring", "ldpc_hip_bp_set_small_code_kernel", "ldpc_hip_bp_set_handoff", "ldpc_hip_last_error", "ldpc_hip_version",
    "ldpc_hip_bp_set_debug_switch", "ldpc_hip_bp_multi_create", "ldpc_hip_bp_multi_destroy", "ldpc_hip_bp_multi_devices", "ldpc_hip_bp_multi_handle",
    "ldpc_hip_bp_multi_decode_batch", "ldpc_hip_bp_multi_last_kernel_ms", "ldpc_hip_bp_multi_set_staging",
)


class BpDesc(C.Structure):
    """``ldpc_hip_bp_desc`` (include/ldpc_hip.h)."""
    _fields_ = [
        ("m", C.c_int32), ("n", C.c_int32), ("nnz", C.c_int32),
        ("csr_row_ptr", C.POINTER(C.c_int32)), ("csr_col_idx", C.POINTER(C.c_int32)),
        ("channel_probs", C.POINTER(C.c_double)),
        ("max_iter", C.c_int32), ("bp_method", C.c_int32),
        ("ms_scaling_factor", C.c_double), ("device", C.c_int32),
    ]


class LdpcHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load libldpc_hip.so once; raise loudly if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'  or  make -C ldpc_amd/csrc). "
            "ldpc_amd has no CPU fallback.")
    # torch bundles its own libamdhip64/libhsa-runtime64 (SONAME libamdhip64.so.7).  If this library
    # pulled in /opt/rocm's copy first, a later `import torch` would load a SECOND HIP runtime and find
    # no GPUs; importing torch first makes both share torch's runtime.  (A plain C consumer of the ABI
    # links /opt/rocm's runtime and never meets torch.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
    lib.ldpc_hip_bp_create.argtypes = [C.POINTER(BpDesc), C.POINTER(vp)]
    lib.ldpc_hip_bp_destroy.argtypes = [vp]
    lib.ldpc_hip_bp_destroy.restype = None
    lib.ldpc_hip_bp_set_channel.argtypes = [vp, C.POINTER(dbl), i32]
    lib.ldpc_hip_bp_set_params.argtypes = [vp, i32, i32, dbl]
    lib.ldpc_hip_bp_set_stream.argtypes = [vp, vp]
    lib.ldpc_hip_bp_set_schedule.argtypes = [vp, i32, vp]
    lib.ldpc_hip_bp_set_random_serial.argtypes = [vp, i32, C.c_uint32]
    lib.ldpc_hip_bp_get_schedule_order.argtypes = [vp, vp]
    lib.ldpc_hip_bp_decode_batch.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    lib.ldpc_hip_bp_decode_batch_async.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    lib.ldpc_hip_bposd0_decode_batch.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    lib.ldpc_hip_bposd0_decode_batch_async.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    lib.ldpc_hip_bp_set_osd.argtypes = [vp, i32, i32]
    lib.ldpc_hip_bposd_get_status.argtypes = [vp, vp, i64]
    lib.ldpc_hip_bp_set_osd_kernel.argtypes = [vp, i32]
    lib.ldpc_hip_bp_set_repack.argtypes = [vp, i32]
    lib.ldpc_hip_bp_set_serial_kernel.argtypes = [vp, i32]
    lib.ldpc_hip_bp_set_observables.argtypes = [vp, i32, vp, vp]
    lib.ldpc_hip_bp_decode_b8.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp]
    lib.ldpc_hip_pack_b8.argtypes = [vp, vp, i64, i32, vp]
    lib.ldpc_hip_unpack_b8.argtypes = [vp, vp, i64, i32, vp]
    lib.ldpc_hip_bp_soft_info_decode_batch.argtypes = [vp, vp, i64, C.c_double, C.c_double, vp, vp, vp, vp, vp]
    lib.ldpc_hip_bposd_decode_batch.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    lib.ldpc_hip_bposd_decode_batch_async.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    lib.ldpc_hip_gf2_mulvec_batch.argtypes = [vp, vp, i64, vp]
    lib.ldpc_hip_gen_bsc_syndromes.argtypes = [vp, u64, u64, i64, i64, vp, vp]
    lib.ldpc_hip_bp_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.ldpc_hip_bp_last_phase_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.ldpc_hip_bp_clock_probe.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_double)]
    if not os.environ.get("LDPC_HIP_LIB") or hasattr(lib, "ldpc_hip_bp_copy_probe"):  # (an older build under A/B measurement may lack the probe)
        lib.ldpc_hip_bp_copy_probe.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_double)]
    lib.ldpc_hip_host_alloc.argtypes = [C.c_size_t]
    lib.ldpc_hip_host_alloc.restype = C.c_void_p
    lib.ldpc_hip_host_free.argtypes = [vp]
    lib.ldpc_hip_host_free.restype = None
    lib.ldpc_hip_bp_workspace_bytes.argtypes = [vp, i64]
    lib.ldpc_hip_bp_workspace_bytes.restype = i64
    lib.ldpc_hip_bp_set_tuning.argtypes = [vp, i32, i32]
    lib.ldpc_hip_bp_set_math.argtypes = [vp, i32]
    lib.ldpc_hip_bp_set_ring.argtypes = [vp, i32]
    lib.ldpc_hip_bp_set_small_code_kernel.argtypes = [vp, i32]
    lib.ldpc_hip_bp_set_debug_switch.argtypes = [vp, C.c_char_p, i32]
    lib.ldpc_hip_bp_set_handoff.argtypes = [vp, i32]
    lib.ldpc_hip_bp_multi_create.argtypes = [C.POINTER(BpDesc), C.POINTER(i32), i32, C.POINTER(vp)]
    lib.ldpc_hip_bp_multi_destroy.argtypes = [vp]
    lib.ldpc_hip_bp_multi_destroy.restype = None
    lib.ldpc_hip_bp_multi_devices.argtypes = [vp]
    lib.ldpc_hip_bp_multi_devices.restype = i32
    lib.ldpc_hip_bp_multi_handle.argtypes = [vp, i32]
    lib.ldpc_hip_bp_multi_handle.restype = vp
    lib.ldpc_hip_bp_multi_decode_batch.argtypes = [vp, i32, vp, i64, vp, vp, vp, vp]
    lib.ldpc_hip_bp_multi_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.ldpc_hip_bp_multi_set_staging.argtypes = [vp, i32]
    lib.ldpc_hip_last_error.restype = C.c_char_p
    lib.ldpc_hip_version.restype = C.c_char_p
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().ldpc_hip_last_error().decode("utf-8", "replace")
        raise LdpcHipError(f"libldpc_hip error {rc}: {msg}")


class PinnedBlock:
    """A block of page-locked host memory (``ldpc_hip_host_alloc``) that NumPy arrays can sit on: ``array(shape, dtype)`` gives an array
    whose ``base`` is this object -- views of it refer to the block too -- and the memory goes back when the last of them is gone.
    ``PinnedBlock.try_new`` returns None where the memory cannot be had."""

    def __init__(self, lib, ptr: int, nbytes: int):
        self._lib, self.ptr, self.nbytes = lib, ptr, nbytes

    @classmethod
    def try_new(cls, nbytes: int):
        try:
            lib = load()
            ptr = lib.ldpc_hip_host_alloc(int(nbytes))
        except Exception:
            return None
        return cls(lib, ptr, int(nbytes)) if ptr else None

    def array(self, shape, dtype):
        import numpy as np
        dt = np.dtype(dtype)
        if int(np.prod(shape)) * dt.itemsize > self.nbytes:
            raise ValueError("array larger than the block")

        class _Iface:  # (what np.asarray reads; it keeps `owner` as the array's base)
            pass
        holder = _Iface()
        holder.owner = self
        holder.__array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": dt.str, "data": (self.ptr, False), "version": 3}
        return np.asarray(holder)

    def __del__(self):
        ptr, self.ptr = getattr(self, "ptr", 0), 0
        if ptr:
            try:
                self._lib.ldpc_hip_host_free(ptr)
            except Exception:
                pass
