"""Host-side mirror of the reference's ``BpDecoder`` (src_python/ldpc/bp_decoder/_bp_decoder.pyx).

Same constructor keywords, property names, alias strings, exception types and messages as the
reference, so code written against ``ldpc.BpDecoder`` runs unchanged; the arithmetic happens in
libldpc_hip.so (HIP kernels for gfx950) behind ``ldpc_amd.engine.HipBpEngine``.  There is no CPU
fallback: ``decode`` raises if the HIP library or a GPU is missing.

Additions over the reference (it has no batch API): ``decode_batch`` and the
``*_batch`` result views.

Reference behaviours deliberately reproduced (each cited where implemented):
  * effective defaults come from the base initialiser's ``kwargs.get`` (pyx:90-100), not from the
    subclass signature: ``bp_method`` defaults to product_sum, ``max_iter=0`` means ``n``;
  * an all-zero input returns zeros with ``converge=True`` WITHOUT running BP, leaving
    ``log_prob_ratios`` / ``iter`` / ``decoding`` from the previous call (pyx:679-681, 688-690);
  * the output array has the input's dtype (pyx:673, 693-695).
"""
from __future__ import annotations

import warnings
from typing import List, Optional, Union

import numpy as np
import scipy.sparse

from ldpc_amd.helpers.scipy_helpers import convert_to_binary_sparse

# ldpc::bp enums (bp.hpp:23-38)
PRODUCT_SUM, MINIMUM_SUM = 0, 1
SERIAL, PARALLEL, SERIAL_RELATIVE = 0, 1, 2
SYNDROME, RECEIVED_VECTOR, AUTO = 0, 1, 2

_PS_ALIASES = ("prod_sum", "product_sum", "ps", "0", "prod sum")  # pyx:386
_MS_ALIASES = ("min_sum", "minimum_sum", "ms", "1", "minimum sum", "min sum")  # pyx:388
_UNSET = object()  # "keyword not passed by the caller"


def _bp_method_code(value) -> int:
    """The alias table of the ``bp_method`` setter (pyx:384-394) as a function."""
    key = str(value).lower()
    if key in _PS_ALIASES:
        return PRODUCT_SUM
    if key in _MS_ALIASES:
        return MINIMUM_SUM
    raise ValueError(f"BP method '{value}' is invalid. \
                    Please choose from the following methods: \
                    'product_sum', 'minimum_sum'")


def _check_pcm_type(pcm, spaces=12):
    if not isinstance(pcm, (np.ndarray, scipy.sparse.spmatrix)):  # pyx:17-21, 113-117
        # (the reference's message carries the indentation of its source line across a backslash continuation: 12 spaces in
        # BpDecoderBase.__cinit__, pyx:113-117, 8 in Py2BpSparse, pyx:17-21)
        raise TypeError("The input matrix is of an invalid type. Please input" + " " * spaces +
                        f"a np.ndarray or scipy.sparse.spmatrix object, not {type(pcm)}")


def _ingest(pcm) -> scipy.sparse.csr_matrix:
    """Py2BpSparse (pyx:9-49): validate and return canonical CSR (sorted columns, ones only)."""
    _check_pcm_type(pcm, spaces=8)
    # (a copy: the normalisation below must not reach into the caller's arrays -- the reference only reads pcm.nonzero();
    # like the reference's helper, convert_to_binary_sparse itself drops explicit zeros of a sparse input in place)
    h = scipy.sparse.csr_matrix(convert_to_binary_sparse(pcm), copy=True)
    h.sum_duplicates()  # insert_entry returns the existing entry for a repeated coordinate
    h.data[:] = 1       # (sparse_matrix_base.hpp:437-440), so duplicates collapse to a single one
    h.sort_indices()
    return h


def io_test(pcm):
    """Round-trip a matrix through the ingest path (reference ``io_test``, pyx:74-78)."""
    return scipy.sparse.csr_matrix(_ingest(pcm), dtype=np.uint8)


# ---- what Cython does with annotated arguments -------------------------------------------------------------------
# The reference's classes are Cython `cdef class`es compiled with annotation typing: an argument annotated with a builtin
# type (`value: int`, `input_type: str`, `Optional[float]`, `List[int]`) must be EXACTLY of that type (None allowed only for
# Optional[...]), `value: float` / `sigma: float` become C doubles (anything with __float__ / __index__ converts), `value: bool`
# becomes a C truth value, and a violation is a TypeError whose text is Cython's.  python_test/test_bp_decoder.py:143-168 pins
# some of these; tests/golden/api_reference.json (generated from the reference) pins all of them.
def _type_name(t) -> str:
    return t.__name__ if t.__module__ == "builtins" else f"{t.__module__}.{t.__name__}"


def _typed(name, value, typ, optional=True):
    """Exact-type test of an argument annotated with the builtin / extension type ``typ``."""
    if value is None and optional:
        return
    if type(value) is typ or (typ is np.ndarray and isinstance(value, np.ndarray)):
        return
    msg = f"Argument '{name}' has incorrect type (expected {_type_name(typ)}, got {_type_name(type(value))})"
    if isinstance(value, typ):
        msg += (". Note that Cython is deliberately stricter than PEP-484 and rejects subclasses of builtin types. If you need to "
                "pass subclasses then set the 'annotation_typing' directive to False.")
    raise TypeError(msg)


def _c_double(value) -> float:
    """Conversion of a Python object to a C double (``__pyx_PyFloat_AsDouble``)."""
    if isinstance(value, float):
        return float(value)
    t = type(value)
    if hasattr(t, "__float__") or hasattr(t, "__index__"):
        return float(value)
    raise TypeError(f"must be real number, not {t.__name__}")


def _readonly(name, owner):
    """Setter of a read-only property: the message of the reference's Cython getset descriptor."""
    def refuse(self, value):
        raise AttributeError(f"attribute '{name}' of 'ldpc.bp_decoder._bp_decoder.{owner}' objects is not writable")
    return refuse


class BpDecoderBase:
    """Parameter/validation surface shared by the BP decoder family (``cdef class BpDecoderBase``, pyx:82-579)."""

    def __init__(self, pcm, **kwargs):
        error_rate = kwargs.get("error_rate", None)  # pyx:90-100
        error_channel = kwargs.get("error_channel", None)
        max_iter = kwargs.get("max_iter", 0)
        bp_method = kwargs.get("bp_method", 0)
        ms_scaling_factor = kwargs.get("ms_scaling_factor", 1.0)
        schedule = kwargs.get("schedule", 0)
        omp_thread_count = kwargs.get("omp_thread_count", 1)
        random_serial_schedule = kwargs.get("random_serial_schedule", False)
        random_schedule_seed = kwargs.get("random_schedule_seed", 0)
        serial_schedule_order = kwargs.get("serial_schedule_order", None)
        channel_probs = kwargs.get("channel_probs", [None])

        self._engine = None
        self._engine_key = None
        self._device = kwargs.get("_device", -1)
        # additive: device_ids=[...] shards every decode_batch over those GPUs inside this process (ldpc_hip_bp_multi)
        self._device_ids = kwargs.get("device_ids", None)
        if self._device_ids is not None:
            self._device_ids = [int(d) for d in self._device_ids]
            if not self._device_ids or any(d < 0 for d in self._device_ids):
                raise ValueError("device_ids must be a non-empty list of GPU ordinals")
        # "cython": NumPy inputs go through the Cython binding of the C++ host class (ldpc_amd/bp_decoder/_bp_core.pyx,
        # the reference's own binding style); "ctypes": everything through ldpc_amd/engine.py.  Same C ABI underneath.
        self._backend = kwargs.get("_backend", None)
        self._cy = None

        self._h = _ingest(pcm)
        self.m, self.n = int(pcm.shape[0]), int(pcm.shape[1])

        # state of the C++ object (`new BpDecoderCpp(...)`, pyx:132 / bp.hpp:77-132)
        self._channel_probs = np.zeros(self.n, np.float64)
        self._channel_dirty = True
        self._max_iter = 0
        self._bp_method = PRODUCT_SUM
        self._schedule = PARALLEL
        self._ms_scaling_factor = 1.0
        self._omp_thread_count = 1
        self._serial_schedule_order = np.arange(self.n, dtype=np.int64)  # bp.hpp:120-124
        self._random_schedule_seed = 0
        self._seed_epoch = 0
        self._random_serial_schedule = False
        self._bp_input_type = SYNDROME
        self._decoding = np.zeros(self.n, np.uint8)
        self._log_prob_ratios = np.zeros(self.n, np.float64)
        self._iterations = 0
        self._converge = False
        # batch views (additive API)
        self.converge_batch = None
        self.iter_batch = None
        self.log_prob_ratios_batch = None

        self.bp_method = bp_method  # pyx:135-142
        self.max_iter = max_iter
        self.ms_scaling_factor = ms_scaling_factor
        self.schedule = schedule
        self.serial_schedule_order = serial_schedule_order
        self.random_schedule_seed = random_schedule_seed
        self.omp_thread_count = omp_thread_count
        self.random_serial_schedule = random_serial_schedule

        if isinstance(channel_probs, (list, np.ndarray)):  # ldpc_v1 compatibility, pyx:145-147
            if len(channel_probs) > 0 and channel_probs[0] is not None:
                error_channel = channel_probs

        if error_channel is not None:
            self.error_channel = error_channel
        elif error_rate is not None:
            self.error_rate = error_rate
        else:  # pyx:153-155 (the reference forgets the f-prefix; the literal braces are its message)
            raise ValueError("Please specify the error channel. Either: 1) error_rate: float or 2) error_channel:\
            list of floats of length equal to the block length of the code {self.n}.")

    # ---- channel (pyx:167-233) ------------------------------------------------------------------
    @property
    def error_rate(self) -> np.ndarray:
        return self._channel_probs.astype(float).copy()

    @error_rate.setter
    def error_rate(self, value: Optional[float]) -> None:
        _typed("value", value, float)
        if value is not None:
            if not isinstance(value, float):
                raise ValueError("The `error_rate` parameter must be specified as a single float value.")
            self._channel_probs[:] = value
            self._channel_dirty = True

    @property
    def error_channel(self) -> np.ndarray:
        return self._channel_probs.astype(float).copy()

    @error_channel.setter
    def error_channel(self, value) -> None:
        if value is not None:
            if len(value) != self.n:
                raise ValueError(f"The error channel vector must have length {self.n}, not {len(value)}.")
            self._channel_dirty = True
            for i in range(self.n):  # element by element into the C++ vector (pyx:222-223): a bad element stops HERE
                self._channel_probs[i] = _c_double(value[i])

    def update_channel_probs(self, value) -> None:
        self.error_channel = value

    @property
    def channel_probs(self) -> np.ndarray:
        return self._channel_probs.astype(float).copy()

    channel_probs = channel_probs.setter(_readonly("channel_probs", "BpDecoderBase"))

    # ---- input vector type (pyx:236-276) --------------------------------------------------------
    @property
    def input_vector_type(self) -> str:
        return {SYNDROME: "syndrome", RECEIVED_VECTOR: "received_vector", AUTO: "auto"}[self._bp_input_type]

    @input_vector_type.setter
    def input_vector_type(self, input_type: str):
        _typed("input_type", input_type, str, optional=False)
        key = input_type.lower()
        if key in ("auto", "a", "2"):
            if self.m == self.n:
                raise ValueError("Please specify the input vector type. Either: 1) input_vector_type: 'syndrome' or 2) input_vector_type:\
                'received_vector'.")
            self._bp_input_type = AUTO
        elif key in ("syndrome", "s", "0"):
            self._bp_input_type = SYNDROME
        elif key in ("received_vector", "r", "1"):
            self._bp_input_type = RECEIVED_VECTOR
        else:
            raise ValueError(f"The input vector type '{input_type}' is invalid. \
                    Please choose from the following methods: \
                    'input_vector_type=syndrome', 'input_vector_type=received_vector'")

    # ---- results (pyx:279-329) ------------------------------------------------------------------
    @property
    def log_prob_ratios(self) -> np.ndarray:
        return np.array(self._log_prob_ratios, dtype=np.float64)

    @property
    def converge(self) -> bool:
        return bool(self._converge)

    @property
    def iter(self) -> int:
        return int(self._iterations)

    @property
    def check_count(self) -> int:
        return self.m

    @property
    def bit_count(self) -> int:
        return self.n

    log_prob_ratios = log_prob_ratios.setter(_readonly("log_prob_ratios", "BpDecoderBase"))
    converge = converge.setter(_readonly("converge", "BpDecoderBase"))
    iter = iter.setter(_readonly("iter", "BpDecoderBase"))
    check_count = check_count.setter(_readonly("check_count", "BpDecoderBase"))
    bit_count = bit_count.setter(_readonly("bit_count", "BpDecoderBase"))

    # ---- algorithm parameters (pyx:332-579) -----------------------------------------------------
    @property
    def max_iter(self) -> int:
        return self._max_iter

    @max_iter.setter
    def max_iter(self, value: int) -> None:
        _typed("value", value, int, optional=False)
        if not isinstance(value, int):
            raise ValueError("max_iter input parameter is invalid. This must be specified as a positive int.")
        if value < 0:
            raise ValueError(f"max_iter input parameter must be a positive int. Not {value}.")
        self._max_iter = value if value != 0 else self.n  # pyx:357

    @property
    def bp_method(self) -> str:
        return "product_sum" if self._bp_method == PRODUCT_SUM else "minimum_sum"

    @bp_method.setter
    def bp_method(self, value: Union[str, int]) -> None:
        self._bp_method = _bp_method_code(value)

    @property
    def schedule(self) -> str:
        return {PARALLEL: "parallel", SERIAL: "serial", SERIAL_RELATIVE: "serial_relative"}[self._schedule]

    @schedule.setter
    def schedule(self, value: Union[str, int]) -> None:
        key = str(value).lower()
        if key in ("parallel", "p", "0"):
            self._schedule = PARALLEL
        elif key in ("serial", "s", "1"):
            self._schedule = SERIAL
        elif key in ("serial_relative", "sr", "2"):
            self._schedule = SERIAL_RELATIVE
        else:
            raise ValueError(f"The BP schedule method '{value}' is invalid. \
                    Please choose from the following methods: \
                    'schedule=parallel', 'schedule=serial', 'schedule=serial_relative'")

    @property
    def serial_schedule_order(self):
        if self._serial_schedule_order is None or len(self._serial_schedule_order) == 0:
            return None
        return np.array(self._serial_schedule_order, dtype=int)

    @serial_schedule_order.setter
    def serial_schedule_order(self, value) -> None:
        if value is None:
            return
        if not len(value) == self.n:
            raise Exception("Input error. The `serial_schedule_order` input parameter must have length equal to the length of the code.")
        for i in range(self.n):
            if not isinstance(value[i], (int, np.int64, np.int32)) or value[i] < 0 or value[i] >= self.n:
                raise ValueError(f"serial_schedule_order[{i}] is invalid. It must be a non-negative integer less than {self.n}.")
        self._serial_schedule_order = np.asarray(value, dtype=np.int64).copy()
        self.random_serial_schedule = False

    @property
    def ms_scaling_factor(self) -> float:
        return self._ms_scaling_factor

    @ms_scaling_factor.setter
    def ms_scaling_factor(self, value: float) -> None:
        value = _c_double(value)
        if not isinstance(value, (float, int)):
            raise TypeError("The ms_scaling factor must be specified as a float")
        self._ms_scaling_factor = float(value)

    @property
    def omp_thread_count(self) -> int:
        if self._omp_thread_count != 1:
            warnings.warn("The OpenMP functionality is not yet implemented")
        return self._omp_thread_count

    @omp_thread_count.setter
    def omp_thread_count(self, value: int) -> None:
        _typed("value", value, int, optional=False)
        if not isinstance(value, int) or value < 1:
            raise TypeError("The omp_thread_count must be specified as a\
            positive integer.")
        self._omp_thread_count = value
        if self._omp_thread_count != 1:
            warnings.warn("The OpenMP functionality is not yet implemented")

    @property
    def random_schedule_seed(self) -> int:
        return self._random_schedule_seed

    @random_schedule_seed.setter
    def random_schedule_seed(self, value: int) -> None:
        _typed("value", value, int, optional=False)
        if not isinstance(value, int) or value < -2:
            raise ValueError("The value of random_schedule_seed must\
            be a positive integer. Set as -1 to disable to the random\
            schedule. Set as 0 to use the system clock.")
        self._random_serial_schedule = True  # pyx:551 (the constructor resets it right after, pyx:142)
        self._random_schedule_seed = value
        self._seed_epoch = getattr(self, "_seed_epoch", 0) + 1  # every call re-seeds the generator (bp.hpp:142-145)

    @property
    def random_serial_schedule(self) -> bool:
        return self._random_serial_schedule

    @random_serial_schedule.setter
    def random_serial_schedule(self, value: bool) -> None:
        self._random_serial_schedule = bool(value)  # a C truth value

    # ---- device engine --------------------------------------------------------------------------
    def _get_engine(self):
        """Create / refresh the HIP handle lazily so that construction and validation need no GPU."""
        from ldpc_amd.engine import HipBpEngine, HipBpMultiEngine
        if self._engine is None:
            if self._device_ids is not None:
                self._engine = HipBpMultiEngine(self._h.indptr, self._h.indices, self.n, self._channel_probs,
                                                self._max_iter, self._bp_method, self._ms_scaling_factor, self._device_ids)
            else:
                self._engine = HipBpEngine(self._h.indptr, self._h.indices, self.n, self._channel_probs,
                                           self._max_iter, self._bp_method, self._ms_scaling_factor,
                                           device=self._device)
            self._channel_dirty = False
            self._engine_key = (self._max_iter, self._bp_method, self._ms_scaling_factor)
        if self._channel_dirty:  # the reference re-reads channel_probabilities on every decode (bp.hpp:149-151)
            self._engine.set_channel(self._channel_probs)
            self._channel_dirty = False
        key = (self._max_iter, self._bp_method, self._ms_scaling_factor)
        if key != self._engine_key:
            self._engine.set_params(*key)
            self._engine_key = key
        sched = (self._schedule, None if self._schedule == PARALLEL else tuple(int(v) for v in self._serial_schedule_order))
        if sched != getattr(self, "_engine_sched", (PARALLEL, None)):
            if self._schedule == PARALLEL:
                self._engine.set_schedule("parallel")
            else:  # (re)sets the object's serial_schedule_order, the state serial_relative / the random schedule work on
                self._engine.set_schedule({SERIAL: "serial", SERIAL_RELATIVE: "serial_relative"}[self._schedule],
                                          np.asarray(self._serial_schedule_order, np.int32))
            self._engine_sched = sched
        rnd = (bool(self._random_serial_schedule), self._random_schedule_seed, self._seed_epoch)
        if rnd != getattr(self, "_engine_random", (False, 0, 0)):
            self._engine.set_random_serial(rnd[0], rnd[1])  # re-seeds, as bpd.set_random_schedule_seed does (pyx:553-554)
            self._engine_random = rnd
        return self._engine

    def _get_cy(self):
        """The Cython-bound C++ host object (None if the extension is not built or the ctypes backend was requested)."""
        if self._backend == "ctypes" or self._device_ids is not None:
            return None
        if self._cy is None:
            try:
                import torch  # noqa: F401  (load torch's HIP runtime first, see ldpc_amd/_lib.py)
                from ldpc_amd.bp_decoder import _bp_core
            except ImportError:
                if self._backend == "cython":
                    raise
                self._backend = "ctypes"
                return None
            self._cy = _bp_core.CyBpCore(self._h.indptr, self._h.indices, self.n, self._channel_probs, self._max_iter,
                                         self._bp_method, self._ms_scaling_factor, self._device)
            self._cy_channel = self._channel_probs.copy()
        # push the mutable members, as the reference's setters write bpd.* directly (pyx:180-223, 342-394)
        if not np.array_equal(self._cy_channel, self._channel_probs):
            self._cy.channel_probabilities = self._channel_probs
            self._cy_channel = self._channel_probs.copy()
        self._cy.maximum_iterations = self._max_iter
        self._cy.bp_method = self._bp_method
        self._cy.ms_scaling_factor = self._ms_scaling_factor
        return self._cy

    def _decode_numpy(self, synd2d, want_llr=True, osd0=False, llr_out=None):
        """(B, m) uint8 NumPy -> (decoding, llr, iterations, converge) through the active backend.  ``llr_out``: a (B, n) float64
        C-contiguous array to receive the log-ratios instead of a new one."""
        cy = self._get_cy() if self._schedule == PARALLEL else None  # the schedule setters live on the ctypes engine
        if cy is not None:
            return cy.decode_batch(np.ascontiguousarray(synd2d, np.uint8), want_llr, osd0, llr_out)
        return self._get_engine().decode_batch(synd2d, want_llr=want_llr, osd0=osd0, llr_out=llr_out)

    # Whether ``decode_batch`` may overwrite the log-ratio array it handed out last time (same shape).  Off by default: ownership is never
    # inferred (reference counts are an interpreter detail -- CPython 3.14 borrows stack references, other interpreters have none).  A
    # caller who is done with ``log_prob_ratios_batch`` when the next call starts sets this (or passes ``reuse_log_prob_ratios=True`` /
    # its own array as ``log_prob_ratios_out``): at 65 536 x 10 000 the array is 5.2 GB, and a new one per call costs 1.3 million
    # first-touch faults going in and a 0.27 s ``munmap`` of the old one.
    recycle_log_prob_ratios = False

    def _llr_destination(self, rows: int, out=None, reuse=None):
        """Where a batch's log-ratios go: the caller's array; else, if the caller SAID the previous batch's array may be overwritten, that
        array -- or, when there is none yet and it would be 256 MiB or more, a new one on page-locked memory (``ldpc_hip_host_alloc``: the
        device-to-host copies write it directly, no staging buffer, no host-side copy); else None (the backend makes an ordinary array)."""
        if out is not None:
            if not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.shape == (rows, self.n) and out.flags.c_contiguous and out.flags.writeable):
                raise ValueError(f"log_prob_ratios_out must be a writeable C-contiguous float64 array of shape ({rows}, {self.n}).")
            return out
        if self.recycle_log_prob_ratios if reuse is None else reuse:
            old = getattr(self, "log_prob_ratios_batch", None)
            if isinstance(old, np.ndarray) and old.dtype == np.float64 and old.shape == (rows, self.n) and old.flags.c_contiguous and old.flags.writeable:
                return old
            # nothing to reuse yet: the caller has said the array will be written again and again -- that is when page-locked memory pays
            # (a caller on the default path gets an ordinary array: a pinned block used once as an ordinary array is a waste of a limited resource)
            nbytes = rows * self.n * 8
            if nbytes >= (256 << 20):
                from .._lib import PinnedBlock
                blk = PinnedBlock.try_new(nbytes)
                if blk is not None:
                    return blk.array((rows, self.n), np.float64)
        return None

    def _require_parallel(self):
        """Every schedule of the reference runs on the device: 'parallel' (bp.hpp:192-325), 'serial' (bp.hpp:451-545) with a
        fixed order, and the two that keep state in the decoder object -- 'serial_relative' and random_serial_schedule
        (bp.hpp:467-483; include/ldpc_hip.h: ldpc_hip_bp_set_schedule).  For those, ``decode`` calls follow one another as on
        a reference object (the order / generator carry over); the rows of a ``decode_batch`` each start from the state at the
        time of the call."""
        return None

    def _schedule_keeps_state(self) -> bool:
        return self._schedule != PARALLEL and (self._schedule == SERIAL_RELATIVE or bool(self._random_serial_schedule))

    def _pull_schedule_state(self):
        """After a decode with a schedule that rearranges ``serial_schedule_order`` (bp.hpp:467-483): read it back, as the
        reference's property shows the decoder object's rearranged member."""
        if self._schedule_keeps_state() and self._engine is not None and hasattr(self._engine, "schedule_order"):
            self._serial_schedule_order = self._engine.schedule_order().astype(np.int64)
            self._engine_sched = (self._schedule, tuple(int(v) for v in self._serial_schedule_order))


def _zero_rows(vec: np.ndarray) -> np.ndarray:
    """(B,) bool: rows of a C-contiguous (B, w) uint8 array that are all zero (eight bytes at a time where the row length allows)."""
    if vec.shape[1] and vec.shape[1] % 8 == 0:
        return ~vec.view(np.uint64).any(axis=1)
    return ~vec.any(axis=1)


class BpDecoder(BpDecoderBase):
    """Belief-propagation decoder for binary linear codes (drop-in for ``ldpc.BpDecoder``, pyx:581-709)."""

    def __init__(self, pcm, *, error_rate=_UNSET, error_channel=_UNSET, max_iter=_UNSET, bp_method=_UNSET,
                 ms_scaling_factor=_UNSET, schedule=_UNSET, omp_thread_count=_UNSET, random_schedule_seed=_UNSET,
                 serial_schedule_order=_UNSET, input_vector_type: str = "auto", random_serial_schedule=_UNSET,
                 **kwargs):
        """Keywords as in the reference (pyx:620-623): ``error_rate: float``, ``error_channel``,
        ``max_iter: int = 0`` (0 -> n), ``bp_method: str`` ('product_sum' | 'minimum_sum' and aliases),
        ``ms_scaling_factor = 1.0``, ``schedule = 'parallel'``, ``omp_thread_count = 1``,
        ``random_schedule_seed = 0``, ``serial_schedule_order = None``, ``input_vector_type = 'auto'``,
        ``random_serial_schedule = False``, ``channel_probs`` (ldpc_v1 alias of ``error_channel``).

        Only keywords the caller actually passes reach the base initialiser -- in the reference Cython
        forwards the call's own kwargs to ``BpDecoderBase.__cinit__``, whose ``kwargs.get`` defaults
        (pyx:90-100) therefore win over the signature's: an omitted ``bp_method`` means product_sum
        (pinned by python_test/test_bp_decoder.py:121-136).
        """
        _check_pcm_type(pcm)
        given = dict(error_rate=error_rate, error_channel=error_channel, max_iter=max_iter, bp_method=bp_method,
                     ms_scaling_factor=ms_scaling_factor, schedule=schedule, omp_thread_count=omp_thread_count,
                     random_schedule_seed=random_schedule_seed, serial_schedule_order=serial_schedule_order,
                     random_serial_schedule=random_serial_schedule)
        passed = dict(kwargs)
        passed.update({k: v for k, v in given.items() if v is not _UNSET})
        # order of events in the reference: the base class's __cinit__ runs first, with every keyword (its setters raise
        # their own errors); then this class's signature is type-checked; then its body (pyx:620-629)
        super().__init__(pcm, **passed)
        _typed("error_rate", passed.get("error_rate"), float)
        _typed("max_iter", passed.get("max_iter", 0), int)
        _typed("bp_method", passed.get("bp_method", "minimum_sum"), str)
        _typed("schedule", passed.get("schedule", "parallel"), str)
        _typed("omp_thread_count", passed.get("omp_thread_count", 1), int)
        _typed("random_schedule_seed", passed.get("random_schedule_seed", 0), int)
        _typed("serial_schedule_order", passed.get("serial_schedule_order"), list)
        _typed("input_vector_type", input_vector_type, str, optional=False)
        for key in kwargs.keys():  # pyx:625-627
            if key not in ["channel_probs", "_device", "_backend", "device_ids"]:
                raise ValueError(f"Unknown parameter '{key}' passed to the BpDecoder constructor.")
        self.input_vector_type = input_vector_type  # pyx:629

    # ---- single input, reference signature (pyx:642-695) ----------------------------------------
    def decode(self, input_vector: np.ndarray) -> np.ndarray:
        _typed("input_vector", input_vector, np.ndarray, optional=False)
        ln = len(input_vector)
        if self._bp_input_type == SYNDROME and not ln == self.m:
            raise ValueError(f"The input_vector must have length {self.m} (for syndrome decoding). Not length {ln}.")
        elif self._bp_input_type == RECEIVED_VECTOR and not ln == self.n:
            raise ValueError(f"The input_vector must have length {self.n} (for received vector decoding). Not length {ln}.")
        elif self._bp_input_type == AUTO and not (ln == self.m or ln == self.n):
            raise ValueError(f"The input_vector must have length {self.m} (for syndrome decoding) or length {self.n} (for received vector decoding). Not length {ln}.")
        dtype = input_vector.dtype
        vec = np.asarray(input_vector).astype(np.uint8)  # element-wise uint8 narrowing, pyx:676-678
        if not vec.any():  # pyx:679-681 / 688-690
            self._converge = True
            return np.zeros(self.n, dtype=dtype)
        self._require_parallel()
        as_syndrome = self._bp_input_type == SYNDROME or (self._bp_input_type == AUTO and ln == self.m)
        if as_syndrome:
            dec, llr, it, cv = self._decode_numpy(vec[None, :])
            out = dec[0]
        else:  # bp.hpp:162-180: decode H r, then XOR the received vector back in
            synd = self._get_engine().mulvec_batch(vec[None, :])
            dec, llr, it, cv = self._decode_numpy(synd)
            out = dec[0] ^ vec
        self._decoding = out.astype(np.uint8)
        self._log_prob_ratios = llr[0]
        self._iterations = int(it[0])
        self._converge = bool(cv[0])
        self._pull_schedule_state()
        return out.astype(dtype)

    # ---- batch (additive) -----------------------------------------------------------------------
    def decode_batch(self, input_vectors, want_log_prob_ratios: bool = True, *, log_prob_ratios_out=None, reuse_log_prob_ratios=None):
        """Decode every row of a 2-D array in one launch.

        ``log_prob_ratios_out`` (NumPy inputs): a ``(B, n)`` float64 C-contiguous array that receives the log-ratios;
        ``reuse_log_prob_ratios=True`` (or the attribute ``recycle_log_prob_ratios``): the array the previous call handed out as
        ``log_prob_ratios_batch`` may be overwritten by this one -- nothing is reused unless the caller says so.

        ``input_vectors``: ``(B, m)`` syndromes or ``(B, n)`` received vectors (same rule as
        ``decode``), NumPy array (any integer dtype; result has the same dtype) or a torch CUDA uint8
        tensor (results stay in HBM as torch tensors).  Row ``b`` of the result equals
        ``decode(input_vectors[b])`` of the reference, including the all-zero shortcut.  Afterwards
        ``converge_batch`` (B,) bool, ``iter_batch`` (B,) int32 (0 for shortcut rows) and
        ``log_prob_ratios_batch`` (B, n) float64 (zeros for shortcut rows) describe every row and
        the scalar properties describe the last row that actually ran BP.
        """
        from ldpc_amd.engine import _is_torch
        if input_vectors.ndim != 2:
            raise ValueError("decode_batch expects a 2-D array of shape (batch, m) or (batch, n).")
        ln = int(input_vectors.shape[1])
        if self._bp_input_type == SYNDROME and not ln == self.m:
            raise ValueError(f"The input_vector must have length {self.m} (for syndrome decoding). Not length {ln}.")
        elif self._bp_input_type == RECEIVED_VECTOR and not ln == self.n:
            raise ValueError(f"The input_vector must have length {self.n} (for received vector decoding). Not length {ln}.")
        elif self._bp_input_type == AUTO and not (ln == self.m or ln == self.n):
            raise ValueError(f"The input_vector must have length {self.m} (for syndrome decoding) or length {self.n} (for received vector decoding). Not length {ln}.")
        self._require_parallel()
        as_syndrome = self._bp_input_type == SYNDROME or (self._bp_input_type == AUTO and ln == self.m)
        eng = self._get_engine()
        if _is_torch(input_vectors):
            import torch
            if log_prob_ratios_out is not None or reuse_log_prob_ratios:
                raise ValueError("log_prob_ratios_out / reuse_log_prob_ratios apply to NumPy inputs only: with device tensors the log-ratios "
                                 "are returned as a device tensor (log_prob_ratios_batch).")
            vec = input_vectors
            synd = vec if as_syndrome else eng.mulvec_batch(vec)
            dec, llr, it, cv = eng.decode_batch(synd, want_llr=want_log_prob_ratios)
            if not as_syndrome:
                dec ^= vec
            zero = vec.any(dim=1).logical_not()  # all-zero shortcut rows (pyx:679-681); uint8.any() is uint8
            if bool(zero.any()):
                dec[zero] = 0
                cv[zero] = 1
                it[zero] = 0
                if llr is not None:
                    llr[zero] = 0
            self.converge_batch, self.iter_batch, self.log_prob_ratios_batch = cv.bool(), it, llr
            self._pull_schedule_state()
            return dec
        dtype = input_vectors.dtype
        vec = np.ascontiguousarray(np.asarray(input_vectors).astype(np.uint8, copy=False))
        synd = vec if as_syndrome else eng.mulvec_batch(vec)
        # which rows take the all-zero shortcut (pyx:679-681): a scan of the whole input -- on a helper thread for large batches, while
        # the device decodes (the C call releases the GIL)
        zero_box = []
        scan = None
        if vec.size >= (1 << 24):
            import threading
            scan = threading.Thread(target=lambda: zero_box.append(_zero_rows(vec)))
            scan.start()
        try:
            dec, llr, it, cv = self._decode_numpy(synd, want_llr=want_log_prob_ratios,
                                                  llr_out=self._llr_destination(len(synd), log_prob_ratios_out, reuse_log_prob_ratios) if want_log_prob_ratios else None)
        finally:
            if scan is not None:
                scan.join()
        if not as_syndrome:
            dec ^= vec
        zero = zero_box[0] if zero_box else _zero_rows(vec)
        if zero.any():
            dec[zero] = 0
            cv[zero] = True
            it[zero] = 0
            if llr is not None:
                llr[zero] = 0.0
        self.converge_batch, self.iter_batch, self.log_prob_ratios_batch = cv, it, llr
        ran = np.flatnonzero(~zero)
        if len(ran):
            last = int(ran[-1])
            self._decoding = dec[last].astype(np.uint8)
            if llr is not None:
                self._log_prob_ratios = llr[last].copy()  # (a copy, not a view: a view would pin the whole batch array)
            self._iterations = int(it[last])
        if len(vec):
            self._converge = bool(cv[-1])
        self._pull_schedule_state()
        return dec.astype(dtype, copy=False)

    @property
    def decoding(self) -> np.ndarray:
        return np.array(self._decoding).astype(int)  # pyx:698-709

    decoding = decoding.setter(_readonly("decoding", "BpDecoder"))


class SoftInfoBpDecoder(BpDecoderBase):
    """Serial minimum-sum BP with analog syndrome information (drop-in for ``ldpc.bp_decoder.SoftInfoBpDecoder``,
    pyx:712-812; algorithm ``soft_info_decode_serial``, bp.hpp:547-660).

    Keywords as in the reference (pyx:743-745): ``error_rate``, ``error_channel``, ``max_iter = 0`` (-> n),
    ``bp_method`` (ignored: always minimum_sum, pyx:752), ``ms_scaling_factor = 1.0``, ``cutoff = inf``,
    ``sigma = 2.0``.  ``decode`` takes one vector of ``m`` analog readouts; ``decode_batch`` (additive) takes ``(B, m)``.
    """

    def __init__(self, pcm, *, error_rate=_UNSET, error_channel=_UNSET, max_iter=_UNSET, bp_method=_UNSET,
                 ms_scaling_factor=_UNSET, cutoff=np.inf, sigma: float = 2.0, **kwargs):
        _check_pcm_type(pcm)
        given = dict(error_rate=error_rate, error_channel=error_channel, max_iter=max_iter, bp_method=bp_method,
                     ms_scaling_factor=ms_scaling_factor)
        passed = dict(kwargs)
        passed.update({k: v for k, v in given.items() if v is not _UNSET})
        super().__init__(pcm, **passed)
        _typed("error_rate", passed.get("error_rate"), float)  # the signature of pyx:743-745, checked after the base class ran
        _typed("error_channel", passed.get("error_channel"), list)
        _typed("max_iter", passed.get("max_iter", 0), int)
        _typed("bp_method", passed.get("bp_method", "minimum_sum"), str)
        _typed("ms_scaling_factor", passed.get("ms_scaling_factor", 1.0), float)
        _typed("cutoff", cutoff, float)
        sigma = _c_double(sigma)
        self.cutoff = cutoff
        if not isinstance(sigma, float) or sigma <= 0:  # pyx:748-749
            raise ValueError("The sigma value must be a float greater than 0.")
        self.sigma = sigma
        self.schedule = "serial"
        self.bp_method = "minimum_sum"
        self.input_vector_type = "syndrome"
        self._soft_syndrome = np.zeros(self.m, np.float64)
        self.soft_syndrome_batch = None

    def _soft_decode(self, soft2d):
        """With random_serial_schedule the routine rearranges the order the object carries at the top of every iteration it runs
        (bp.hpp:573-577: ``shuffle(order, std::default_random_engine(random_schedule_seed))``, a new engine each time); the engine
        does the same draws, every row of a batch from the order at the time of the call, and leaves its last row's order."""
        out = self._get_engine().soft_info_decode_batch(soft2d, self.cutoff, self.sigma)
        self._pull_schedule_state()
        return out

    def decode(self, soft_info_syndrome: np.ndarray) -> np.ndarray:
        """One analog syndrome (pyx:761-785); returns the decoding as uint8."""
        soft = np.asarray([soft_info_syndrome[i] for i in range(self.m)], dtype=np.float64)  # pyx:776-779
        dec, llr, it, cv, so = self._soft_decode(soft[None, :])
        self._decoding = dec[0].copy()
        self._log_prob_ratios = llr[0]
        self._iterations = int(it[0])
        self._converge = bool(cv[0])
        self._soft_syndrome = so[0]
        return dec[0].astype(np.uint8)

    def decode_batch(self, soft_info_syndromes):
        """``(B, m)`` analog syndromes in one launch; row b equals ``decode(soft_info_syndromes[b])``.  Afterwards
        ``converge_batch``, ``iter_batch``, ``log_prob_ratios_batch`` and ``soft_syndrome_batch`` describe every row."""
        if soft_info_syndromes.ndim != 2 or soft_info_syndromes.shape[1] != self.m:
            raise ValueError(f"The soft syndromes must have shape (batch, {self.m}).")
        from ldpc_amd.engine import _is_torch
        soft = soft_info_syndromes if _is_torch(soft_info_syndromes) else np.ascontiguousarray(soft_info_syndromes, np.float64)
        dec, llr, it, cv, so = self._soft_decode(soft)
        self.converge_batch, self.iter_batch, self.log_prob_ratios_batch, self.soft_syndrome_batch = cv, it, llr, so
        return dec

    @property
    def soft_syndrome(self) -> np.ndarray:
        return np.array(self._soft_syndrome, dtype=np.float64)

    @property
    def decoding(self) -> np.ndarray:
        return np.array(self._decoding).astype(int)
