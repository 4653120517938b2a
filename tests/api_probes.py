"""Probes of the Python surface of ``ldpc.BpDecoder`` and friends.

One list of declarative probes, one executor.  ``tests/golden/make_golden_api.py`` runs them against the REAL reference
(its Cython modules built in a scratch directory) and stores what came back -- value, or exception type + message, plus
warnings -- in ``tests/golden/api_reference.json``; ``tests/test_api_fixture.py`` runs the same probes against ``ldpc_amd`` and
compares.  The probes restate what the reference's own tests exercise (python_test/test_bp_decoder.py:94-211,
test_bp_decoder_input.py, test_scipy_helpers.py) and go further: every keyword alias (pyx:376-435), every setter's bad-type /
bad-value case (pyx:167-579), every ingest container and dtype, and decode()'s length / dtype / zero-shortcut rules (pyx:642-695).

A probe = {"id", "cls", "pcm", "kwargs", "ops", "gpu"}: construct ``cls(pcm, **kwargs)``, then apply ``ops`` in order:
  ["get", attr]                 read a property
  ["set", attr, value]          assign a property
  ["call", method, [args...]]   call a method
"gpu": True marks probes whose ops run the decoder on a non-zero input (the mirror needs the device for those).
Values that are not JSON (arrays, numpy scalars, matrices) are written as {"nd": nested list, "dtype": name} etc.
"""
from __future__ import annotations

import warnings

import numpy as np
import scipy.sparse as sp

# ---- inputs ------------------------------------------------------------------------------------------------------
H7 = np.array([[0, 0, 0, 1, 1, 1, 1], [0, 1, 1, 0, 0, 1, 1], [1, 0, 1, 0, 1, 0, 1]], dtype=np.uint8)  # hamming_code(3)
REP5 = np.array([[1, 1, 0, 0, 0], [0, 1, 1, 0, 0], [0, 0, 1, 1, 0], [0, 0, 0, 1, 1]], dtype=np.uint8)  # rep_code(5)
SQ3 = np.array([[1, 1, 0], [0, 1, 1], [1, 0, 1]], dtype=np.uint8)  # m == n


def _with_explicit_zero():
    m = sp.csr_matrix(H7)
    m.data[2] = 0  # stored zero
    return m


def _with_duplicates():
    rows, cols = np.nonzero(H7)
    return sp.coo_matrix((np.ones(len(rows) + 1, np.uint8), (np.append(rows, rows[0]), np.append(cols, cols[0]))), shape=H7.shape)


def _big_values():
    a = H7.astype(np.int64)
    a[0, 0] = 256  # narrows to 0 in uint8
    return a


PCMS = {
    "h7_u8": lambda: H7.copy(),
    "h7_i8": lambda: H7.astype(np.int8),
    "h7_int": lambda: H7.astype(int),
    "h7_f64": lambda: H7.astype(float),
    "h7_f32": lambda: H7.astype(np.float32),
    "h7_bool": lambda: H7.astype(bool),
    "h7_u16": lambda: H7.astype(np.uint16),
    "h7_i32": lambda: H7.astype(np.int32),
    "h7_fortran": lambda: np.asfortranarray(H7),
    "h7_csr": lambda: sp.csr_matrix(H7),
    "h7_csc": lambda: sp.csc_matrix(H7),
    "h7_coo": lambda: sp.coo_matrix(H7),
    "h7_lil": lambda: sp.lil_matrix(H7),
    "h7_csr_f64": lambda: sp.csr_matrix(H7.astype(float)),
    "h7_csr_f32": lambda: sp.csr_matrix(H7.astype(np.float32)),
    "h7_csr_int": lambda: sp.csr_matrix(H7.astype(int)),
    "h7_csr_i8": lambda: sp.csr_matrix(H7.astype(np.int8)),
    "h7_csr_zero": _with_explicit_zero,
    "h7_coo_dup": _with_duplicates,
    "h7_list": lambda: H7.tolist(),
    "h7_tuple": lambda: tuple(map(tuple, H7.tolist())),
    "h7_matrix": lambda: np.asmatrix(H7),
    "h7_two": lambda: (H7 * 2).astype(np.uint8),
    "h7_neg": lambda: -H7.astype(np.int8),
    "h7_half": lambda: H7.astype(float) * 0.5,
    "h7_256": _big_values,
    "h7_csr_two": lambda: sp.csr_matrix((H7 * 2).astype(np.uint8)),
    "rep5": lambda: REP5.copy(),
    "rep5_csr": lambda: sp.csr_matrix(REP5),
    "sq3": lambda: SQ3.copy(),
    "empty_row": lambda: np.array([[1, 1, 0], [0, 0, 0]], dtype=np.uint8),
    "string": lambda: "not a matrix",
    "none": lambda: None,
    "vector": lambda: np.array([1, 0, 1], dtype=np.uint8),
}


def ND(values, dtype="float64"):
    return {"nd": values, "dtype": dtype}


def _decode_arg(v):
    if isinstance(v, dict) and "nd" in v:
        return np.array(v["nd"], dtype=v["dtype"])
    if isinstance(v, dict) and "pcm" in v:
        return PCMS[v["pcm"]]()
    if isinstance(v, list):
        return [_decode_arg(x) for x in v]
    return v


def _encode(v):
    if isinstance(v, np.ndarray):
        return {"nd": v.tolist(), "dtype": str(v.dtype), "shape": list(v.shape)}
    if sp.issparse(v):
        c = sp.coo_matrix(v)
        order = np.lexsort((c.col, c.row))
        return {"sparse": type(v).__name__, "shape": list(v.shape), "dtype": str(v.dtype), "row": c.row[order].tolist(),
                "col": c.col[order].tolist(), "data": c.data[order].tolist()}
    if isinstance(v, (np.integer,)):
        return {"np": int(v), "type": type(v).__name__}
    if isinstance(v, (np.floating,)):
        return {"np": float(v), "type": type(v).__name__}
    if isinstance(v, (np.bool_,)):
        return {"np": bool(v), "type": "bool_"}
    if isinstance(v, float) and v != v:
        return {"nan": True}
    if isinstance(v, (list, tuple)):
        return [_encode(x) for x in v]
    if v is None or isinstance(v, (bool, int, float, str)):
        return v
    return {"repr": repr(v)}


def _outcome(fn):
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        try:
            out = {"v": _encode(fn())}
        except BaseException as exc:  # noqa: BLE001 -- the exception IS the result
            out = {"exc": type(exc).__name__, "msg": str(exc)}
    w = [[type(c.message).__name__, str(c.message)] for c in caught]
    if w:
        out["warnings"] = w
    return out


def run_probe(probe, namespace):
    """``namespace``: {"BpDecoder": cls, "BpOsdDecoder": cls, "SoftInfoBpDecoder": cls, "convert_to_binary_sparse": fn, "io_test": fn}."""
    target = namespace[probe["cls"]]
    res = {"id": probe["id"]}
    kwargs = {k: _decode_arg(v) for k, v in probe.get("kwargs", {}).items()}
    if probe.get("function"):
        res["call"] = _outcome(lambda: target(PCMS[probe["pcm"]]()))
        return res
    holder = {}
    pcm = PCMS[probe["pcm"]]()

    def construct():
        holder["obj"] = target(pcm, **kwargs)
        return None
    res["ctor"] = _outcome(construct)
    if "obj" not in holder:
        return res
    obj = holder["obj"]
    steps = []
    for op in probe.get("ops", []):
        if op[0] == "get":
            steps.append(_outcome(lambda: getattr(obj, op[1])))
        elif op[0] == "set":
            val = _decode_arg(op[2])
            steps.append(_outcome(lambda: setattr(obj, op[1], val)))
        elif op[0] == "call":
            args = [_decode_arg(a) for a in (op[2] if len(op) > 2 else [])]
            steps.append(_outcome(lambda: getattr(obj, op[1])(*args)))
        else:
            raise ValueError(op)
    res["steps"] = steps
    return res


# ---- the probes --------------------------------------------------------------------------------------------------
_STATE = [["get", "bp_method"], ["get", "max_iter"], ["get", "schedule"], ["get", "ms_scaling_factor"], ["get", "omp_thread_count"],
          ["get", "random_schedule_seed"], ["get", "random_serial_schedule"], ["get", "serial_schedule_order"], ["get", "input_vector_type"],
          ["get", "error_rate"], ["get", "error_channel"], ["get", "channel_probs"], ["get", "check_count"], ["get", "bit_count"],
          ["get", "converge"], ["get", "iter"], ["get", "log_prob_ratios"], ["get", "decoding"]]


def _probes():
    P = []

    def add(pid, cls="BpDecoder", pcm="h7_u8", kwargs=None, ops=None, gpu=False, function=False):
        P.append({"id": pid, "cls": cls, "pcm": pcm, "kwargs": kwargs or {}, "ops": ops or [], "gpu": gpu, "function": function})

    # -- ingest: every container / dtype (a2: Py2BpSparse + convert_to_binary_sparse) --
    for name in PCMS:
        add(f"ingest_{name}", pcm=name, kwargs={"error_rate": 0.1}, ops=[["get", "check_count"], ["get", "bit_count"], ["get", "max_iter"]])
        add(f"helper_{name}", cls="convert_to_binary_sparse", pcm=name, function=True)
    for name in ("h7_u8", "h7_csr", "h7_csc", "h7_coo_dup", "h7_csr_zero", "h7_f64", "h7_256", "rep5_csr", "empty_row", "h7_list", "h7_two"):
        add(f"io_test_{name}", cls="io_test", pcm=name, function=True)

    # -- constructor: defaults, every keyword and alias --
    add("ctor_defaults", kwargs={"error_rate": 0.1}, ops=_STATE)
    add("ctor_no_channel", kwargs={})
    add("ctor_error_channel", kwargs={"error_channel": [0.1, 0.2, 0.3, 0.1, 0.2, 0.3, 0.05]}, ops=_STATE[9:12])
    add("ctor_error_channel_nd", kwargs={"error_channel": ND([0.1, 0.2, 0.3, 0.1, 0.2, 0.3, 0.05])}, ops=_STATE[9:12])
    add("ctor_channel_probs_alias", kwargs={"channel_probs": [0.1, 0.2, 0.3, 0.1, 0.2, 0.3, 0.05]}, ops=_STATE[9:12])
    add("ctor_channel_probs_over_error_rate", kwargs={"error_rate": 0.4, "channel_probs": ND([0.1] * 7)}, ops=_STATE[9:12])
    add("ctor_channel_probs_none_list", kwargs={"error_rate": 0.4, "channel_probs": [None]}, ops=_STATE[9:12])
    add("ctor_channel_probs_empty", kwargs={"error_rate": 0.4, "channel_probs": []}, ops=_STATE[9:12])
    add("ctor_error_channel_short", kwargs={"error_channel": [0.1, 0.2]})
    add("ctor_error_channel_long", kwargs={"error_channel": [0.1] * 8})
    add("ctor_both_channel_and_rate", kwargs={"error_rate": 0.3, "error_channel": [0.1] * 7}, ops=_STATE[9:12])
    add("ctor_unknown_kwarg", kwargs={"error_rate": 0.1, "bogus": 1})
    add("ctor_unknown_kwarg_osd", cls="BpOsdDecoder", kwargs={"error_rate": 0.1, "bogus": 1})
    for v in (0.1, 0, 1, "0.1", None, [0.1], ND(0.1), 0.0, 1.0, -0.5, 2.5):
        add(f"ctor_error_rate_{type(v).__name__}_{v if not isinstance(v, dict) else 'nd'}", kwargs={"error_rate": v}, ops=_STATE[9:10])
    for v in (0, 1, 5, 100, -1, 1.5, "3", None, True, ND(4, "int64")):
        add(f"ctor_max_iter_{type(v).__name__}_{v if not isinstance(v, dict) else 'nd'}", kwargs={"error_rate": 0.1, "max_iter": v}, ops=[["get", "max_iter"]])
    for v in ("product_sum", "ps", "prod_sum", "prod sum", "PRODUCT_SUM", "Ps", "0", "minimum_sum", "ms", "min_sum", "min sum", "minimum sum",
              "MS", "1", "bp", "2", "", "product-sum", 0, 1, None, 2.0):
        add(f"ctor_bp_method_{v!r}", kwargs={"error_rate": 0.1, "bp_method": v}, ops=[["get", "bp_method"]])
    for v in ("parallel", "p", "1", "serial", "s", "0", "serial_relative", "sr", "2", "PARALLEL", "Serial", "flooding", "", "3", 0, 1, None):
        add(f"ctor_schedule_{v!r}", kwargs={"error_rate": 0.1, "schedule": v}, ops=[["get", "schedule"]])
    for v in (1.0, 0.625, 0, 1, 0.0, -1.0, 2, "a", "0.5", None, [0.5], ND(0.5), True):
        add(f"ctor_ms_scaling_factor_{type(v).__name__}_{v if not isinstance(v, dict) else 'nd'}", kwargs={"error_rate": 0.1, "ms_scaling_factor": v},
            ops=[["get", "ms_scaling_factor"]])
    for v in (1, 4, 0, -1, 2.0, "2", None, True):
        add(f"ctor_omp_thread_count_{type(v).__name__}_{v}", kwargs={"error_rate": 0.1, "omp_thread_count": v}, ops=[["get", "omp_thread_count"]])
    for v in (0, 7, -3, 2.5, "4", None):
        add(f"ctor_random_schedule_seed_{type(v).__name__}_{v}", kwargs={"error_rate": 0.1, "random_schedule_seed": v},
            ops=[["get", "random_schedule_seed"], ["get", "random_serial_schedule"]])
    for v in (True, False, 1, 0, "yes", None):
        add(f"ctor_random_serial_schedule_{type(v).__name__}_{v}", kwargs={"error_rate": 0.1, "random_serial_schedule": v},
            ops=[["get", "random_serial_schedule"]])
    for tag, v in (("perm", [6, 5, 4, 3, 2, 1, 0]), ("nd", ND([0, 2, 4, 6, 1, 3, 5], "int64")), ("repeat", [0, 0, 0, 0, 0, 0, 0]), ("short", [0, 1, 2]),
                   ("long", list(range(8))), ("range", [0, 1, 2, 3, 4, 5, 7]), ("neg", [0, 1, 2, 3, 4, 5, -1]), ("float", [0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0]),
                   ("none", None), ("empty", [])):
        add(f"ctor_serial_schedule_order_{tag}", kwargs={"error_rate": 0.1, "schedule": "serial", "serial_schedule_order": v},
            ops=[["get", "serial_schedule_order"], ["get", "schedule"]])
    for v in ("auto", "a", "2", "syndrome", "s", "0", "received_vector", "r", "1", "AUTO", "Syndrome", "vector", "", 0, None):
        add(f"ctor_input_vector_type_{v!r}", kwargs={"error_rate": 0.1, "input_vector_type": v}, ops=[["get", "input_vector_type"]])
        add(f"ctor_input_vector_type_square_{v!r}", pcm="sq3", kwargs={"error_rate": 0.1, "input_vector_type": v}, ops=[["get", "input_vector_type"]])
    add("ctor_square_default", pcm="sq3", kwargs={"error_rate": 0.1}, ops=[["get", "input_vector_type"]])
    add("ctor_positional_only_pcm", kwargs={"error_rate": 0.1, "max_iter": 3, "bp_method": "ms", "ms_scaling_factor": 0.5, "schedule": "serial",
                                            "omp_thread_count": 2, "random_schedule_seed": 5, "serial_schedule_order": [1, 0, 2, 3, 4, 5, 6],
                                            "input_vector_type": "syndrome", "random_serial_schedule": False}, ops=_STATE)

    # -- setters after construction: good and bad values (a13) --
    base = {"error_rate": 0.1}
    for attr, values in {
        "error_rate": [0.2, 0, 1, "0.1", None, [0.2], ND(0.3), 0.0, 1.0],
        "error_channel": [[0.2] * 7, ND([0.05] * 7), [0.2] * 6, [0.2] * 8, None, 0.2, "abcdefg", [0.1, "x", 0.1, 0.1, 0.1, 0.1, 0.1], [1, 0, 1, 0, 1, 0, 1]],
        "max_iter": [0, 1, 10, -1, -100, 1.5, "5", None, True],
        "bp_method": ["ms", "ps", "minimum_sum", "product_sum", "min sum", 1, 0, "x", None, 3],
        "schedule": ["serial", "parallel", "serial_relative", "s", "p", "sr", 0, 1, 2, "x", None],
        "ms_scaling_factor": [0.5, 0, 1, 2, -1.0, "a", None, [1.0], True, ND(0.75)],
        "omp_thread_count": [2, 1, 0, -2, 1.5, "2", None, True],
        "random_schedule_seed": [3, 0, -1, 1.5, "3", None],
        "random_serial_schedule": [True, False, 1, 0, None, "x"],
        "serial_schedule_order": [[6, 5, 4, 3, 2, 1, 0], ND([3, 2, 1, 0, 6, 5, 4], "int64"), None, [0, 1], [0, 1, 2, 3, 4, 5, 9], [0, 1, 2, 3, 4, 5, -2], "0123456",
                                  [0.0] * 7, ND([0, 1, 2, 3, 4, 5, 6], "float64")],
        "input_vector_type": ["syndrome", "received_vector", "auto", "s", "r", "a", "0", "1", "2", "x", "", 1, None],
    }.items():
        for i, v in enumerate(values):
            add(f"set_{attr}_{i}", kwargs=base, ops=[["set", attr, v], ["get", attr]] + (_STATE[9:12] if attr.startswith("error") else []))
    add("set_readonly_converge", kwargs=base, ops=[["set", "converge", True]])
    add("set_readonly_iter", kwargs=base, ops=[["set", "iter", 3]])
    add("set_readonly_check_count", kwargs=base, ops=[["set", "check_count", 3]])
    add("set_readonly_log_prob_ratios", kwargs=base, ops=[["set", "log_prob_ratios", ND([0.0] * 7)]])
    add("set_readonly_decoding", kwargs=base, ops=[["set", "decoding", ND([0] * 7, "uint8")]])
    add("set_channel_probs_readonly", kwargs=base, ops=[["set", "channel_probs", [0.2] * 7]])
    add("update_channel_probs", kwargs=base, ops=[["call", "update_channel_probs", [[0.3] * 7]], ["get", "channel_probs"],
                                                    ["call", "update_channel_probs", [[0.3] * 5]], ["call", "update_channel_probs", [ND([0.01] * 7)]],
                                                    ["get", "error_rate"]])
    add("max_iter_zero_means_n", pcm="rep5", kwargs={"error_rate": 0.1, "max_iter": 0}, ops=[["get", "max_iter"], ["set", "max_iter", 3], ["get", "max_iter"],
                                                                                                ["set", "max_iter", 0], ["get", "max_iter"]])
    add("square_then_auto", pcm="sq3", kwargs={"error_rate": 0.1, "input_vector_type": "syndrome"}, ops=[["set", "input_vector_type", "auto"], ["get", "input_vector_type"]])

    # -- decode(): lengths, dtypes, the all-zero shortcut (a12); none of these runs BP --
    for ivt in ("auto", "syndrome", "received_vector"):
        for ln in (0, 2, 3, 4, 7, 8):
            add(f"decode_zero_len{ln}_{ivt}", kwargs={"error_rate": 0.1, "input_vector_type": ivt},
                ops=[["call", "decode", [ND([0] * ln, "uint8")]], ["get", "converge"], ["get", "iter"], ["get", "decoding"]])
    for dt in ("uint8", "int8", "int32", "int64", "float64", "float32", "bool", "uint16"):
        add(f"decode_zero_dtype_{dt}", kwargs=base, ops=[["call", "decode", [ND([0] * 3, dt)]], ["get", "converge"]])
    add("decode_list_input", kwargs=base, ops=[["call", "decode", [[0, 0, 0]]]])
    add("decode_2d_input", kwargs=base, ops=[["call", "decode", [ND([[0, 0, 0]], "uint8")]]])
    add("decode_none", kwargs=base, ops=[["call", "decode", [None]]])
    add("decode_zero_osd", cls="BpOsdDecoder", kwargs={"error_rate": 0.1, "osd_method": "osd_0"},
        ops=[["call", "decode", [ND([0, 0, 0], "int64")]], ["get", "converge"], ["call", "decode", [ND([0, 0], "uint8")]], ["call", "decode", [ND([0] * 7, "uint8")]]])

    # -- BpOsdDecoder parameters (pyx:139-234) --
    for v in ("osd_0", "osd0", "0", "OSD_0", "osd_e", "e", "exhaustive", "osd_cs", "cs", "1", "combination_sweep", "off", "osd_off", "deactivated", -1, 0, 1, "x", None):
        add(f"osd_method_{v!r}", cls="BpOsdDecoder", kwargs={"error_rate": 0.1, "osd_method": v}, ops=[["get", "osd_method"], ["get", "osd_order"]])
    for m_, o in (("osd_0", 0), ("osd_0", 3), ("osd_e", 0), ("osd_e", 5), ("osd_e", 16), ("osd_e", -1), ("osd_cs", 10), ("osd_cs", 70), ("osd_cs", -2), ("osd_e", 2.0), ("osd_cs", "3"),
                  ("off", 4)):
        add(f"osd_order_{m_}_{o}", cls="BpOsdDecoder", kwargs={"error_rate": 0.1, "osd_method": m_, "osd_order": o}, ops=[["get", "osd_method"], ["get", "osd_order"]])
    add("osd_defaults", cls="BpOsdDecoder", kwargs={"error_rate": 0.1},
        ops=[["get", "osd_method"], ["get", "osd_order"], ["get", "input_vector_type"], ["get", "bp_method"], ["get", "max_iter"], ["get", "schedule"],
             ["get", "bp_decoding"], ["get", "osd0_decoding"], ["get", "osdw_decoding"]])
    add("osd_set_after", cls="BpOsdDecoder", kwargs={"error_rate": 0.1, "osd_method": "osd_cs", "osd_order": 4},
        ops=[["set", "osd_method", "osd_0"], ["get", "osd_order"], ["set", "osd_order", 2], ["set", "osd_method", "osd_e"], ["set", "osd_order", 2], ["get", "osd_order"],
             ["set", "osd_order", 20], ["get", "osd_order"], ["set", "osd_method", "nope"], ["get", "osd_method"]])
    add("osd_input_vector_type_kw", cls="BpOsdDecoder", kwargs={"error_rate": 0.1, "input_vector_type": "received_vector"}, ops=[["get", "input_vector_type"]])

    # -- SoftInfoBpDecoder parameters (pyx:712-812) --
    add("soft_defaults", cls="SoftInfoBpDecoder", kwargs={"error_rate": 0.1}, ops=[["get", "bp_method"], ["get", "schedule"], ["get", "max_iter"], ["get", "ms_scaling_factor"]])
    for tag, kw in (("cutoff_sigma", {"cutoff": 5.0, "sigma": 1.0}), ("sigma_zero", {"sigma": 0.0}), ("sigma_neg", {"sigma": -1.0}), ("sigma_str", {"sigma": "1"}),
                    ("cutoff_str", {"cutoff": "x"}), ("sigma_int", {"sigma": 2}), ("bp_method_ps", {"bp_method": "ps"}), ("schedule_parallel", {"schedule": "parallel"}),
                    ("unknown", {"bogus": 1})):
        add(f"soft_{tag}", cls="SoftInfoBpDecoder", kwargs={"error_rate": 0.1, **kw}, ops=[["get", "bp_method"], ["get", "schedule"]])
    add("soft_decode_wrong_length", cls="SoftInfoBpDecoder", kwargs={"error_rate": 0.1}, ops=[["call", "decode", [ND([1.0, 1.0])]]])

    # -- decoding on real inputs (the mirror needs the GPU): known answers and result properties --
    for method in ("ps", "ms"):
        for k, s in enumerate(([0, 0, 0, 1], [0, 1, 0, 1], [1, 0, 1, 0], [1, 1, 1, 1])):
            add(f"decode_rep5_{method}_{k}", pcm="rep5", kwargs={"error_rate": 0.1, "bp_method": method, "max_iter": 5}, gpu=True,
                ops=[["call", "decode", [ND(s, "uint8")]], ["get", "converge"], ["get", "iter"], ["get", "decoding"]])
    add("decode_dtype_preserved", pcm="rep5", kwargs={"error_rate": 0.1, "max_iter": 5}, gpu=True,
        ops=[["call", "decode", [ND([0, 0, 0, 1], "int64")]], ["call", "decode", [ND([0, 0, 0, 1], "float64")]], ["call", "decode", [ND([0, 0, 0, 1], "int8")]]])
    add("decode_received_vector", pcm="rep5", kwargs={"error_rate": 0.1, "max_iter": 5, "input_vector_type": "received_vector"}, gpu=True,
        ops=[["call", "decode", [ND([0, 0, 0, 0, 1], "uint8")]], ["get", "converge"], ["call", "decode", [ND([1, 1, 0, 1, 1], "uint8")]]])
    add("decode_auto_by_length", pcm="rep5", kwargs={"error_rate": 0.1, "max_iter": 5}, gpu=True,
        ops=[["call", "decode", [ND([0, 0, 0, 1], "uint8")]], ["call", "decode", [ND([0, 0, 0, 0, 1], "uint8")]]])
    add("decode_syndrome_gt1", pcm="rep5", kwargs={"error_rate": 0.1, "max_iter": 3}, gpu=True,
        ops=[["call", "decode", [ND([0, 0, 0, 2], "uint8")]], ["get", "converge"], ["get", "iter"]])
    add("decode_zero_prior", pcm="rep5", kwargs={"error_channel": [0.0, 0.1, 0.1, 0.1, 0.1], "max_iter": 5}, gpu=True,
        ops=[["call", "decode", [ND([1, 0, 0, 0], "uint8")]], ["get", "converge"], ["get", "log_prob_ratios"]])
    add("decode_then_zero_keeps_state", pcm="rep5", kwargs={"error_rate": 0.1, "max_iter": 5}, gpu=True,
        ops=[["call", "decode", [ND([0, 1, 0, 1], "uint8")]], ["get", "iter"], ["call", "decode", [ND([0, 0, 0, 0], "uint8")]], ["get", "iter"], ["get", "decoding"],
             ["get", "log_prob_ratios"], ["get", "converge"]])
    add("decode_serial", pcm="rep5", kwargs={"error_rate": 0.1, "max_iter": 5, "schedule": "serial", "serial_schedule_order": [4, 3, 2, 1, 0]}, gpu=True,
        ops=[["call", "decode", [ND([0, 1, 0, 1], "uint8")]], ["get", "converge"], ["get", "iter"]])
    add("decode_osd0", cls="BpOsdDecoder", kwargs={"error_rate": 0.1, "max_iter": 1, "osd_method": "osd_0"}, gpu=True,
        ops=[["call", "decode", [ND([1, 0, 1], "uint8")]], ["get", "converge"], ["get", "osd0_decoding"], ["get", "osdw_decoding"], ["get", "bp_decoding"]])
    add("decode_osd_cs", cls="BpOsdDecoder", kwargs={"error_rate": 0.3, "max_iter": 1, "osd_method": "osd_cs", "osd_order": 3, "bp_method": "ms"}, gpu=True,
        ops=[["call", "decode", [ND([1, 1, 1], "uint8")]], ["get", "converge"], ["get", "osd0_decoding"], ["get", "osdw_decoding"], ["get", "bp_decoding"]])
    add("decode_soft", cls="SoftInfoBpDecoder", pcm="rep5", kwargs={"error_rate": 0.1, "max_iter": 5, "cutoff": 2.0, "sigma": 0.7}, gpu=True,
        ops=[["call", "decode", [ND([2.0, -0.3, 2.0, -2.0])]], ["get", "converge"], ["get", "iter"], ["get", "soft_syndrome"]])
    return P


PROBES = _probes()
