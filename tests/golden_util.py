"""Loader for tests/golden/*.npz (reference outputs captured by tests/golden/make_golden.py)."""
from __future__ import annotations

import glob
import os
import zlib

import numpy as np
import scipy.sparse as sp

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names(prefix: str = ""):
    """BP fixtures (``decoding`` = BpDecoder output).  BP+OSD-0 fixtures are listed by ``osd_case_names``."""
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))
    return [n for n in names if not n.startswith(("osd_", "osdw_", "serial_", "mcs_", "soft_", "lusolve_", "qcodes_", "window_", "outimage_", "stateful_", "sinter_"))]


def serial_case_names():
    """Fixtures captured with schedule = SERIAL (bp.hpp:451-545); ``order`` = serial_schedule_order or None."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "serial_*.npz")))


def osd_case_names():
    """Fixtures whose ``decoding`` is BpOsdDecoder's (OSD_0) output; converge/iterations are BP's."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "osd_*.npz")))


def osdw_case_names():
    """Fixtures of BpOsdDecoder with osd_method OSD_E / OSD_CS and osd_order > 0 (``osd_method``, ``osd_order``,
    ``osd0_decoding`` stored besides the usual fields; ``decoding`` is the swept solution)."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "osdw_*.npz")))


def _h_from_recipe(recipe: str):
    from ldpc_amd import codes
    allowed = {"regular_ldpc_code": codes.regular_ldpc_code, "irregular_ldpc_code": codes.irregular_ldpc_code}
    fn, args = recipe.split("(", 1)
    args = args.rstrip(")")
    pos, kw = [], {}
    for a in args.split(","):
        if "=" in a:
            k, v = a.split("=")
            kw[k.strip()] = int(v)
        else:
            pos.append(int(a))
    return allowed[fn](*pos, **kw)


_H_CACHE: dict = {}


def load_case(name: str) -> dict:
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    m, n = int(z["m"]), int(z["n"])
    recipe = str(z["recipe"])
    if recipe:
        if recipe not in _H_CACHE:
            _H_CACHE[recipe] = _h_from_recipe(recipe)
        h = _H_CACHE[recipe]
    else:
        rp, ci = z["row_ptr"], z["col_idx"]
        h = sp.csr_matrix((np.ones(len(ci), np.uint8), ci, rp), shape=(m, n), dtype=np.uint8)
    h = sp.csr_matrix(h)
    h.sort_indices()
    rp = np.ascontiguousarray(h.indptr, np.int32)
    ci = np.ascontiguousarray(h.indices, np.int32)
    crc = zlib.crc32(ci.tobytes(), zlib.crc32(rp.tobytes(), zlib.crc32(np.array([m, n], np.int64).tobytes())))
    assert np.uint32(crc) == z["h_crc"], f"{name}: parity-check matrix does not match the fixture's checksum"
    synd = z["syndromes"]
    if bool(z["syndromes_packed"]):
        synd = np.unpackbits(synd, axis=1, count=m)
    dec = np.unpackbits(z["decoding"], axis=1, count=n)
    return dict(
        name=name, h=h, m=m, n=n, channel_probs=z["channel_probs"], max_iter=int(z["max_iter"]),
        bp_method=("product_sum", "minimum_sum")[int(z["bp_method"])],
        ms_scaling_factor=float(z["ms_scaling_factor"]), syndromes=np.ascontiguousarray(synd, np.uint8),
        decoding=dec, converge=z["converge"].astype(bool), iterations=z["iterations"].astype(np.int32),
        llr=z["llr"], llr_rowsum=z["llr_rowsum"], note=str(z["note"]),
        order=(z["order"].astype(np.int32) if "order" in z.files and z["order"].size else None),
        osd_method=int(z["osd_method"]) if "osd_method" in z.files else 1,
        osd_order=int(z["osd_order"]) if "osd_order" in z.files else 0,
        osd0_decoding=np.unpackbits(z["osd0_decoding"], axis=1, count=n) if "osd0_decoding" in z.files else None,
    )


from oracle import bits_equal, llr_close  # noqa: E402,F401  (single definition of the comparison rules)


def rowsum(llr: np.ndarray) -> np.ndarray:
    return np.sum(np.where(np.abs(llr) < 1e100, llr, 0.0), axis=1)
