"""The reference's own solver fixture (cpp_test/test_inputs/gf2_lu_solve_test.csv, 400 consistent GF(2) systems; repacked by
tests/golden/make_golden_lusolve.py).  The reference asserts A x == y for fast_solve / lu_solve on each of them
(cpp_test/TestGF2RowReduce.cpp:327-368, 456-497); OSD-0 is fast_solve over a column order, so the same property must
hold for the oracle's OSD-0, for the real reference behind oracle/_ref, and for the device kernels (both the
register-resident and the LDS-resident elimination: the systems range from 1 x 1 to 497 x 487)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from golden_util import GOLDEN_DIR


def systems():
    z = np.load(os.path.join(GOLDEN_DIR, "lusolve_reference_systems.npz"))
    rhs = np.unpackbits(z["rhs"], count=int(z["rhs_bits"]))
    rp_all, ci_all = z["row_ptr"], z["col_idx"]
    row0 = bit0 = 0
    for m, n in zip(z["m"], z["n"]):
        m, n = int(m), int(n)
        rp = (rp_all[row0:row0 + m + 1] - rp_all[row0]).astype(np.int32)
        ci = ci_all[rp_all[row0]:rp_all[row0 + m]].astype(np.int32)
        h = sp.csr_matrix((np.ones(len(ci), np.uint8), ci, rp), shape=(m, n), dtype=np.uint8)
        yield h, rhs[bit0:bit0 + m].astype(np.uint8)
        row0 += m
        bit0 += m


def _solves(h, x, y):
    return np.array_equal(np.asarray(h @ x % 2, dtype=np.uint8).ravel(), y)


def test_fixture_shape():
    all_systems = list(systems())
    assert len(all_systems) == 400
    assert max(h.shape[0] for h, _ in all_systems) == 497 and max(h.shape[1] for h, _ in all_systems) == 487


def test_oracle_osd0_solves_every_reference_system(oracle_built):
    have_ref = oracle.have_ref()
    checked_ref = 0
    for k, (h, y) in enumerate(systems()):
        m, n = h.shape
        o = oracle_built.BpOracle(h, error_rate=0.1, max_iter=1)
        llr = np.zeros(n)  # all equal: the stable sort keeps the natural column order, i.e. plain fast_solve(y)
        x = o.osd0(y, llr)
        assert _solves(h, x, y), f"system {k} ({m} x {n})"
        if have_ref and k % 4 == 0:
            r = oracle.RefBpOsd(h, error_rate=0.1, max_iter=1)
            assert np.array_equal(x, r.osd0(y, llr)), f"system {k}: differs from the reference's fast_solve"
            checked_ref += 1
    assert not have_ref or checked_ref == 100


@pytest.mark.gpu
@pytest.mark.parametrize("osd_kernel", [-1, 0])
def test_device_bposd_solves_every_reference_system(osd_kernel, oracle_built):
    from ldpc_amd.engine import HipBpEngine
    bp_conv = 0
    for k, (h, y) in enumerate(systems()):
        m, n = h.shape
        if osd_kernel == 0 and k % 3:  # the LDS variant on a third of them (the large ones run there in either mode)
            continue
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, 0.1), 2, 1, 0.75)
        eng.set_osd_kernel(osd_kernel)
        dec, llr, it, cv = eng.decode_batch(y[None, :], osd0=True)
        assert _solves(h, dec[0], y), f"system {k} ({m} x {n})"
        bp_conv += int(cv[0])
        if k % 16 == 0:  # and bit for bit what the oracle's BP + OSD-0 gives
            want = oracle_built.BpOracle(h, error_rate=0.1, max_iter=2, bp_method="minimum_sum", ms_scaling_factor=0.75).bposd0_decode_batch(y[None, :])
            assert np.array_equal(dec, want[0]) and np.array_equal(cv, want[3])
        eng.close()
    assert bp_conv < 400  # OSD really ran
