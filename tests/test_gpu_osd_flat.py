"""osd0_flat_kernel (OSD-0 on small matrices with the columns permuted into their sorted order, round 6): every instantiation <R, D> --
R = 1 / 2 register rows (m <= 64 / 128), D = 2 / 3 / 5 / 8 dwords of columns (n <= 64 / 96 / 160 / 256) -- against the CPU checker and
against osd0_reg_kernel (LDPC_HIP_OSD_NO_FLAT), decisions and status bytes; rows outside the image of a rank-deficient H included
(the second pass runs through the same kernel)."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

# (n, dv, dc) -> m = n dv / dc; the kernel the host picks: <R, D>
SHAPES = [
    (60, 2, 6, (1, 2)),     # m = 20
    (96, 2, 4, (1, 3)),     # m = 48
    (128, 2, 8, (1, 5)),    # m = 32, n in (96, 160]
    (200, 2, 8, (1, 8)),    # m = 50, n in (160, 256]
    (64, 3, 2, (2, 2)),     # m = 96: more checks than bits
    (96, 3, 4, (2, 3)),     # m = 72
    (144, 3, 6, (2, 5)),    # m = 72 (the BB [[144,12,12]] shape)
    (200, 4, 8, (2, 8)),    # m = 100
]


@pytest.mark.parametrize("n,dv,dc,kern", SHAPES)
@pytest.mark.parametrize("method", ["product_sum", "minimum_sum"])
def test_flat_osd0_every_instantiation(oracle_built, n, dv, dc, kern, method):
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    h = sp.csr_matrix(codes.regular_ldpc_code(n, dv, dc, seed=11 + n + dv))
    m = h.shape[0]
    assert (1 if m <= 64 else 2, next(d for d in (2, 3, 5, 8) if 32 * d >= n)) == kern
    rng = np.random.default_rng(1000 * n + dv)
    p, B, max_iter = 0.08, 300, 6  # few iterations at a high error rate: most rows go through OSD
    e = (rng.random((B, n)) < p).astype(np.uint8)
    s = np.asarray((h @ e.T % 2).T, dtype=np.uint8)
    s[::7, int(rng.integers(0, m))] ^= 1  # outside the image wherever H is rank-deficient: status 2, the second pass
    probs = np.full(n, p)
    probs[::5] = 0.03  # non-uniform priors: fewer exact ties, another order
    orc = oracle_built.BpOracle(h, error_channel=probs, max_iter=max_iter, bp_method=method, ms_scaling_factor=0.8)
    want = orc.bposd0_decode_batch(s)
    eng = HipBpEngine(h.indptr, h.indices, n, probs, max_iter, 0 if method == "product_sum" else 1, 0.8)
    got = eng.decode_batch(s, osd0=True)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3])
    status = eng.osd_status(B)
    solves = np.all((got[0].astype(np.int64) @ h.T.toarray().astype(np.int64)) % 2 == s, axis=1)
    assert np.array_equal(status, np.where(got[3] != 0, 0, np.where(solves, 1, 2)))
    assert (status > 0).sum() >= 20, "the case is meant to send rows through OSD"
    eng.set_debug_switch("OSD_NO_FLAT", 1)
    reg = eng.decode_batch(s, osd0=True)
    assert np.array_equal(reg[0], got[0]) and np.array_equal(eng.osd_status(B), status)
    eng.close()
