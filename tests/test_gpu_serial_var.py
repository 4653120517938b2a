"""bp_serial_stream_var_kernel / bp_serial_lane_var_kernel (csrc/bp_serial_var_kernel.h): the streamed serial schedule (bp.hpp:451-545)
for any degree profile -- (4,8)-regular and irregular codes (rows of 3 .. 16, columns of 2 .. 8), and the (6,3) code forced onto the item
form -- against the REAL reference's fixtures (tests/golden/serial_ldpc48_*, serial_irregular_*: make_golden_serial_big.py --shapes), the
CPU checker, and the bit-by-bit kernel.  Every queue size and workgroup size, a caller's order, orders that are no permutation, syndrome
bytes > 1, the decode in passes with compaction and with the per-syndrome kernel.  Bar: decisions, flags, iteration counts, log-ratio BITS."""
import numpy as np
import pytest

from golden_util import load_case

pytestmark = pytest.mark.gpu

SHAPES = ["serial_ldpc48_n2400_ps30_p055", "serial_ldpc48_n2400_ms24_p050_adaptive_bytes", "serial_irregular_n2400_ps30_p045",
          "serial_irregular_n2400_ms30_p040_order", "serial_irregular_n2400_ps12_p040_repeats", "serial_irregular_n10000_ps50_p050",
          "serial_ldpc48_n10000_ps50_p050"]
SIX_THREE = ["serial_ldpc36_n2400_ps40_p078_bytes", "serial_ldpc36_n10000_ms50_p050_order"]


def _engine(c):
    from ldpc_amd.engine import HipBpEngine
    return HipBpEngine(c["h"].indptr, c["h"].indices, c["n"], c["channel_probs"], c["max_iter"],
                       0 if c["bp_method"] == "product_sum" else 1, c["ms_scaling_factor"])


def _check(c, got, tag):
    from oracle import bits_equal
    dec, llr, it, cv = got
    assert np.array_equal(dec, c["decoding"]) and np.array_equal(cv, c["converge"]) and np.array_equal(it, c["iterations"]), tag
    assert bits_equal(llr[: len(c["llr"])], c["llr"]), tag
    assert np.array_equal(np.sum(np.where(np.abs(llr) < 1e100, llr, 0.0), axis=1), c["llr_rowsum"]), tag


@pytest.mark.parametrize("name", SHAPES + SIX_THREE)
def test_item_form_reproduces_the_reference(name):
    c = load_case(name)
    eng = _engine(c)
    order = c.get("order")
    eng.set_schedule("serial", order if order is not None and len(order) else None)
    eng.set_serial_kernel(2)        # the streamed kernels whatever the width of the levels
    eng.set_debug_switch("SER_VAR", 1)  # ... and the item form on (6,3) matrices too
    eng.set_debug_switch("SER_LANE_MAX", 0)  # tiles only
    for repack, switches in ((0, ()), (0, (("SER_VAR_UNITS", 16), ("SER_WAVES", 5))), (0, (("SER_VAR_UNITS", 9), ("SER_WAVES", 16))), (0, (("SER_WAVES", 1),)), (2, ()), (-1, (("SER_WAVES", 7),))):
        eng.set_repack(repack)
        for k, v in switches:
            eng.set_debug_switch(k, v)
        got = eng.decode_batch(c["syndromes"])
        for k, _ in switches:
            eng.set_debug_switch(k, -1)
        _check(c, got, (name, repack, switches))
    # a workgroup per syndrome from the start (<= 256 rows), and for what a pass leaves
    eng.set_debug_switch("SER_LANE_MAX", -1)
    eng.set_repack(-1)
    _check(c, eng.decode_batch(c["syndromes"]), (name, "lanes"))
    eng.set_debug_switch("SER_LANE_THREADS", 192)
    _check(c, eng.decode_batch(c["syndromes"]), (name, "lanes, 3 wavefronts"))
    one = eng.decode_batch(c["syndromes"][:1])
    assert np.array_equal(one[0][0], c["decoding"][0]) and int(one[2][0]) == int(c["iterations"][0])
    eng.close()


@pytest.mark.parametrize("method,alpha,p,max_iter", [("product_sum", 1.0, 0.04, 30), ("minimum_sum", 0.0, 0.035, 25)])
@pytest.mark.parametrize("code", ["irregular", "ldpc48", "ldpc510", "ragged"])
def test_item_form_against_the_checker_in_passes(code, method, alpha, p, max_iter, oracle_built):
    """1 500 rows: passes, compaction into dense tiles, the per-syndrome kernel for stragglers, the remainder rule -- identical results,
    and equal to the CPU checker and to the bit-by-bit kernel."""
    import scipy.sparse as sp
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    from ldpc_amd.noise_models import generate_bsc_batch
    from oracle import bits_equal
    if code == "irregular":
        h = codes.irregular_ldpc_code(1800, 900, seed=11)
    elif code == "ldpc48":
        h = codes.regular_ldpc_code(1600, 4, 8, seed=4)
    elif code == "ldpc510":
        h = codes.regular_ldpc_code(1500, 5, 10, seed=6)   # rows of 10, columns of 5: the <16, 8> instantiation on a regular code
        p *= 0.6
    else:  # rows of 1 .. 12 entries, columns of 1 .. 7, an odd number of columns
        rng = np.random.default_rng(8)
        dense = (rng.random((700, 1501)) < 0.004).astype(np.uint8)
        for j in np.flatnonzero(dense.sum(axis=0) == 0):
            dense[rng.integers(700), j] = 1
        for j in np.flatnonzero(dense.sum(axis=0) > 7):
            dense[np.flatnonzero(dense[:, j])[7:], j] = 0
        for i in np.flatnonzero(dense.sum(axis=1) > 12):
            extra = np.flatnonzero(dense[i])[12:]
            extra = [j for j in extra if dense[:, j].sum() > 1]
            dense[i, extra] = 0
        dense = dense[dense.sum(axis=1) > 0]
        assert dense.sum(axis=1).max() <= 16 and dense.sum(axis=0).min() >= 1 and dense.sum(axis=0).max() <= 8
        h = sp.csr_matrix(dense)
        p *= 0.5
    m, n = h.shape
    err = generate_bsc_batch(n, p, seed=5, shot0=0, shots=1500)
    synd = np.ascontiguousarray((h.astype(np.int64) @ err.T.astype(np.int64)).T % 2, np.uint8)
    synd[3, 5] = 3    # never converges
    synd[100] = 0     # converges in the first iteration
    meth = 0 if method == "product_sum" else 1
    rng = np.random.default_rng(5)
    perm = rng.permutation(n).astype(np.int32)
    holes = perm.copy()
    holes[0:n - 1:9] = holes[1:n:9][: len(holes[0:n - 1:9])]
    for order in (None, perm, holes):
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), max_iter, meth, alpha)
        eng.set_schedule("serial", order)
        eng.set_serial_kernel(0)   # one wavefront walks the bits
        d0, l0, i0, c0 = eng.decode_batch(synd)
        want = oracle_built.BpOracle(h, error_rate=p, max_iter=max_iter, bp_method=method, ms_scaling_factor=alpha).decode_serial_batch(synd[:120], order)
        assert np.array_equal(d0[:120], want[0]) and np.array_equal(i0[:120], want[2]) and np.array_equal(c0[:120], want[3]) and bits_equal(l0[:120], want[1])
        eng.set_serial_kernel(2)
        for repack, lane_max, round_tiles in ((0, -1, -1), (-1, -1, -1), (1, 0, -1), (2, 40, -1), (3, 200, 4), (7, -1, -1)):
            eng.set_repack(repack)
            eng.set_debug_switch("SER_LANE_MAX", lane_max)
            eng.set_debug_switch("SER_ROUND_TILES", round_tiles)
            for want_llr in (True, False):
                d1, l1, i1, c1 = eng.decode_batch(synd, want_llr=want_llr)
                tag = (code, method, "order" if order is not None else None, repack, lane_max, want_llr)
                assert np.array_equal(d0, d1) and np.array_equal(i0, i1) and np.array_equal(c0, c1), tag
                assert (l1 is None) if not want_llr else bits_equal(l0, l1), tag
        eng.close()


def test_bits_without_a_check_keep_the_level_kernel(oracle_built):
    """A column of weight 0 has no item: such matrices stay with bp_serial_level_kernel (host_serial.h: plan_serial_stream)."""
    import scipy.sparse as sp
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    from oracle import bits_equal
    h = sp.lil_matrix(codes.irregular_ldpc_code(600, 300, seed=2))
    h[:, 17] = 0
    h = sp.csr_matrix(h)
    h.eliminate_zeros()
    synd = (np.random.default_rng(1).random((130, 300)) < 0.1).astype(np.uint8)
    eng = HipBpEngine(h.indptr, h.indices, 600, np.full(600, 0.03), 10, 0, 1.0)
    eng.set_schedule("serial")
    eng.set_serial_kernel(2)
    got = eng.decode_batch(synd)
    want = oracle_built.BpOracle(h, error_rate=0.03, max_iter=10, bp_method="product_sum").decode_serial_batch(synd, None)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]) and bits_equal(got[1], want[1])
    eng.close()
