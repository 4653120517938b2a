"""bp_serial_stream_kernel (csrc/bp_serial_stream_kernel.h): the serial schedule (bp.hpp:451-545) streamed through LDS rings.

Every form of the kernel -- ring depth, wavefronts per workgroup, initial messages implicit or written out, a caller's order, orders
that are no permutation, partial tiles, syndrome bytes > 1, continuing lanes compacted out of a first pass -- against the real
reference's fixtures (tests/golden/serial_ldpc36_*.npz), the CPU checker, and the bit-by-bit kernel.  Bar: decisions, flags, iteration
counts and log-ratio BITS."""
import numpy as np
import pytest

from golden_util import load_case

pytestmark = pytest.mark.gpu


def _engine(c):
    from ldpc_amd.engine import HipBpEngine
    return HipBpEngine(c["h"].indptr, c["h"].indices, c["n"], c["channel_probs"], c["max_iter"],
                       0 if c["bp_method"] == "product_sum" else 1, c["ms_scaling_factor"])


def _synd(h, p, seed, shots):
    from ldpc_amd.noise_models import generate_bsc_batch
    n = h.shape[1]
    err = generate_bsc_batch(n, p, seed=seed, shot0=0, shots=shots)
    return np.ascontiguousarray((h.astype(np.int64) @ err.T.astype(np.int64)).T % 2, np.uint8)


BIG = ["serial_ldpc36_n10000_ps50_p050", "serial_ldpc36_n10000_ms50_p050_order", "serial_ldpc36_n10000_ms12_p090_adaptive",
       "serial_ldpc36_n2400_ps40_p078_bytes", "serial_ldpc36_n600_ps20"]


@pytest.mark.parametrize("name", BIG)
def test_streamed_serial_kernel_reproduces_the_reference(name):
    from oracle import bits_equal
    c = load_case(name)
    eng = _engine(c)
    order = c.get("order")
    eng.set_schedule("serial", order if order is not None and len(order) else None)
    eng.set_repack(0)
    for mode, switches in ((2, ()), (2, (("SER_RING", 2), ("SER_WAVES", 8))), (2, (("EXPLICIT_INIT", 1), ("SER_WAVES", 5))), (1, ())):
        eng.set_serial_kernel(mode)
        for k, v in switches:
            eng.set_debug_switch(k, v)
        dec, llr, it, cv = eng.decode_batch(c["syndromes"])
        for k, _ in switches:
            eng.set_debug_switch(k, -1)
        assert np.array_equal(dec, c["decoding"]) and np.array_equal(cv, c["converge"]) and np.array_equal(it, c["iterations"]), (name, mode, switches)
        assert bits_equal(llr[: len(c["llr"])], c["llr"]), (name, mode, switches)
        rowsum = np.sum(np.where(np.abs(llr) < 1e100, llr, 0.0), axis=1)
        assert np.array_equal(rowsum, c["llr_rowsum"]), (name, mode, switches)


@pytest.mark.parametrize("method,alpha,p,max_iter", [("product_sum", 1.0, 0.07, 30), ("minimum_sum", 0.0, 0.06, 25), ("minimum_sum", 0.8, 0.085, 12)])
@pytest.mark.parametrize("ring,waves", [(1, 16), (1, 3), (2, 8), (2, 1)])
def test_streamed_serial_kernel_against_the_checker_and_the_walking_kernel(method, alpha, p, max_iter, ring, waves, oracle_built):
    from ldpc_amd.engine import HipBpEngine
    from ldpc_amd.codes import regular_ldpc_code
    from oracle import bits_equal
    n = 1800
    h = regular_ldpc_code(n, 3, 6, seed=9)
    synd = _synd(h, p, seed=31, shots=333)  # (five full tiles and 13 rows)
    synd[7, 11] = 2   # a byte > 1: never converges
    synd[100] = 0     # an all-zero row: converges in the first iteration
    meth = 0 if method == "product_sum" else 1
    rng = np.random.default_rng(5)
    perm = rng.permutation(n).astype(np.int32)
    holes = perm.copy()
    holes[0:1800:9] = holes[1:1800:9]  # some bits twice, some never: the initial messages must be written out
    for order in (None, perm, holes):
        want = oracle_built.BpOracle(h, error_rate=p, max_iter=max_iter, bp_method=method, ms_scaling_factor=alpha).decode_serial_batch(synd, order)
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), max_iter, meth, alpha)
        eng.set_schedule("serial", order)
        eng.set_repack(0)
        eng.set_serial_kernel(2)
        eng.set_debug_switch("SER_RING", ring)
        eng.set_debug_switch("SER_WAVES", waves)
        dec, llr, it, cv = eng.decode_batch(synd)
        assert np.array_equal(dec, want[0]) and np.array_equal(it, want[2]) and np.array_equal(cv, want[3])
        assert bits_equal(llr, want[1])
        d2, l2, i2, c2 = eng.decode_batch(synd, want_llr=False)
        assert l2 is None and np.array_equal(d2, dec) and np.array_equal(i2, it) and np.array_equal(c2, cv)
        eng.set_serial_kernel(0)
        d0, l0, i0, c0 = eng.decode_batch(synd)
        assert np.array_equal(d0, dec) and np.array_equal(i0, it) and bits_equal(l0, llr)


@pytest.mark.parametrize("lane_max", [-1, 0, 40, 200])
@pytest.mark.parametrize("first_pass", [-1, 1, 2, 4, 7])
def test_streamed_serial_decode_in_passes_gives_identical_results(first_pass, lane_max, oracle_built):
    """Passes that end after 4, 8, 16, ... iterations (or first_pass, 2 first_pass, ...): after each the rows still decoding either carry on
    in their tiles, are compacted into dense tiles, or -- a handful -- finish a workgroup per syndrome (bp_serial_lane_kernel; lane_max 0:
    never, 40: only the last few).  Nothing restarts: same bits as one pass."""
    from ldpc_amd.engine import HipBpEngine
    from ldpc_amd.codes import regular_ldpc_code
    from oracle import bits_equal
    n = 1800
    h = regular_ldpc_code(n, 3, 6, seed=9)
    synd = _synd(h, 0.072, seed=77, shots=1500)
    synd[3, 5] = 3  # never converges
    for meth, alpha in ((0, 1.0), (1, 0.0)):
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, 0.072), 30, meth, alpha)
        eng.set_schedule("serial")
        eng.set_serial_kernel(2)
        eng.set_repack(0)
        d0, l0, i0, c0 = eng.decode_batch(synd)
        assert 0.02 < 1 - c0.mean() < 0.95 and i0[c0].min() < i0[c0].max()
        eng.set_repack(first_pass)
        eng.set_debug_switch("SER_LANE_MAX", lane_max)
        if lane_max == 200:  # a "round" of 4 tiles instead of 256: what a pass leaves beyond whole rounds goes to the lane kernel, the rest on in tiles
            eng.set_debug_switch("SER_ROUND_TILES", 4)
        d1, l1, i1, c1 = eng.decode_batch(synd)
        assert np.array_equal(d0, d1) and np.array_equal(i0, i1) and np.array_equal(c0, c1) and bits_equal(l0, l1)
        d2, l2, i2, c2 = eng.decode_batch(synd, want_llr=False)
        assert l2 is None and np.array_equal(d2, d0) and np.array_equal(i2, i0) and np.array_equal(c2, c0)
    want = oracle_built.BpOracle(h, error_rate=0.072, max_iter=30, bp_method="minimum_sum", ms_scaling_factor=0.0).decode_serial_batch(synd[:200], None)
    assert np.array_equal(d1[:200], want[0]) and np.array_equal(i1[:200], want[2]) and bits_equal(l1[:200], want[1])


@pytest.mark.parametrize("name", BIG)
def test_small_batches_take_the_lane_kernel_from_the_start(name):
    """<= 256 rows: a workgroup per syndrome from the first iteration (what a `for shot: decode(shot)` caller of the big code gets)."""
    from oracle import bits_equal
    c = load_case(name)
    eng = _engine(c)
    order = c.get("order")
    eng.set_schedule("serial", order if order is not None and len(order) else None)
    eng.set_serial_kernel(2)
    dec, llr, it, cv = eng.decode_batch(c["syndromes"])
    assert np.array_equal(dec, c["decoding"]) and np.array_equal(cv, c["converge"]) and np.array_equal(it, c["iterations"])
    assert bits_equal(llr[: len(c["llr"])], c["llr"])
    d1, l1, i1, c1 = eng.decode_batch(c["syndromes"][:1])
    assert np.array_equal(d1[0], c["decoding"][0]) and int(i1[0]) == int(c["iterations"][0]) and bits_equal(l1[0], c["llr"][0])


def test_lane_kernel_with_orders_that_skip_bits(oracle_built):
    from ldpc_amd.engine import HipBpEngine
    from ldpc_amd.codes import regular_ldpc_code
    from oracle import bits_equal
    n = 1800
    h = regular_ldpc_code(n, 3, 6, seed=9)
    synd = _synd(h, 0.07, seed=3, shots=90)
    rng = np.random.default_rng(8)
    holes = rng.permutation(n).astype(np.int32)
    holes[0:1792:7] = holes[1:1792:7]  # some bits twice, some never
    for method, alpha in (("product_sum", 1.0), ("minimum_sum", 0.7)):
        want = oracle_built.BpOracle(h, error_rate=0.07, max_iter=20, bp_method=method, ms_scaling_factor=alpha).decode_serial_batch(synd, holes)
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, 0.07), 20, 0 if method == "product_sum" else 1, alpha)
        eng.set_schedule("serial", holes)
        eng.set_serial_kernel(2)
        dec, llr, it, cv = eng.decode_batch(synd)
        assert np.array_equal(dec, want[0]) and np.array_equal(it, want[2]) and np.array_equal(cv, want[3]) and bits_equal(llr, want[1])


def test_batches_beyond_device_memory_are_decoded_in_resident_pieces(oracle_built):
    """A batch whose message state is not resident at once (forced here: at most 70 tiles at a time) is cut into pieces that are, each
    decoded in passes by itself; below 64 tiles the chunk loop of the one-pass path takes over.  Same bits either way."""
    from ldpc_amd.engine import HipBpEngine
    from ldpc_amd.codes import regular_ldpc_code
    from oracle import bits_equal
    n = 1800
    h = regular_ldpc_code(n, 3, 6, seed=9)
    synd = _synd(h, 0.072, seed=12, shots=10000)
    synd[77, 1] = 2
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, 0.072), 25, 0, 1.0)
    eng.set_schedule("serial")
    eng.set_serial_kernel(2)
    d0, l0, i0, c0 = eng.decode_batch(synd)
    for cap in (70, 9):
        eng.set_tuning(max_chunk_tiles=cap)
        d1, l1, i1, c1 = eng.decode_batch(synd)
        assert np.array_equal(d0, d1) and np.array_equal(i0, i1) and np.array_equal(c0, c1) and bits_equal(l0, l1), cap
    want = oracle_built.BpOracle(h, error_rate=0.072, max_iter=25, bp_method="product_sum").decode_serial_batch(synd[:150], None)
    assert np.array_equal(d0[:150], want[0]) and np.array_equal(i0[:150], want[2]) and bits_equal(l0[:150], want[1])
