"""The variable-degree LDS ring of the streamed kernel (csrc/bp_stream_kernel.h: bp_decode_kernel<., ., 16, 8, LDPC_RING_VAR>; items and their
order: host_stream.h ensure_var_ring_items): irregular matrices -- rows of 0 ... 16 entries, columns of 0 ... 8, an odd number of columns --
against the register variant (VAR_RING 0) on every row, bit for bit, and against the CPU checker (bp.hpp:192-325); regular matrices
against their fixed-degree ring.  Queue sizes from the smallest that holds one item (8 units) up, every workgroup size, hand-off
thresholds, the two-pass decode, both methods and both arithmetic modes."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _decode(eng, s, **kw):
    got = eng.decode_batch(s, **kw)
    return [x.cpu().numpy() if x is not None and hasattr(x, "cpu") else x for x in got]


def _same(a, b, tag):
    from golden_util import bits_equal
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]), tag
    if a[1] is not None and b[1] is not None:
        assert bits_equal(a[1], b[1]), "llr " + tag


@pytest.mark.parametrize("method,alpha,math", [("product_sum", 1.0, "exact"), ("minimum_sum", 0.0, "exact"), ("minimum_sum", 0.75, "exact"), ("product_sum", 1.0, "fast")])
def test_irregular_code_ring_equals_register_variant_and_checker(method, alpha, math, oracle_built):
    from golden_util import bits_equal
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    h = codes.irregular_ldpc_code(600, 300, seed=3)
    n, p, max_iter = 600, 0.03, 16
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), max_iter, 0 if method == "product_sum" else 1, alpha)
    if math == "fast":
        eng.set_math("fast")
    eng.set_small_code_kernel(0)
    s = eng.gen_bsc_syndromes(13, p, shot0=0, shots=40000, device="cuda:0")
    eng.set_debug_switch("VAR_RING", 0)
    ref = _decode(eng, s, want_llr=True)
    assert 0.2 < ref[3].mean() < 0.999
    if math == "exact":
        rows = np.r_[0:150, 39850:40000]
        want = oracle_built.BpOracle(h, error_rate=p, max_iter=max_iter, bp_method=method, ms_scaling_factor=alpha).decode_batch(s.cpu().numpy()[rows])
        assert np.array_equal(ref[0][rows], want[0]) and np.array_equal(ref[2][rows], want[2]) and bits_equal(ref[1][rows], want[1])
    eng.set_debug_switch("VAR_RING", 1)
    for units, waves, handoff, repack in ((-1, 0, -1, -1), (8, 12, 0, 0), (9, 5, 64, 0), (16, 7, -1, 3), (40, 3, -1, 0), (11, 1, 4, 0), (-1, 16, -1, 0)):
        eng.set_debug_switch("VAR_RING_UNITS", units)
        eng.set_tuning(waves_per_workgroup=waves)
        eng.set_handoff(handoff)
        eng.set_repack(repack)
        for want_llr in (True, False):
            got = _decode(eng, s, want_llr=want_llr)
            _same(got, ref, f"units {units} waves {waves} handoff {handoff} repack {repack} llr {want_llr}")
    eng.set_debug_switch("VAR_RING_UNITS", -1)
    eng.set_tuning(waves_per_workgroup=0)
    eng.set_handoff(-1)
    eng.set_repack(-1)
    few = _decode(eng, s[:700].contiguous(), want_llr=True)  # 11 tiles: per-pass kernels only
    _same(few, [x[:700] for x in ref], "700 rows")
    eng.close()


def test_ragged_matrix_with_empty_rows_isolated_bits_and_an_odd_number_of_columns(oracle_built):
    from golden_util import bits_equal
    from ldpc_amd.engine import HipBpEngine
    rng = np.random.default_rng(11)
    m, n = 210, 421
    rows, cols = [], []
    for i in range(m):
        d = int(rng.choice([0, 1, 2, 3, 5, 9, 12, 16], p=[0.03, 0.05, 0.1, 0.3, 0.2, 0.15, 0.1, 0.07]))
        for j in rng.choice(n, size=d, replace=False):
            rows.append(i)
            cols.append(int(j))
    h = sp.csr_matrix((np.ones(len(rows), np.uint8), (rows, cols)), shape=(m, n))
    h.sum_duplicates()
    h.data[:] = 1
    cdeg = np.asarray(h.sum(axis=0)).ravel()
    while cdeg.max() > 8:  # thin the heaviest columns to at most 8 entries
        j = int(np.argmax(cdeg))
        i = h[:, j].nonzero()[0][0]
        h = h.tolil(); h[i, j] = 0; h = h.tocsr(); h.eliminate_zeros()
        cdeg = np.asarray(h.sum(axis=0)).ravel()
    rdeg = np.diff(h.indptr)
    assert rdeg.min() == 0 and rdeg.max() > 8 and cdeg.min() == 0 and cdeg.max() <= 8 and n % 2 == 1
    p = 0.02
    from ldpc_amd.noise_models import generate_bsc_batch
    err = generate_bsc_batch(n, p, seed=5, shot0=0, shots=1500)
    synd = np.ascontiguousarray((h.astype(np.int64) @ err.T.astype(np.int64)).T % 2, np.uint8)
    for method, alpha in (("product_sum", 1.0), ("minimum_sum", 0.625)):
        want = oracle_built.BpOracle(h, error_rate=p, max_iter=12, bp_method=method, ms_scaling_factor=alpha).decode_batch(synd[:200])
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), 12, 0 if method == "product_sum" else 1, alpha)
        eng.set_small_code_kernel(0)
        eng.set_handoff(0)  # the persistent kernel all the way
        eng.set_debug_switch("VAR_RING", 0)
        ref = eng.decode_batch(synd)
        for units in (-1, 8, 13):
            eng.set_debug_switch("VAR_RING", 1)
            eng.set_debug_switch("VAR_RING_UNITS", units)
            got = eng.decode_batch(synd)
            _same(got, ref, f"{method} units {units}")
            assert np.array_equal(got[0][:200], want[0]) and np.array_equal(got[2][:200], want[2]) and bits_equal(got[1][:200], want[1])
        eng.close()


@pytest.mark.parametrize("dv,dc", [(3, 6), (4, 8)])
def test_regular_codes_forced_onto_the_variable_ring_equal_the_fixed_ring(dv, dc):
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    n, p = 1200, 0.06 if dv == 3 else 0.05
    h = codes.regular_ldpc_code(n, dv, dc, seed=4)
    for meth, alpha in ((0, 1.0), (1, 0.0)):
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), 20, meth, alpha)
        eng.set_small_code_kernel(0)
        s = eng.gen_bsc_syndromes(3, p, shot0=0, shots=20000, device="cuda:0")
        ref = _decode(eng, s, want_llr=True)
        assert 0.05 < ref[3].mean() < 0.999
        eng.set_debug_switch("VAR_RING", 1)
        for handoff in (-1, 0):
            eng.set_handoff(handoff)
            _same(_decode(eng, s, want_llr=True), ref, f"({dv},{dc}) method {meth} handoff {handoff}")
        eng.close()


IRREGULAR_FIXTURES = ["irregular_ldpc_n600_ps16_p030", "irregular_ldpc_n600_ms16_p030_adaptive", "irregular_ldpc_n2400_ps30_p075",
                      "irregular_ldpc_n2400_ms30_p060_a0625", "irregular8_ldpc_n1200_ps20_p040"]


@pytest.mark.parametrize("name", IRREGULAR_FIXTURES)
def test_real_reference_fixtures_on_every_streamed_path(name):
    """Irregular-code fixtures captured from the real reference (tests/golden/make_golden_irregular.py), forced off the on-chip kernels: the
    default streamed path (product-sum: per-pass kernels from the first iteration; min-sum: the persistent register variant with its
    hand-off), the persistent kernel alone, the variable-degree ring with two queue sizes, per-pass kernels with 1 and 16 rows per
    wavefront -- decisions, flags, iteration counts, log-ratio BITS and the row sums of every log-ratio vector."""
    from golden_util import bits_equal, load_case, rowsum
    from ldpc_amd.engine import HipBpEngine
    c = load_case(name)
    eng = HipBpEngine(c["h"].indptr, c["h"].indices, c["n"], c["channel_probs"], c["max_iter"], 0 if c["bp_method"] == "product_sum" else 1, c["ms_scaling_factor"])
    eng.set_small_code_kernel(0)
    k = len(c["llr"])
    for tag, handoff, switches in (("default", -1, ()), ("persistent", 0, ()), ("ring", 0, (("VAR_RING", 1),)), ("ring 8 units + hand-off", 1, (("VAR_RING", 1), ("VAR_RING_UNITS", 8))),
                                   ("per-pass, 1 row a wavefront", 100000, (("SPREAD_NODES", 1),)), ("per-pass, 16 rows a wavefront", 100000, (("SPREAD_NODES", 16),))):
        eng.set_handoff(handoff)
        for key, val in switches:
            eng.set_debug_switch(key, val)
        dec, llr, it, cv = eng.decode_batch(c["syndromes"])
        for key, _ in switches:
            eng.set_debug_switch(key, -1)
        assert np.array_equal(dec, c["decoding"]) and np.array_equal(cv, c["converge"]) and np.array_equal(it, c["iterations"]), (name, tag)
        assert bits_equal(llr[:k], c["llr"]), (name, tag)
        assert np.array_equal(rowsum(llr), c["llr_rowsum"]), (name, tag)
    eng.close()


@pytest.mark.parametrize("dv,dc,p", [(3, 4, 0.15), (3, 5, 0.11), (5, 10, 0.055), (3, 6, 0.07), (4, 8, 0.06)])
def test_fixed_degree_ring_variants_of_every_regular_shape(dv, dc, p, oracle_built):
    """The LDS-DMA ring instantiations (csrc/tu_stream.hip: pick_kernel) -- (6,3), (8,4) and, since round 6, (4,3), (5,3)-shaped matrices (rows x columns;
    a (10,5)-shaped one takes the register variant / per-pass kernels) -- against the register variant of the same persistent kernel, the per-pass kernels and the CPU checker: every row bit for
    bit, both methods, every hand-off, the two-pass decode, with and without the table of initial values."""
    from golden_util import bits_equal
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    n = 1200 if dc != 10 else 1500
    h = codes.regular_ldpc_code(n, dv, dc, seed=6)
    for meth, alpha, name in ((0, 1.0, "product_sum"), (1, 0.0, "minimum_sum"), (1, 0.8, "minimum_sum")):
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), 24, meth, alpha)
        eng.set_small_code_kernel(0)
        s = eng.gen_bsc_syndromes(19, p, shot0=0, shots=21000, device="cuda:0")
        eng.set_ring(0)
        eng.set_handoff(0)   # the persistent kernel's register variant for every tile
        eng.set_repack(0)
        ref = _decode(eng, s, want_llr=True)
        assert 0.01 < ref[3].mean() < 0.9999, (dv, dc, ref[3].mean())
        rows = np.r_[0:60, 20940:21000]
        want = oracle_built.BpOracle(h, error_rate=p, max_iter=24, bp_method=name, ms_scaling_factor=alpha).decode_batch(s.cpu().numpy()[rows])
        assert np.array_equal(ref[0][rows], want[0]) and np.array_equal(ref[2][rows], want[2]) and bits_equal(ref[1][rows], want[1])
        for ring, handoff, repack, switches in ((1, -1, -1, ()), (1, 0, 0, ()), (2, 64, 0, ()), (3, -1, 3, ()), (1, 0, 0, (("EXPLICIT_INIT", 1),)), (0, -1, 0, ())):
            eng.set_ring(ring)
            eng.set_handoff(handoff)
            eng.set_repack(repack)
            for k, v in switches:
                eng.set_debug_switch(k, v)
            for want_llr in (True, False):
                _same(_decode(eng, s, want_llr=want_llr), ref, f"({dv},{dc}) method {meth} alpha {alpha} ring {ring} handoff {handoff} repack {repack} {switches} llr {want_llr}")
            for k, _ in switches:
                eng.set_debug_switch(k, -1)
        eng.close()
