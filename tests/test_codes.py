"""The matrix generators the benches, tests and fixture generators share: shapes, weights, and the algebra they promise."""
import numpy as np
import scipy.sparse as sp

from ldpc_amd import codes
from golden_util import load_case


def test_baseline_matrices():
    h = codes.regular_ldpc_code(10000, 3, 6, seed=1)
    assert h.shape == (5000, 10000) and h.nnz == 30000
    assert set(np.diff(h.indptr)) == {6} and set(np.diff(h.tocsc().indptr)) == {3}
    s = codes.rotated_surface_code_x(21)
    assert s.shape == (220, 441) and s.nnz == 840
    b = codes.bivariate_bicycle_hx()
    assert b.shape == (72, 144) and set(np.diff(b.indptr)) == {6} and set(np.diff(b.tocsc().indptr)) == {3}
    assert codes.hamming_code(5).shape == (5, 31)


def test_hypergraph_product():
    h1 = codes.regular_ldpc_code(n=32, dv=3, dc=4, seed=5)
    m1, n1 = h1.shape
    hx = codes.hypergraph_product_hx(h1)
    assert hx.shape == (m1 * n1, n1 * n1 + m1 * m1) == (768, 1600)
    # the partner Z-check matrix [I (x) H1 | H1^T (x) I] commutes with it: a CSS code
    hz = sp.hstack([sp.kron(sp.identity(n1, dtype=np.uint8), h1), sp.kron(h1.T, sp.identity(m1, dtype=np.uint8))]).tocsr()
    assert not ((hx.astype(np.int64) @ hz.T.astype(np.int64)).toarray() % 2).any()
    assert (sp.csr_matrix(load_case("hgp1600_ms20_p030")["h"]) != hx).nnz == 0, "the committed fixtures were made on this matrix"
    h2 = codes.hamming_code(3)
    assert codes.hypergraph_product_hx(h1, h2).shape == (m1 * 7, n1 * 7 + m1 * 3)


def test_irregular_ldpc_code_profile():
    """The irregular generator (tools/bench_configs.py irregular, tests/test_gpu_parity.py): degrees as asked for, no multi-edges, the
    same matrix for the same seed."""
    from ldpc_amd import codes
    h = codes.irregular_ldpc_code(600, 300, seed=3)
    assert h.shape == (300, 600) and int(h.data.max()) == 1
    rw = np.diff(h.indptr)
    assert sorted(set(rw.tolist())) == [3, 4, 5, 6, 7, 8, 9, 10, 12, 16]
    cw = np.bincount(h.indices, minlength=600)
    assert cw.min() >= 2 and cw.max() <= 8 and cw.sum() == rw.sum() == h.nnz
    assert (codes.irregular_ldpc_code(600, 300, seed=3) != h).nnz == 0
    assert (codes.irregular_ldpc_code(600, 300, seed=4) != h).nnz > 0


def test_schedules_soak_generator_and_checker(oracle_built):
    """The code generator of tests/fuzz_schedules.py stays inside what the on-chip serial_relative kernel takes (columns <= 8, rows <= 16
    entries) and the checker runs its cases (no GPU here: the soak itself is tests/test_gpu_fuzz_soak.py)."""
    import numpy as np
    import fuzz_schedules
    rng = np.random.default_rng(12)
    shapes = set()
    for _ in range(25):
        h = fuzz_schedules.random_code(rng)
        m, n = h.shape
        shapes.add((m, n))
        assert h.nnz > 0 and int(h.sum(0).max()) <= 8 and int(h.sum(1).max()) <= 16 and n <= 1024
    assert len(shapes) >= 6
    h = fuzz_schedules.random_code(np.random.default_rng(3))
    n = h.shape[1]
    o = oracle_built.BpOracle(h, error_rate=0.05, max_iter=6, bp_method=1, ms_scaling_factor=0.625)
    e = (np.random.default_rng(4).random((5, n)) < 0.05).astype(np.uint8)
    s = np.asarray((h @ e.T % 2).T, dtype=np.uint8)
    order = np.random.default_rng(5).integers(0, n, n).astype(np.int32)  # repeats allowed
    d, l, it, cv, last = o.decode_serial_relative_batch(s, order_state=order, fresh=True)
    assert d.shape == (5, n) and sorted(last.tolist()) == sorted(order.tolist())  # the order is rearranged, never changed as a multiset


def test_pinned_block_arrays_keep_the_block_alive():
    """ldpc_amd._lib.PinnedBlock: arrays on the block refer to it through their base, views of them too; the memory goes back when the
    last one is gone (here on ordinary memory behind a stand-in for the library: no GPU)."""
    import ctypes
    import gc
    import numpy as np
    from ldpc_amd._lib import PinnedBlock
    buf = ctypes.create_string_buffer(80)
    freed = []

    class FakeLib:
        def ldpc_hip_host_free(self, p):
            freed.append(p)

    blk = PinnedBlock(FakeLib(), ctypes.addressof(buf), 80)
    a = blk.array((2, 5), np.float64)
    assert not a.flags.owndata and a.flags.writeable and a.flags.c_contiguous and getattr(a.base, "owner", None) is blk
    a[:] = 1.5
    v = a[1]
    del a, blk
    gc.collect()
    assert freed == [] and v.tolist() == [1.5] * 5
    del v
    gc.collect()
    assert freed == [ctypes.addressof(buf)]
    import pytest
    with pytest.raises(ValueError):
        PinnedBlock(FakeLib(), ctypes.addressof(buf), 80).array((3, 5), np.float64)
