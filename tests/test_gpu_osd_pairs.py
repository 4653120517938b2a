"""OSD_CS orders around the places where the pair number -> (i, j) routine changes its method (csrc/osd_kernels.h: osd_pair_of walks the
rows up to order 48 and inverts the triangular number above; the kernels change from masks to named columns at 64): the decisions of
every row against the CPU checker (osd.hpp:91-99 through oracle/), on the BB [[144,12,12]] code (k = n - rank = 78 non-pivot columns)
and on a hypergraph-product code with a few hundred, through every OSD kernel the dispatch offers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("order", [47, 48, 49, 57, 64, 65, 78, 300])
def test_osd_cs_orders_around_the_method_changes(order, oracle_built):
    from ldpc_amd.engine import HipBpEngine
    from ldpc_amd.codes import bivariate_bicycle_hx
    h = bivariate_bicycle_hx()
    p = 0.09
    eng = HipBpEngine(h.indptr, h.indices, 144, np.full(144, p), 4, 1, 0.625)
    s = eng.gen_bsc_syndromes(11, p, shot0=0, shots=192, device="cuda:0")
    sh = s.cpu().numpy()
    want = oracle_built.BpOracle(h, error_rate=p, max_iter=4, bp_method="minimum_sum", ms_scaling_factor=0.625).bposd_decode_batch(sh, 3, order, want_llr=False)
    assert (~want[3].astype(bool)).sum() > 40, "most rows are meant to reach OSD"
    eng.set_osd(3, order)
    for osd_kernel in (-1, 0, 2):
        eng.set_osd_kernel(osd_kernel)
        got = eng.decode_batch(s, want_llr=False, osd=True)[0].cpu().numpy()
        assert np.array_equal(got, want[0]), f"order {order} kernel {osd_kernel}"
    eng.close()


@pytest.mark.parametrize("order", [50, 130])
def test_osd_cs_on_a_hypergraph_product_code(order, oracle_built):
    from ldpc_amd.engine import HipBpEngine
    from ldpc_amd import codes
    h = codes.hypergraph_product_hx(codes.regular_ldpc_code(16, 3, 4, seed=5))  # 192 x 400
    m, n = h.shape
    p = 0.05
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), 3, 1, 0.625)
    s = eng.gen_bsc_syndromes(5, p, shot0=0, shots=24, device="cuda:0")
    sh = s.cpu().numpy()
    want = oracle_built.BpOracle(h, error_rate=p, max_iter=3, bp_method="minimum_sum", ms_scaling_factor=0.625).bposd_decode_batch(sh, 3, order, want_llr=False)
    eng.set_osd(3, order)
    got = eng.decode_batch(s, want_llr=False, osd=True)[0].cpu().numpy()
    assert np.array_equal(got, want[0])
    eng.close()
