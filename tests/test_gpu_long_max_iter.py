"""The reference's default ``max_iter = n`` with one hopeless syndrome in a large batch: a product-sum batch on a
matrix without a fixed-degree ring variant must not queue thousands of full-size, empty per-pass rounds (host_stream.h: per_pass_first is
bounded by max_iter; beyond it the persistent kernel's hand-off parks at most 256 tiles).  Results = the CPU checker's, and the decode is
not slower than the same batch with a short max_iter by more than the hopeless row's own iterations cost."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_default_max_iter_with_one_hopeless_row_stays_bounded(oracle_built):
    import torch
    from golden_util import bits_equal
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    n, m, p = 600, 300, 0.01
    h = codes.irregular_ldpc_code(n, m, seed=5)
    B = 40000  # 625 tiles: beyond the 256-tile hand-off
    timings = {}
    outs = {}
    for max_iter in (24, n):
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), max_iter, 0, 1.0)
        eng.set_small_code_kernel(0)  # the streamed kernels (what a code beyond LDS takes)
        eng.set_repack(0)             # single pass: the path the advice is about
        s = eng.gen_bsc_syndromes(21, p, shot0=0, shots=B, device="cuda:0")
        s[12345, 7] = 2               # a syndrome byte > 1 never converges (bp.hpp:300): runs all max_iter iterations
        eng.decode_batch(s, want_llr=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = [x.cpu().numpy() for x in eng.decode_batch(s, want_llr=True)]
        torch.cuda.synchronize()
        timings[max_iter] = time.perf_counter() - t0
        outs[max_iter] = got
        rows = np.r_[0:40, 12340:12350, B - 40:B]
        want = oracle_built.BpOracle(h, error_rate=p, max_iter=max_iter, bp_method="product_sum").decode_batch(s.cpu().numpy()[rows])
        assert np.array_equal(got[0][rows], want[0]) and np.array_equal(got[2][rows], want[2]) and np.array_equal(got[3][rows], want[3])
        assert bits_equal(got[1][rows], want[1])
        assert got[2][12345] == max_iter and not got[3][12345]
        eng.close()
    conv = outs[24][3].astype(bool)
    assert conv.mean() > 0.99  # everything else converges within 24 iterations, so both settings decode those rows alike
    assert np.array_equal(outs[24][0][conv], outs[n][0][conv]) and np.array_equal(outs[24][2][conv], outs[n][2][conv])
    # 576 more iterations of ONE tile on the per-pass kernels (~4 launches of a 256-row grid each): well under a second; the unbounded
    # form queued 576 rounds x 4 launches x 625 rows of workgroups
    assert timings[n] < timings[24] + 1.5, timings
