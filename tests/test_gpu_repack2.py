"""The second compaction of the streamed two-pass decode (csrc/host_stream.h: decode_stream_repacked, round 6): first pass k1 iterations, the
rows still decoding compacted lane by lane into dense tiles, k2 more iterations, the rows STILL decoding compacted once more, a third pass to the
end -- against the plain decode (no pass structure at all), the two-pass decode, and the CPU checker: every row bit for bit, regular (ring
variant) and irregular (per-pass kernels) codes, both methods, with and without log-ratios, a hopeless row, a partial last tile."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _decode(eng, s, **kw):
    return [x.cpu().numpy() if x is not None and hasattr(x, "cpu") else x for x in eng.decode_batch(s, **kw)]


@pytest.mark.parametrize("code,method,alpha,p,max_iter", [("ldpc36", 0, 1.0, 0.055, 30), ("ldpc36", 1, 0.8, 0.05, 40), ("irregular", 0, 1.0, 0.035, 24), ("ldpc48", 1, 0.0, 0.04, 30)])
def test_second_compaction_gives_the_plain_decodes_bits(code, method, alpha, p, max_iter, oracle_built):
    from golden_util import bits_equal
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    n = 600
    h = {"ldpc36": lambda: codes.regular_ldpc_code(n, 3, 6, seed=3), "ldpc48": lambda: codes.regular_ldpc_code(n, 4, 8, seed=3),
         "irregular": lambda: codes.irregular_ldpc_code(n, n // 2, seed=3)}[code]()
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), max_iter, method, alpha)
    eng.set_small_code_kernel(0)   # the streamed kernels (what a code beyond LDS takes)
    B = 40000 + 37                 # 626 tiles, the last one partial
    s = eng.gen_bsc_syndromes(5, p, shot0=0, shots=B, device="cuda:0")
    s[777, 3] = 2                  # never converges: in every pass to the end
    eng.set_repack(0)
    ref = _decode(eng, s, want_llr=True)
    assert 0.5 < ref[3].mean() < 0.9999 and ref[2][ref[3].astype(bool)].min() < ref[2][ref[3].astype(bool)].max()
    rows = np.r_[0:50, 770:790, B - 40:B]
    name = "product_sum" if method == 0 else "minimum_sum"
    want = oracle_built.BpOracle(h, error_rate=p, max_iter=max_iter, bp_method=name, ms_scaling_factor=alpha).decode_batch(s.cpu().numpy()[rows])
    assert np.array_equal(ref[0][rows], want[0]) and np.array_equal(ref[2][rows], want[2]) and np.array_equal(ref[3][rows].astype(bool), want[3].astype(bool))
    assert bits_equal(ref[1][rows], want[1])
    for k1 in (2, 3, 5):
        for k2 in (0, 1, 2, 3, 4):
            eng.set_repack(k1)
            eng.set_debug_switch("REPACK2", k2)
            for want_llr in (True, False):
                got = _decode(eng, s, want_llr=want_llr)
                tag = (code, method, k1, k2, want_llr)
                assert np.array_equal(got[0], ref[0]) and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]), tag
                assert got[1] is None if not want_llr else bits_equal(got[1], ref[1]), tag
    # steered by the histogram of the previous decode (repack -1): whatever it chooses, the same bits
    eng.set_repack(-1)
    eng.set_debug_switch("REPACK2", 4)
    for _ in range(3):
        got = _decode(eng, s, want_llr=True)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]) and bits_equal(got[1], ref[1])
    eng.close()
