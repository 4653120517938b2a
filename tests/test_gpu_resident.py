"""A single decode() of a small code through the RESIDENT workgroup (csrc/host_onchip.h: decode_onchip_resident): the kernel that decoded
one syndrome stays for the next one.  Whatever happens between two calls -- nothing, a pause longer than the kernel lingers, new priors or
parameters, a batch call, another decoder, the decoder's destruction -- every call returns what the checker returns."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases(h, p, k, seed):
    rng = np.random.default_rng(seed)
    e = (rng.random((k, h.shape[1])) < p).astype(np.uint8)
    return np.ascontiguousarray((e @ h.T.toarray().astype(np.int64) % 2).astype(np.uint8))


@pytest.mark.parametrize("code", ["bb144", "hamming5", "surface7"])
def test_single_decodes_through_the_resident_workgroup(code, oracle_built):
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    from oracle import bits_equal
    h = {"bb144": codes.bivariate_bicycle_hx, "hamming5": lambda: codes.hamming_code(5), "surface7": lambda: codes.rotated_surface_code_x(7)}[code]()
    m, n = h.shape
    p, max_iter = 0.05, 30
    synd = _cases(h, p, 60, 3)
    synd[7] = 0
    synd[9, 0] = 2  # a byte > 1: never converges
    orc = oracle_built.BpOracle(h, error_rate=p, max_iter=max_iter, bp_method="product_sum")
    want = orc.decode_batch(synd)
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), max_iter, 0, 1.0)
    for rep in range(3):
        for k in range(len(synd)):
            if rep == 1 and k % 7 == 0:
                time.sleep(0.002)  # longer than the kernel lingers: it has left, the call launches a new one
            dec, llr, it, cv = eng.decode_batch(synd[k:k + 1], want_llr=(k % 3 != 0))
            assert np.array_equal(dec[0], want[0][k]) and int(it[0]) == int(want[2][k]) and bool(cv[0]) == bool(want[3][k]), (code, rep, k)
            if llr is not None:
                assert bits_equal(llr[0], want[1][k]), (code, rep, k)
            if rep == 2 and k % 11 == 0:  # a batch call in between (another kernel, the same host-mapped block)
                d2, l2, i2, c2 = eng.decode_batch(synd[:5])
                assert np.array_equal(d2, want[0][:5]) and bits_equal(l2, want[1][:5])
    # new priors, new parameters: the lingering kernel must not answer with the old ones
    probs = np.linspace(0.02, 0.09, n)
    eng.set_channel(probs)
    w2 = oracle_built.BpOracle(h, error_channel=probs, max_iter=max_iter, bp_method="product_sum").decode_batch(synd[:12])
    for k in range(12):
        dec, llr, it, cv = eng.decode_batch(synd[k:k + 1])
        assert np.array_equal(dec[0], w2[0][k]) and int(it[0]) == int(w2[2][k]) and bits_equal(llr[0], w2[1][k])
    eng.set_params(7, 0, 1.0)
    w3 = oracle_built.BpOracle(h, error_channel=probs, max_iter=7, bp_method="product_sum").decode_batch(synd[:12])
    for k in range(12):
        dec, llr, it, cv = eng.decode_batch(synd[k:k + 1])
        assert np.array_equal(dec[0], w3[0][k]) and int(it[0]) == int(w3[2][k]) and bits_equal(llr[0], w3[1][k])
    eng.set_debug_switch("RESIDENT", 0)  # and the ordinary path gives the same
    for k in range(12):
        dec, llr, it, cv = eng.decode_batch(synd[k:k + 1])
        assert np.array_equal(dec[0], w3[0][k]) and bits_equal(llr[0], w3[1][k])
    eng.close()


def test_two_decoders_and_destruction_while_the_kernel_lingers(oracle_built):
    from ldpc_amd import codes
    from ldpc_amd.bp_decoder import BpDecoder
    h1, h2 = codes.bivariate_bicycle_hx(), codes.hamming_code(4)
    s1, s2 = _cases(h1, 0.05, 40, 1), _cases(h2, 0.06, 40, 2)
    s1[s1.sum(axis=1) == 0, 0] = 1
    w1 = oracle_built.BpOracle(h1, error_rate=0.05, max_iter=50, bp_method="product_sum").decode_batch(s1)
    w2 = oracle_built.BpOracle(h2, error_rate=0.06, max_iter=15, bp_method="product_sum").decode_batch(s2)
    for _ in range(3):
        d1 = BpDecoder(h1, error_rate=0.05, max_iter=50, bp_method="product_sum", input_vector_type="syndrome")
        d2 = BpDecoder(h2, error_rate=0.06, max_iter=15, bp_method="product_sum", input_vector_type="syndrome")
        for k in range(40):
            assert np.array_equal(d1.decode(s1[k]), w1[0][k]) and d1.iter == int(w1[2][k]) and d1.converge == bool(w1[3][k])
            if s2[k].any():
                assert np.array_equal(d2.decode(s2[k]), w2[0][k]) and d2.iter == int(w2[2][k])
        del d1, d2  # (their kernels are still lingering)
