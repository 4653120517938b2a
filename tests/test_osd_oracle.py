"""OSD-0 checker (oracle/bp_oracle.c: osd0_oracle) against the reference: golden fixtures and, when the real
reference build is present, direct comparison on adversarial log-ratio vectors (exact ties, +-inf)."""
import numpy as np
import pytest

import oracle
from golden_util import load_case, osd_case_names


@pytest.mark.parametrize("name", osd_case_names())
def test_oracle_bposd0_reproduces_golden(name, oracle_built):
    c = load_case(name)
    o = oracle_built.BpOracle(c["h"], error_channel=c["channel_probs"], max_iter=c["max_iter"], bp_method=c["bp_method"],
                              ms_scaling_factor=c["ms_scaling_factor"])
    dec, llr, it, cv = o.bposd0_decode_batch(c["syndromes"])
    assert np.array_equal(dec, c["decoding"])
    assert np.array_equal(cv, c["converge"]) and np.array_equal(it, c["iterations"])
    # every OSD-0 output reproduces its syndrome (reference property tests, cpp_test/TestOsdDecoder.cpp:9-35)
    chk = (dec.astype(np.int64) @ c["h"].T.toarray().astype(np.int64)) % 2
    assert np.array_equal(chk, c["syndromes"])
    assert (~cv).sum() > 0, "fixture should exercise OSD"


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_osd0_alone_vs_real_reference_with_ties_and_infinities(oracle_built):
    from ldpc_amd import codes
    rng = np.random.default_rng(0)
    for h in (codes.bivariate_bicycle_hx(), codes.hamming_code(5), codes.rotated_surface_code_x(5), codes.ring_code(12)):
        m, n = h.shape
        o = oracle.BpOracle(h, error_rate=0.05, max_iter=3)
        r = oracle.RefBpOsd(h, error_rate=0.05, max_iter=3)
        for t in range(120):
            e = (rng.random(n) < 0.1).astype(np.uint8)
            s = np.asarray(h @ e % 2, dtype=np.uint8).ravel()
            kind = t % 4
            if kind == 0:
                llr = rng.normal(size=n)
            elif kind == 1:
                llr = rng.integers(-2, 3, size=n).astype(float)  # many exact ties
            elif kind == 2:
                llr = np.where(rng.random(n) < 0.1, np.inf, rng.integers(0, 3, size=n).astype(float))
                llr[rng.random(n) < 0.05] = -np.inf
            else:
                llr = np.zeros(n)
            assert np.array_equal(o.osd0(s, llr), r.osd0(s, llr))


from golden_util import osdw_case_names  # noqa: E402


@pytest.mark.parametrize("name", osdw_case_names())
def test_oracle_bposdw_reproduces_golden(name, oracle_built):
    """oracle/bp_oracle.c: osdw_oracle (OSD_E / OSD_CS, osd.hpp:119-187 restated) vs the real reference's captures."""
    c = load_case(name)
    o = oracle_built.BpOracle(c["h"], error_channel=c["channel_probs"], max_iter=c["max_iter"], bp_method=c["bp_method"],
                              ms_scaling_factor=c["ms_scaling_factor"])
    dec, llr, it, cv = o.bposd_decode_batch(c["syndromes"], c["osd_method"], c["osd_order"])
    assert np.array_equal(dec, c["decoding"])
    assert np.array_equal(cv, c["converge"]) and np.array_equal(it, c["iterations"])
    chk = (dec.astype(np.int64) @ c["h"].T.toarray().astype(np.int64)) % 2
    assert np.array_equal(chk, c["syndromes"])  # cpp_test/TestOsdDecoder.cpp:37-156: every OSD output solves H x = s
    dec0 = o.bposd_decode_batch(c["syndromes"], 1, 0)[0]
    assert np.array_equal(dec0, c["osd0_decoding"])
    # the sweep can only lower the weight sum_j x_j log(1 / p_j) (osd.hpp:177)
    wt = np.log(1 / c["channel_probs"])
    assert np.all(dec @ wt <= dec0 @ wt + 1e-9)


def test_osdw_goldens_exercise_the_sweep():
    changed = sum(int((load_case(n)["decoding"] != load_case(n)["osd0_decoding"]).any(axis=1).sum()) for n in osdw_case_names())
    assert changed > 100


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_osdw_alone_vs_real_reference(oracle_built):
    """OSD_E / OSD_CS on adversarial log-ratios (ties, infinities) and non-uniform priors, straight against osd.hpp."""
    from ldpc_amd import codes
    rng = np.random.default_rng(2)
    for h, settings in ((codes.bivariate_bicycle_hx(), [(3, 7), (2, 6)]), (codes.hamming_code(4), [(3, 4), (2, 12)]),
                        (codes.rotated_surface_code_x(5), [(3, 9), (2, 5)])):
        m, n = h.shape
        chan = 0.01 + 0.2 * rng.random(n)
        o = oracle.BpOracle(h, error_channel=chan, max_iter=1)
        for method, order in settings:
            r = oracle.RefBpOsd(h, error_channel=chan, max_iter=1, osd_method=method, osd_order=order)
            lib = r.lib
            for t in range(40):
                e = (rng.random(n) < 0.15).astype(np.uint8)
                s = np.asarray(h @ e % 2, dtype=np.uint8).ravel()
                llr = [rng.normal(size=n), rng.integers(-2, 3, size=n).astype(float), np.zeros(n),
                       np.where(rng.random(n) < 0.1, np.inf, rng.integers(0, 3, size=n).astype(float))][t % 4]
                assert np.array_equal(o.osdw(s, llr, method, order)[0], r.osd0(s, llr))  # ref_osd0 = OsdDecoder::decode


from golden_util import serial_case_names  # noqa: E402


@pytest.mark.parametrize("name", serial_case_names())
def test_oracle_serial_schedule_reproduces_golden(name, oracle_built):
    """oracle/bp_oracle.c: bp_oracle_decode_serial vs bp_decode_serial outputs captured from the real reference."""
    from golden_util import bits_equal
    c = load_case(name)
    o = oracle_built.BpOracle(c["h"], error_channel=c["channel_probs"], max_iter=c["max_iter"], bp_method=c["bp_method"],
                              ms_scaling_factor=c["ms_scaling_factor"])
    dec, llr, it, cv = o.decode_serial_batch(c["syndromes"], c["order"])
    assert np.array_equal(dec, c["decoding"]) and np.array_equal(cv, c["converge"]) and np.array_equal(it, c["iterations"])
    assert bits_equal(llr[: len(c["llr"])], c["llr"]) or np.allclose(llr[: len(c["llr"])], c["llr"], rtol=1e-9, equal_nan=True)
