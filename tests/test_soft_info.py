"""SoftInfoBpDecoder path: soft_info_decode_serial (bp.hpp:547-660).  CPU: the C restatement against goldens captured from
the real reference (tests/golden/make_golden_soft.py) and against the reference directly; GPU: ldpc_hip_bp_soft_info_decode_batch
and the SoftInfoBpDecoder mirror against the same goldens, bit for bit (minimum-sum arithmetic is exact)."""
import glob
import os
import zlib

import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from golden_util import GOLDEN_DIR, bits_equal

SOFT_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "soft_*.npz")))


def load_soft(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    m, n = int(z["m"]), int(z["n"])
    rp, ci = z["row_ptr"], z["col_idx"]
    h = sp.csr_matrix((np.ones(len(ci), np.uint8), ci, rp), shape=(m, n), dtype=np.uint8)
    crc = zlib.crc32(ci.tobytes(), zlib.crc32(rp.tobytes(), zlib.crc32(np.array([m, n], np.int64).tobytes())))
    assert np.uint32(crc) == z["h_crc"]
    return dict(h=h, m=m, n=n, channel_probs=z["channel_probs"], max_iter=int(z["max_iter"]),
                ms_scaling_factor=float(z["ms_scaling_factor"]), cutoff=float(z["cutoff"]), sigma=float(z["sigma"]),
                soft=z["soft_syndromes"], decoding=np.unpackbits(z["decoding"], axis=1, count=n), converge=z["converge"].astype(bool),
                iterations=z["iterations"].astype(np.int32), llr=z["llr"], soft_out=z["soft_out"])


def _same(got, c):
    dec, llr, it, cv, so = got
    assert np.array_equal(dec, c["decoding"]) and np.array_equal(np.asarray(cv, bool), c["converge"])
    assert np.array_equal(it, c["iterations"])
    assert bits_equal(llr, c["llr"]), "posterior log-ratios differ from the reference"
    assert bits_equal(so, c["soft_out"]), "soft syndrome after decoding differs from the reference"


def test_cases_present():
    assert len(SOFT_CASES) >= 10


@pytest.mark.parametrize("name", SOFT_CASES)
def test_oracle_reproduces_golden(name, oracle_built):
    c = load_soft(name)
    o = oracle_built.BpOracle(c["h"], error_channel=c["channel_probs"], max_iter=c["max_iter"], bp_method="minimum_sum",
                              ms_scaling_factor=c["ms_scaling_factor"])
    _same(o.soft_info_decode_batch(c["soft"], c["cutoff"], c["sigma"]), c)


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_vs_real_reference_random_and_custom_order():
    from ldpc_amd import codes
    rng = np.random.default_rng(5)
    for h in (codes.bivariate_bicycle_hx(), codes.hamming_code(4), codes.ring_code(9)):
        h = sp.csr_matrix(h)
        m, n = h.shape
        order = rng.permutation(n).astype(np.int32)
        for cutoff, sigma, alpha in ((3.0, 1.3, 0.8), (np.inf, 2.0, 1.0)):
            r = oracle.RefBp(h, error_rate=0.07, max_iter=9, bp_method="minimum_sum", ms_scaling_factor=alpha, schedule="serial")
            r.set_serial_order(order)
            o = oracle.BpOracle(h, error_rate=0.07, max_iter=9, bp_method="minimum_sum", ms_scaling_factor=alpha)
            soft = rng.normal(scale=2.0, size=(40, m))
            want = r.soft_info_decode_batch(soft, cutoff, sigma)
            got = o.soft_info_decode_batch(soft, cutoff, sigma, order=order)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3])
            assert bits_equal(got[1], want[1]) and bits_equal(got[4], want[4])


def test_mirror_validation():
    from ldpc_amd.bp_decoder import SoftInfoBpDecoder
    h = np.eye(3, dtype=int) + np.roll(np.eye(3, dtype=int), 1, axis=1)
    d = SoftInfoBpDecoder(h, error_rate=0.1, max_iter=3, ms_scaling_factor=1.0, cutoff=10.0)
    assert d.schedule == "serial" and d.bp_method == "minimum_sum" and d.sigma == 2.0 and d.cutoff == 10.0
    assert d.input_vector_type == "syndrome" and d.max_iter == 3
    assert SoftInfoBpDecoder(h, error_rate=0.1, sigma=2).sigma == 2.0  # `sigma: float` is a C double: an int converts (api_reference.json)
    with pytest.raises(TypeError, match="must be real number"):
        SoftInfoBpDecoder(h, error_rate=0.1, sigma="2")
    with pytest.raises(ValueError, match="sigma"):
        SoftInfoBpDecoder(h, error_rate=0.1, sigma=-1.0)
    with pytest.raises(ValueError):
        SoftInfoBpDecoder(h)  # no channel


@pytest.mark.gpu
@pytest.mark.parametrize("name", SOFT_CASES)
def test_device_reproduces_golden(name):
    from ldpc_amd.engine import HipBpEngine
    c = load_soft(name)
    h = c["h"]
    eng = HipBpEngine(h.indptr, h.indices, c["n"], c["channel_probs"], c["max_iter"], 1, c["ms_scaling_factor"])
    for serial_kernel in (-1, 0, 1):  # automatic, bit by bit, level-parallel
        eng.set_serial_kernel(serial_kernel)
        _same(eng.soft_info_decode_batch(c["soft"], c["cutoff"], c["sigma"]), c)
    import torch
    t = eng.soft_info_decode_batch(torch.from_numpy(c["soft"]).cuda(), c["cutoff"], c["sigma"])
    _same(tuple(x.cpu().numpy() for x in t), c)
    dec = eng.soft_info_decode_batch(c["soft"], c["cutoff"], c["sigma"], want_llr=False)
    assert dec[1] is None and np.array_equal(dec[0], c["decoding"])


@pytest.mark.gpu
def test_mirror_decode_and_batch_and_order(oracle_built):
    from ldpc_amd.bp_decoder import SoftInfoBpDecoder
    c = load_soft("soft_surface7_cut2")
    d = SoftInfoBpDecoder(c["h"], error_channel=list(c["channel_probs"]), max_iter=c["max_iter"],
                          ms_scaling_factor=c["ms_scaling_factor"], cutoff=c["cutoff"], sigma=c["sigma"])
    for b in (0, 5, 17):
        out = d.decode(c["soft"][b])
        assert out.dtype == np.uint8 and np.array_equal(out, c["decoding"][b])
        assert d.converge == bool(c["converge"][b]) and d.iter == int(c["iterations"][b])
        assert bits_equal(d.soft_syndrome, c["soft_out"][b]) and bits_equal(d.log_prob_ratios, c["llr"][b])
        assert np.array_equal(d.decoding, c["decoding"][b])
    batch = d.decode_batch(c["soft"])
    assert np.array_equal(batch, c["decoding"]) and bits_equal(d.soft_syndrome_batch, c["soft_out"])
    # a custom serial_schedule_order goes through the same setter as for BpDecoder
    order = np.random.default_rng(1).permutation(c["n"])
    d.serial_schedule_order = [int(v) for v in order]
    o = oracle_built.BpOracle(c["h"], error_channel=c["channel_probs"], max_iter=c["max_iter"], bp_method="minimum_sum",
                              ms_scaling_factor=c["ms_scaling_factor"])
    want = o.soft_info_decode_batch(c["soft"][:70], c["cutoff"], c["sigma"], order=order.astype(np.int32))
    got = d.decode_batch(c["soft"][:70])  # 70 rows: one full tile and a partial one
    assert np.array_equal(got, want[0]) and bits_equal(d.soft_syndrome_batch, want[4]) and np.array_equal(d.iter_batch, want[2])
    # random_serial_schedule on top of it: the custom order is what the reshuffles start from (bp.hpp:573-577; seed 0 as constructed)
    d.random_serial_schedule = True
    want = o.soft_info_decode_random_batch(c["soft"][:5], c["cutoff"], c["sigma"], 0, order.astype(np.int32))
    got = d.decode_batch(c["soft"][:5])
    assert np.array_equal(got, want[0]) and bits_equal(d.soft_syndrome_batch, want[4]) and np.array_equal(d.iter_batch, want[2])
    assert np.array_equal(d.serial_schedule_order, oracle_built.shuffle_orders_reseeded(0, c["n"], int(want[2][-1]), order.astype(np.int32))[1])
