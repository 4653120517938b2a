"""The quantum-code matrices of the reference's own tests (python_test/pcms: HGP [[400,16,6]], toric d=20, planar surface
d=20) with the decoder settings of python_test/test_qcodes.py:110-186 (min-sum 0.625 / product-sum, 5 iterations,
parallel / serial schedule, OSD_0 / OSD_CS 3 / OSD_E 3): outputs captured from the real reference
(tests/golden/make_golden_qcodes.py), reproduced by the oracle on the CPU and by the device path, including the
logical-failure flags `lx (decoding + error) != 0` that test_qcodes counts."""
import glob
import os
import zlib

import numpy as np
import pytest
import scipy.sparse as sp

from golden_util import GOLDEN_DIR

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "qcodes_*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    m, n, k = int(z["m"]), int(z["n"]), int(z["k"])
    rp, ci = z["row_ptr"], z["col_idx"]
    hx = sp.csr_matrix((np.ones(len(ci), np.uint8), ci, rp), shape=(m, n), dtype=np.uint8)
    crc = zlib.crc32(ci.tobytes(), zlib.crc32(rp.tobytes(), zlib.crc32(np.array([m, n], np.int64).tobytes())))
    assert np.uint32(crc) == z["h_crc"]
    lx = sp.csr_matrix((np.ones(len(z["lx_col_idx"]), np.uint8), z["lx_col_idx"], z["lx_row_ptr"]), shape=(k, n), dtype=np.uint8)
    err = np.unpackbits(z["errors"], axis=1, count=n)
    return dict(hx=hx, lx=lx, m=m, n=n, p=float(z["error_rate"]), max_iter=int(z["max_iter"]), bp_method=str(z["bp_method"]),
                alpha=float(z["ms_scaling_factor"]), schedule=str(z["schedule"]), osd_method=int(z["osd_method"]),
                osd_order=int(z["osd_order"]), err=err, synd=np.ascontiguousarray((hx @ err.T % 2).T.astype(np.uint8)),
                decoding=np.unpackbits(z["decoding"], axis=1, count=n), converge=z["converge"].astype(bool),
                iterations=z["iterations"].astype(np.int32), llr_rowsum=z["llr_rowsum"], logical_fail=z["logical_fail"].astype(bool))


def _logical_fail(c, dec):
    return np.asarray((c["lx"] @ (dec ^ c["err"]).T % 2).T, dtype=np.uint8).any(axis=1)


def test_cases_present():
    assert len(CASES) == 15


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_reference(name, oracle_built):
    c = load(name)
    o = oracle_built.BpOracle(c["hx"], error_rate=c["p"], max_iter=c["max_iter"], bp_method=c["bp_method"], ms_scaling_factor=c["alpha"])
    if c["schedule"] == "serial":
        dec, llr, it, cv = o.decode_serial_batch(c["synd"], None)
        for b in np.flatnonzero(~cv):  # BpOsdDecoder.decode: OSD on what BP left (pyx:125-134)
            dec[b] = o.osdw(c["synd"][b], llr[b], c["osd_method"], c["osd_order"])[0]
    else:
        dec, llr, it, cv = o.bposd_decode_batch(c["synd"], c["osd_method"], c["osd_order"])
    assert np.array_equal(dec, c["decoding"]) and np.array_equal(cv, c["converge"]) and np.array_equal(it, c["iterations"])
    assert np.array_equal(_logical_fail(c, dec), c["logical_fail"])
    assert np.array_equal((c["hx"] @ dec.T % 2).T, c["synd"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_reproduces_reference(name):
    from ldpc_amd.engine import HipBpEngine
    c = load(name)
    hx = c["hx"]
    eng = HipBpEngine(hx.indptr, hx.indices, c["n"], np.full(c["n"], c["p"]), c["max_iter"], 0 if c["bp_method"] == "product_sum" else 1, c["alpha"])
    eng.set_schedule(c["schedule"])
    eng.set_osd(c["osd_method"], c["osd_order"])
    dec, llr, it, cv = eng.decode_batch(c["synd"], osd=True)
    assert np.array_equal(cv, c["converge"]) and np.array_equal(it, c["iterations"])
    bad = np.flatnonzero((dec != c["decoding"]).any(axis=1))
    assert bad.size == 0, f"{bad.size} rows differ from the reference (first {bad[:5]})"
    assert np.array_equal(_logical_fail(c, dec), c["logical_fail"])
    # bit-packed in, predicted observables out: lx x for every shot, as the sinter decoders compute it
    eng.set_observables(c["lx"])
    obs = eng.decode_b8(np.packbits(c["synd"], axis=1, bitorder="little"), with_osd=True)[0]
    want = np.packbits(np.asarray((c["lx"] @ c["decoding"].T % 2).T, dtype=np.uint8), axis=1, bitorder="little")
    zero = ~c["synd"].any(axis=1)
    assert np.array_equal(obs[~zero], want[~zero])


@pytest.mark.gpu
def test_bposd_decoder_on_the_hgp_code_like_test_qcodes():
    """test_qcodes.py's loop (:34-70) in batch form: decode, residual, logical check."""
    from ldpc_amd.bposd_decoder import BpOsdDecoder
    c = load("qcodes_400_16_6_ms_par_cs3")
    d = BpOsdDecoder(c["hx"], error_rate=c["p"], max_iter=5, bp_method="ms", ms_scaling_factor=0.625, schedule="parallel",
                     osd_method="osd_cs", osd_order=3)
    dec = d.decode_batch(c["synd"])
    nz = c["synd"].any(axis=1)
    assert np.array_equal(dec[nz], c["decoding"][nz])
    fails = _logical_fail(c, dec)
    assert int(fails.sum()) == int(c["logical_fail"].sum()) == 5
    ds = BpOsdDecoder(c["hx"], error_rate=c["p"], max_iter=5, bp_method="ms", ms_scaling_factor=0.625, schedule="serial", osd_method="osd0")
    cs = load("qcodes_400_16_6_ms_ser_osd0")
    assert np.array_equal(ds.decode_batch(cs["synd"])[nz], cs["decoding"][nz])
