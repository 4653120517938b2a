"""OSD on syndromes outside the image of H (rank-deficient H: toric-code X checks with one check flipped).

tests/golden/outimage_*.npz hold the REAL reference's outputs (tests/golden/make_golden_outimage.py) for rows inside and
outside the image.  Inside, the OSD output is unique given the column order and the device must reproduce it bit for bit;
outside, no x solves H x = s, the reference returns the solution of the subsystem of ITS pivot rows (a by-product of its
linked-list elimination's sparsity heuristic), and the device flags the row instead (ldpc_hip_bposd_get_status == 2,
include/ldpc_hip.h) -- for every OSD kernel family: registers, LDS, workgroup (H in LDS / HBM scratch).
"""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp

from golden_util import GOLDEN_DIR, bits_equal

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "outimage_*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    m, n = int(z["m"]), int(z["n"])
    h = sp.csr_matrix((np.ones(len(z["col_idx"]), np.uint8), z["col_idx"], z["row_ptr"]), shape=(m, n))
    return dict(h=h, m=m, n=n, p=float(z["p"]), max_iter=int(z["max_iter"]), bp_method=int(z["bp_method"]), alpha=float(z["ms_scaling_factor"]),
                osd_method=int(z["osd_method"]), osd_order=int(z["osd_order"]), synd=np.unpackbits(z["syndromes"], axis=1, count=m),
                inside=z["inside_image"].astype(bool), dec=np.unpackbits(z["decoding"], axis=1, count=n), conv=z["converge"].astype(bool),
                it=z["iterations"].astype(np.int32), llr=z["llr"])


def test_fixtures_present_and_consistent():
    assert len(CASES) >= 4
    for name in CASES:
        c = load(name)
        resid = ((c["h"].astype(np.int64) @ c["dec"].T.astype(np.int64)).T % 2) != c["synd"]
        assert not resid[c["inside"]].any(), "inside the image the reference's output solves the syndrome"
        assert resid[~c["inside"]].any(axis=1).all(), "outside the image nothing can"
        assert not c["conv"][~c["inside"]].any()


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_inside_the_image(name, oracle_built):
    c = load(name)
    o = oracle_built.BpOracle(c["h"], error_rate=c["p"], max_iter=c["max_iter"], bp_method=c["bp_method"], ms_scaling_factor=c["alpha"])
    dec, llr, it, cv = o.bposd_decode_batch(c["synd"], c["osd_method"], c["osd_order"])
    assert np.array_equal(cv, c["conv"]) and np.array_equal(it, c["it"]) and bits_equal(llr, c["llr"])
    assert np.array_equal(dec[c["inside"]], c["dec"][c["inside"]])


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [-1, 0, 2])
@pytest.mark.parametrize("name", CASES)
def test_device_flags_rows_outside_the_image(name, kernel):
    from ldpc_amd.engine import HipBpEngine
    c = load(name)
    eng = HipBpEngine(c["h"].indptr, c["h"].indices, c["n"], np.full(c["n"], c["p"]), c["max_iter"], c["bp_method"], c["alpha"])
    eng.set_osd(c["osd_method"], c["osd_order"])
    eng.set_osd_kernel(kernel)
    dec, llr, it, cv = eng.decode_batch(c["synd"], osd=True)
    status = eng.osd_status(len(c["synd"]))
    assert np.array_equal(cv, c["conv"]) and np.array_equal(it, c["it"]) and bits_equal(llr, c["llr"])  # BP's part, every row
    assert np.array_equal(status == 0, c["conv"])
    assert np.array_equal(status == 2, ~c["conv"] & ~c["inside"])
    assert np.array_equal(status == 1, ~c["conv"] & c["inside"])
    ok = status < 2
    assert np.array_equal(dec[ok], c["dec"][ok]), "rows inside the image: the reference's OSD output, bit for bit"
    # rows outside: deterministic, and a solution of as many checks as a rank-deficient system allows (all but the dependent one)
    again = eng.decode_batch(c["synd"], osd=True)[0]
    assert np.array_equal(again, dec)
    resid = (((c["h"].astype(np.int64) @ dec.T.astype(np.int64)).T % 2) != c["synd"]).sum(axis=1)
    assert (resid[status == 2] >= 1).all()
    import torch
    d = eng.decode_batch(torch.from_numpy(c["synd"]).cuda(), osd=True)
    st = torch.empty(len(c["synd"]), dtype=torch.uint8, device="cuda")
    from ldpc_amd import _lib
    _lib.check(eng._lib.ldpc_hip_bposd_get_status(eng._h, st.data_ptr(), len(c["synd"])))
    assert np.array_equal(st.cpu().numpy(), status) and np.array_equal(d[0].cpu().numpy(), dec)
    with pytest.raises(_lib.LdpcHipError):
        eng.osd_status(len(c["synd"]) + 1)


@pytest.mark.gpu
def test_bposd_decoder_reports_status():
    from ldpc_amd.bposd_decoder import BpOsdDecoder
    c = load("outimage_toric6_osd0_ms")
    d = BpOsdDecoder(c["h"], error_rate=c["p"], max_iter=c["max_iter"], bp_method="ms", ms_scaling_factor=c["alpha"], osd_method="osd_0")
    out = d.decode_batch(c["synd"])
    assert np.array_equal(d.osd_status_batch == 2, ~c["conv"] & ~c["inside"])
    assert np.array_equal(out[d.osd_status_batch < 2], c["dec"][d.osd_status_batch < 2])
