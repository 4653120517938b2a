"""OSD on syndromes outside the image of H (rank-deficient H: toric-code X checks with one check flipped).

tests/golden/outimage_*.npz hold the REAL reference's outputs (tests/golden/make_golden_outimage.py) for rows inside and
outside the image.  Inside, the OSD output is unique given the column order and the device must reproduce it bit for bit;
outside, no x solves H x = s and the reference returns the solution of the subsystem of ITS pivot rows (a by-product of its
linked-list elimination's sparsity heuristic).  The device flags such rows (ldpc_hip_bposd_get_status == 2, include/ldpc_hip.h)
AND returns the reference's vector for them: a workgroup per flagged row re-enacts the reference's pivot-row choice
(ldpc_amd/csrc/osd_exact_kernel.h) and the ordinary OSD kernels run once more on the syndrome that keeps those rows -- for every
OSD kernel family: registers, LDS, workgroup (H in LDS / HBM scratch).
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
def test_oracle_matches_reference_on_every_row(name, oracle_built):
    """Inside AND outside the image: the restatement re-enacts the reference's pivot-row choice (oracle/bp_oracle.c,
    osd_reference_pivot_rows_syndrome), so it lands on the reference's vector for the unsolvable rows too."""
    c = load(name)
    o = oracle_built.BpOracle(c["h"], error_rate=c["p"], max_iter=c["max_iter"], bp_method=c["bp_method"], ms_scaling_factor=c["alpha"])
    dec, llr, it, cv = o.bposd_decode_batch(c["synd"], c["osd_method"], c["osd_order"])
    assert np.array_equal(cv, c["conv"]) and np.array_equal(it, c["it"]) and bits_equal(llr, c["llr"])
    assert np.array_equal(dec, c["dec"])


def _toric(L):
    rows, cols = [], []
    for r in range(L):
        for q in range(L):
            for e in (r * L + q, r * L + (q - 1) % L, L * L + r * L + q, L * L + ((r - 1) % L) * L + q):
                rows.append(r * L + q)
                cols.append(e)
    return sp.csr_matrix((np.ones(len(rows), np.uint8), (rows, cols)), shape=(L * L, 2 * L * L))


def _cases_outside(rng, h, count):
    m, n = h.shape
    for trial in range(count):
        e = (rng.random(n) < 0.08).astype(np.uint8)
        y = (h @ e % 2).astype(np.uint8)
        for _ in range(1 if trial % 3 else 3):
            y[rng.integers(m)] ^= 1  # odd number of flips: outside the image of the torus' checks
        llr = rng.normal(2.0, 1.5, size=n)
        if trial % 4 == 3:
            llr = np.round(llr)  # ties in the column order
        yield y, llr


@pytest.mark.parametrize("L", [3, 4, 5, 7])
def test_oracle_pivot_rows_against_the_real_reference(L, oracle_built):
    """Randomised: OSD alone (no BP), syndromes outside the image, log-ratios with and without ties; rank-deficient H with
    redundant rows of several kinds (a torus, and a torus with a duplicated and an all-zero row)."""
    if not oracle_built.have_ref():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(100 + L)
    h0 = _toric(L)
    h1 = sp.vstack([h0, h0[1], sp.csr_matrix((1, h0.shape[1]), dtype=np.uint8), h0[0]]).tocsr()
    for h in (h0, h1):
        ref = oracle_built.RefBpOsd(h, error_rate=0.05, max_iter=1, bp_method="minimum_sum")
        orc = oracle_built.BpOracle(h, error_rate=0.05, max_iter=1, bp_method="minimum_sum")
        for y, llr in _cases_outside(rng, h, 24):
            assert np.array_equal(orc.osd0(y, llr), ref.osd0(y, llr))


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [-1, 0, 2])
@pytest.mark.parametrize("name", CASES)
def test_device_flags_rows_outside_the_image_and_returns_the_reference_vector(name, kernel):
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
    assert np.array_equal(dec, c["dec"]), "rows outside the image: the solution on the reference's own pivot rows, bit for bit"
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
    assert np.array_equal(out, c["dec"])


@pytest.mark.gpu
@pytest.mark.parametrize("L", [3, 5, 8])
@pytest.mark.parametrize("osd", [(1, 0), (3, 4), (2, 3)])
def test_device_against_the_oracle_on_random_rank_deficient_systems(L, osd, oracle_built):
    """A torus with a duplicated, an all-zero and a repeated first row; syndromes with 1 .. 3 faulty bits; BP given one iteration so
    that nearly every row goes to OSD.  Decisions of every row against the CPU restatement (itself pinned to the real reference above)."""
    from ldpc_amd.engine import HipBpEngine
    rng = np.random.default_rng(7 * L + osd[0])
    h0 = _toric(L)
    h = sp.vstack([h0, h0[1], sp.csr_matrix((1, h0.shape[1]), dtype=np.uint8), h0[0]]).tocsr()
    m, n = h.shape
    probs = rng.uniform(0.02, 0.2, size=n)
    synd = np.stack([y for y, _ in _cases_outside(rng, h, 150)])
    synd[::5] = (h @ (rng.random((n, 30)) < 0.1).astype(np.uint8) % 2).T  # some rows inside the image
    eng = HipBpEngine(h.indptr, h.indices, n, probs, 1, 1, 0.75)
    eng.set_osd(*osd)
    dec, llr, it, cv = eng.decode_batch(synd, osd=True)
    o = oracle_built.BpOracle(h, error_channel=probs, max_iter=1, bp_method="minimum_sum", ms_scaling_factor=0.75)
    want = o.bposd_decode_batch(synd, osd[0], osd[1])
    assert np.array_equal(cv, want[3]) and bits_equal(llr, want[1])
    assert np.array_equal(dec, want[0])
    assert (eng.osd_status(len(synd)) == 2).sum() > 50
