"""Bit-packed ("b8") shot I/O: ldpc_hip_bp_decode_b8 and the sinter-style file decoder built on it
(reference: sinter_decoders/sinter_bposd_decoder.py:57-130)."""
import numpy as np
import pytest
import scipy.sparse as sp

from ldpc_amd import codes
from ldpc_amd.sinter_decoders import SinterBpOsdDecoder, read_b8, write_b8


def _pack(bits):
    return np.packbits(bits, axis=1, bitorder="little")


def test_b8_file_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    bits = (rng.random((37, 21)) < 0.3).astype(np.uint8)
    path = tmp_path / "x.b8"
    write_b8(path, _pack(bits))
    assert path.stat().st_size == 37 * 3
    back = read_b8(path, 21, 37)
    assert np.array_equal(np.unpackbits(back, axis=1, bitorder="little", count=21), bits)
    with pytest.raises(ValueError):
        read_b8(path, 21, 36)
    with pytest.raises(ValueError):
        read_b8(path, 9)  # 2-byte shots do not divide 111 bytes


def test_sinter_decoder_keywords_and_model_check(tmp_path):
    d = SinterBpOsdDecoder()
    assert (d.max_iter, d.bp_method, d.ms_scaling_factor, d.schedule, d.osd_method, d.osd_order) == (0, "ms", 0.625, "parallel", "osd0", 0)
    (tmp_path / "a.dem").write_text("error(0.1) D0 D1 L0\nerror(0.1) D1\n")  # read without stim; 2 detectors, 1 observable
    with pytest.raises(ValueError, match="does not match num_dets / num_obs"):
        d.decode_via_files(num_shots=1, num_dets=3, num_obs=1, dem_path=tmp_path / "a.dem", dets_b8_in_path=tmp_path / "d.b8",
                           obs_predictions_b8_out_path=tmp_path / "o.b8", tmp_dir=tmp_path)


def _observables(n, k, seed):
    rng = np.random.default_rng(seed)
    return sp.csr_matrix((rng.random((k, n)) < 0.1).astype(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("osd", [None, ("osd_0", 0), ("osd_cs", 6)])
def test_decode_b8_matches_the_unpacked_pipeline(osd):
    from ldpc_amd.engine import HipBpEngine
    from ldpc_amd.noise_models import generate_bsc_batch
    h = sp.csr_matrix(codes.bivariate_bicycle_hx())
    m, n = h.shape
    obs = _observables(n, 12, 1)
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, 0.06), 12, 0, 1.0)
    e = generate_bsc_batch(n, 0.06, 3, 0, 1000)
    e[::50] = 0  # all-zero shots take the Python-level shortcut in the reference
    s = np.ascontiguousarray((h @ e.T % 2).T.astype(np.uint8))
    if osd is not None:
        eng.set_osd({"osd_0": 1, "osd_cs": 3}[osd[0]], osd[1])
    dec, _, it, cv = eng.decode_batch(s, want_llr=False, osd=osd is not None)
    zero = ~s.any(axis=1)
    dec[zero] = 0
    want_obs = _pack(np.asarray(obs @ dec.T % 2, dtype=np.uint8).T)
    eng.set_observables(obs)
    got_obs, got_dec, git, gcv = eng.decode_b8(_pack(s), with_osd=osd is not None, want_decoding=True)
    assert got_obs.shape == (1000, 2) and got_dec.shape == (1000, 18)
    assert np.array_equal(got_obs, want_obs)
    assert np.array_equal(got_dec, _pack(dec))
    assert np.array_equal(gcv[~zero], cv[~zero]) and gcv[zero].all() and not git[zero].any()
    assert np.array_equal(git[~zero], it[~zero])
    # device tensors
    import torch
    tobs, tdec, _, _ = eng.decode_b8(torch.from_numpy(_pack(s)).cuda(), with_osd=osd is not None, want_decoding=True)
    assert np.array_equal(tobs.cpu().numpy(), want_obs) and np.array_equal(tdec.cpu().numpy(), _pack(dec))


@pytest.mark.gpu
def test_decode_b8_files_against_per_shot_reference_semantics(tmp_path):
    """The file decoder reproduces `(observables_matrix @ BpOsdDecoder.decode(shot)) % 2` shot by shot."""
    from ldpc_amd.bposd_decoder import BpOsdDecoder
    from ldpc_amd.noise_models import generate_bsc_batch
    from ldpc_amd.sinter_decoders import decode_b8_files
    h = sp.csr_matrix(codes.rotated_surface_code_x(7))
    m, n = h.shape
    obs = _observables(n, 3, 2)
    priors = 0.01 + 0.08 * np.random.default_rng(3).random(n)
    e = generate_bsc_batch(n, 0.06, 9, 0, 300)
    s = np.ascontiguousarray((h @ e.T % 2).T.astype(np.uint8))
    write_b8(tmp_path / "dets.b8", _pack(s))
    decode_b8_files(h, priors, obs, num_shots=300, dets_b8_in_path=tmp_path / "dets.b8",
                    obs_predictions_b8_out_path=tmp_path / "obs.b8", max_iter=10, bp_method="ms", ms_scaling_factor=0.625,
                    osd_method="osd_cs", osd_order=5)
    got = np.unpackbits(read_b8(tmp_path / "obs.b8", 3, 300), axis=1, bitorder="little", count=3)
    d = BpOsdDecoder(h, error_channel=list(priors), max_iter=10, bp_method="ms", ms_scaling_factor=0.625, osd_method="osd_cs",
                     osd_order=5)
    for b in range(0, 300, 7):
        assert np.array_equal(got[b], (obs @ d.decode(s[b])) % 2)


@pytest.mark.gpu
def test_b8_argument_errors():
    from ldpc_amd._lib import LdpcHipError
    from ldpc_amd.engine import HipBpEngine
    h = sp.csr_matrix(codes.hamming_code(3))
    eng = HipBpEngine(h.indptr, h.indices, 7, np.full(7, 0.1), 5, 0, 1.0)
    with pytest.raises(ValueError):
        eng.decode_b8(np.zeros((2, 1), np.uint8))  # observables not set
    with pytest.raises(ValueError):
        eng.set_observables(np.zeros((2, 6), np.uint8))
    eng.set_observables(np.array([[1, 1, 0, 0, 0, 0, 0]], np.uint8))
    with pytest.raises(ValueError):
        eng.decode_b8(np.zeros((2, 2), np.uint8))
    obs, _, it, cv = eng.decode_b8(np.array([[0b101], [0]], np.uint8))
    assert obs.shape == (2, 1) and cv[1] and it[1] == 0


@pytest.mark.gpu
def test_pack_unpack_b8_on_device():
    import torch
    from ldpc_amd.engine import HipBpEngine
    h = sp.csr_matrix(codes.hamming_code(3))
    eng = HipBpEngine(h.indptr, h.indices, 7, np.full(7, 0.1), 5, 0, 1.0)
    rng = np.random.default_rng(1)
    for bits in (1, 7, 8, 9, 441, 10000):
        x = (rng.random((67, bits)) < 0.4).astype(np.uint8)
        packed = eng.pack_b8(torch.from_numpy(x).cuda())
        assert np.array_equal(packed.cpu().numpy(), _pack(x))
        assert np.array_equal(eng.unpack_b8(packed, bits).cpu().numpy(), x)
