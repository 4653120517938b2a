"""Overlapping-window decoding on the device against the committed fixtures (window decodes by the real reference) and
against the shot-by-shot checker (oracle/window_oracle.py) on fresh shots."""
import numpy as np
import pytest

from test_ckt_noise_host import _window_fixture, window_fixtures
from window_util import phenomenological_dem, phenomenological_matrices, ring_code, sample_shots

pytestmark = pytest.mark.gpu


def _decoder(fx, model=None):
    from ldpc_amd.ckt_noise import BpOsdOverlappingWindowDecoder
    return BpOsdOverlappingWindowDecoder(model if model is not None else fx["text"], decodings=fx["decodings"], window=fx["window"],
                                         commit=fx["commit"], num_checks=fx["num_checks"], decoder_config=fx["cfg"])


@pytest.mark.parametrize("path", window_fixtures(), ids=lambda p: p.split("/")[-1][:-4])
def test_window_fixture(path):
    fx = _window_fixture(path)
    dec = _decoder(fx)
    shots = fx["shots"].copy()
    corrs = dec._corr_multiple_rounds_batch(shots)
    assert np.array_equal(corrs, fx["corrections"])
    assert np.array_equal(shots, fx["shots_after"]), "the caller's syndromes carry the committed corrections afterwards"
    assert np.array_equal(dec.dem_matrices.priors, fx["priors_after"])
    # a second pass reuses the window decoders (built with the priors as they were on first use)
    preds = dec.decode_batch(fx["shots"].copy())
    assert preds.dtype == bool and np.array_equal(preds, fx["predictions"])
    packed = dec.decode_batch(np.packbits(fx["shots"], axis=1, bitorder="little"), bit_packed_shots=True, bit_packed_predictions=True)
    assert np.array_equal(packed, np.packbits(fx["predictions"], axis=1, bitorder="little"))
    one = dec.decode(fx["shots"][5].copy())
    assert np.array_equal(np.asarray(one).astype(bool), fx["predictions"][5])


@pytest.mark.parametrize("decodings,window,commit", [(3, 2, 2), (5, 2, 1), (2, 5, 1)])
def test_fresh_shots_against_the_checker(decodings, window, commit, oracle_built):
    from oracle.window_oracle import WindowOracle
    rounds = (window - commit) + decodings * commit
    h = ring_code(7)
    p = np.linspace(0.02, 0.07, 7)
    text = phenomenological_dem(h, rounds, p, 0.04, logical=(1, 4))
    check, obs, pri = phenomenological_matrices(h, rounds, p, 0.04, logical=(1, 4))
    shots, _ = sample_shots(check, np.minimum(2 * pri, 0.5), 200, seed=decodings * 10 + window)
    cfg = dict(max_iter=6, bp_method="minimum_sum", ms_scaling_factor=0.8, osd_method="osd_e", osd_order=3)
    want_p, want_c, want_s = WindowOracle(check, obs, pri, decodings=decodings, window=window, commit=commit, num_checks=7, **cfg).decode_batch(shots)
    fx = dict(text=text, decodings=decodings, window=window, commit=commit, num_checks=7, cfg=cfg)
    dec = _decoder(fx)
    got_s = shots.copy()
    assert np.array_equal(dec._corr_multiple_rounds_batch(got_s), want_c)
    assert np.array_equal(got_s, want_s)
    assert np.array_equal(dec.decode_batch(shots.copy()), want_p)


def test_rounds_must_divide_the_detectors():
    from ldpc_amd.ckt_noise import BpOsdOverlappingWindowDecoder
    text = phenomenological_dem(ring_code(5), 6, 0.01, 0.01)
    with pytest.raises(ValueError, match="multiple of the number of rounds"):
        BpOsdOverlappingWindowDecoder(text, decodings=2, window=4, commit=3, num_checks=5)  # 7 rounds, 30 detectors


def test_sinter_adaptors_through_files(tmp_path):
    from ldpc_amd.ckt_noise import SinterDecoder_BPOSD_OWD
    from ldpc_amd.sinter_decoders import SinterBpOsdDecoder
    fx = _window_fixture(window_fixtures()[0])
    (tmp_path / "model.dem").write_text(fx["text"])
    np.packbits(fx["shots"], axis=1, bitorder="little").tofile(tmp_path / "dets.b8")
    kw = dict(num_shots=len(fx["shots"]), num_dets=fx["shots"].shape[1], num_obs=fx["predictions"].shape[1], dem_path=tmp_path / "model.dem",
              dets_b8_in_path=tmp_path / "dets.b8", tmp_dir=tmp_path)
    SinterDecoder_BPOSD_OWD(decodings=fx["decodings"], window=fx["window"], commit=fx["commit"], num_checks=fx["num_checks"],
                            decoder_config=fx["cfg"]).decode_via_files(obs_predictions_b8_out_path=tmp_path / "owd.b8", **kw)
    got = np.fromfile(tmp_path / "owd.b8", np.uint8).reshape(len(fx["shots"]), -1)
    assert np.array_equal(got, np.packbits(fx["predictions"], axis=1, bitorder="little"))
    compiled = SinterDecoder_BPOSD_OWD(decodings=fx["decodings"], window=fx["window"], commit=fx["commit"], num_checks=fx["num_checks"],
                                       decoder_config=fx["cfg"]).compile_decoder_for_dem(dem=fx["text"])
    again = compiled.decode_shots_bit_packed(bit_packed_detection_event_data=np.packbits(fx["shots"], axis=1, bitorder="little"))
    assert np.array_equal(again, got)
    # the plain (single-window) sinter decoder now reads the .dem file itself: same answer as one window over everything
    single = _window_fixture([p for p in window_fixtures() if "single_window" in p][0])
    (tmp_path / "single.dem").write_text(single["text"])
    np.packbits(single["shots"], axis=1, bitorder="little").tofile(tmp_path / "single.b8")
    SinterBpOsdDecoder(max_iter=single["cfg"]["max_iter"], bp_method="minimum_sum", ms_scaling_factor=1.0).decode_via_files(
        num_shots=len(single["shots"]), num_dets=single["shots"].shape[1], num_obs=1, dem_path=tmp_path / "single.dem",
        dets_b8_in_path=tmp_path / "single.b8", obs_predictions_b8_out_path=tmp_path / "single_out.b8", tmp_dir=tmp_path)
    got = np.fromfile(tmp_path / "single_out.b8", np.uint8).reshape(len(single["shots"]), -1)
    assert np.array_equal(got, np.packbits(single["predictions"], axis=1, bitorder="little"))
    # ... and keeps the configured decoder for single shots (reference :128-130); a serial schedule takes the unpacked route
    plain = SinterBpOsdDecoder(max_iter=single["cfg"]["max_iter"], bp_method="minimum_sum", ms_scaling_factor=1.0)
    kw1 = dict(num_shots=len(single["shots"]), num_dets=single["shots"].shape[1], num_obs=1, dem_path=tmp_path / "single.dem",
               dets_b8_in_path=tmp_path / "single.b8", tmp_dir=tmp_path)
    plain.decode_via_files(obs_predictions_b8_out_path=tmp_path / "p.b8", **kw1)
    for b in (0, 3, 17):
        assert np.array_equal(np.asarray(plain.decode(single["shots"][b].copy())).astype(bool), single["predictions"][b])
    serial = SinterBpOsdDecoder(max_iter=single["cfg"]["max_iter"], bp_method="minimum_sum", ms_scaling_factor=1.0, schedule="serial")
    serial.decode_via_files(obs_predictions_b8_out_path=tmp_path / "s.b8", **kw1)
    out = np.unpackbits(np.fromfile(tmp_path / "s.b8", np.uint8).reshape(len(single["shots"]), -1), axis=1, bitorder="little", count=1)
    for b in (0, 3, 17, 40):
        assert np.array_equal(out[b], np.asarray(serial.decode(single["shots"][b].copy())).astype(np.uint8))


def test_window_columns_are_compressed_and_priors_above_one_half_survive(oracle_built):
    """OSD-0 windows run on the window's own columns; an untouched column with prior >= 0.5 is 1 exactly where BP converged."""
    from oracle.window_oracle import WindowOracle
    from ldpc_amd.ckt_noise import BpOsdOverlappingWindowDecoder
    h = ring_code(6)
    p = np.array([0.03, 0.6, 0.03, 0.03, 0.55, 0.03])  # two bits more likely flipped than not, in every round
    text = phenomenological_dem(h, 6, p, 0.03)
    check, obs, pri = phenomenological_matrices(h, 6, p, 0.03)
    shots, _ = sample_shots(check, np.minimum(pri, 0.2), 150, seed=9)
    cfg = dict(max_iter=8, bp_method="minimum_sum", ms_scaling_factor=0.9)
    want_p, want_c, want_s = WindowOracle(check, obs, pri, decodings=2, window=4, commit=2, num_checks=6, **cfg).decode_batch(shots)
    dec = BpOsdOverlappingWindowDecoder(text, decodings=2, window=4, commit=2, num_checks=6, decoder_config=cfg)
    got_s = shots.copy()
    got_c = dec._corr_multiple_rounds_batch(got_s)
    assert len(dec._decoders[0].cols) < check.shape[1] and len(dec._decoders[0].static_ones) > 0
    assert np.array_equal(got_c, want_c) and np.array_equal(got_s, want_s)
    assert np.array_equal(dec.decode_batch(shots.copy()), want_p)


def test_bench_model_against_the_checker(oracle_built, capsys):
    """The model tools/bench_window.py times (BB144 x 12 rounds, 3 windows of 6 committing 3): 100 shots against the
    shot-by-shot checker, whose pace is printed (pytest -s) -- the CPU figure quoted next to the device's in DESIGN.md."""
    import time
    from oracle.window_oracle import WindowOracle
    from ldpc_amd import codes
    from ldpc_amd.ckt_noise import BpOsdOverlappingWindowDecoder
    h = codes.bivariate_bicycle_hx()
    text = phenomenological_dem(h, 12, 0.003, 0.003, logical=tuple(range(12)))
    check, obs, pri = phenomenological_matrices(h, 12, 0.003, 0.003, logical=tuple(range(12)))
    shots, _ = sample_shots(check, pri, 100, seed=11)
    cfg = dict(max_iter=30, bp_method="minimum_sum", ms_scaling_factor=0.625)
    w = WindowOracle(check, obs, pri, decodings=3, window=6, commit=3, num_checks=h.shape[0], **cfg)
    t0 = time.perf_counter()
    want = w.decode_batch(shots)[0]
    pace = len(shots) / (time.perf_counter() - t0)
    dec = BpOsdOverlappingWindowDecoder(text, decodings=3, window=6, commit=3, num_checks=h.shape[0], decoder_config=cfg)
    assert np.array_equal(dec.decode_batch(shots.copy()), want)
    packed = dec.decode_batch(np.packbits(shots, axis=1, bitorder="little"), bit_packed_shots=True, bit_packed_predictions=True)
    assert np.array_equal(packed, np.packbits(want, axis=1, bitorder="little"))
    with capsys.disabled():
        print(f"\n[window checker, inner decodes by {w.inner}: {pace:.0f} shots/s on one core]")

