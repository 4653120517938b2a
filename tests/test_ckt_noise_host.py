"""Host side of ldpc_amd.ckt_noise: the DEM text reader, DEM -> matrices, window indices, and the window checker."""
import numpy as np
import pytest
import scipy.sparse as sp

from ldpc_amd.ckt_noise.dem_text import parse_dem_text
from ldpc_amd.ckt_noise.dem_matrices import detector_error_model_to_check_matrices
from ldpc_amd.ckt_noise.base_overlapping_window_decoder import current_round_inds
from window_util import phenomenological_dem, phenomenological_matrices, ring_code


def test_reader_flattens_repeat_blocks_and_shifts():
    text = """
        error(0.1) D0 D1 L0   # a comment
        repeat 2 {
            error[tagged](0.2) D0 ^ D1 D2 L1
            shift_detectors(1.5) 3
            repeat 2 { error(0.05) D0
                       shift_detectors 1 }
        }
        detector(1, 2) D4
        logical_observable L3
        ERROR(0.3) D1
    """
    dem = parse_dem_text(text)
    got = [(e.probability, e.detectors, e.observables) for e in dem.errors]
    assert got == [
        (0.1, [[0, 1]], [[0]]),
        (0.2, [[0], [1, 2]], [[], [1]]),
        (0.05, [[3]], [[]]), (0.05, [[4]], [[]]),
        (0.2, [[5], [6, 7]], [[], [1]]),
        (0.05, [[8]], [[]]), (0.05, [[9]], [[]]),
        (0.3, [[11]], [[]]),
    ]
    assert dem.num_detectors == 15  # detector D4 after a shift of 10
    assert dem.num_observables == 4


def test_matrices_merge_repeated_detector_sets_and_keep_the_last_observables():
    text = "error(0.1) D0 D1 L0\nerror(0.2) D2\nerror(0.25) D1 D0\nerror(0.5) D0 ^ D0 D3\n"
    mats = detector_error_model_to_check_matrices(text)
    assert mats.check_matrix.shape == (4, 3)
    assert np.array_equal(mats.check_matrix.toarray(), np.array([[1, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.uint8))
    # {D0, D1} seen twice: 0.1 then 0.25 -> 0.1 * 0.75 + 0.25 * 0.9; its observables are those of the LAST occurrence (none)
    assert np.allclose(mats.priors, [0.1 * 0.75 + 0.25 * 0.9, 0.2, 0.5])
    assert mats.observables_matrix.shape == (1, 3) and mats.observables_matrix.nnz == 0
    # edges: {D0,D1}, {D2}, {D0}, {D0,D3}; the third hyperedge {D3} decomposes into the edges {D0} and {D0, D3}
    assert mats.edge_check_matrix.shape == (4, 4)
    assert np.array_equal(mats.hyperedge_to_edge_matrix.toarray()[:, 2], [0, 0, 1, 1])


def test_undecomposed_hyperedge_raises_unless_allowed():
    text = "error(0.1) D0 D1 D2\n"
    with pytest.raises(ValueError, match="not decomposed"):
        detector_error_model_to_check_matrices(text)
    mats = detector_error_model_to_check_matrices(text, allow_undecomposed_hyperedges=True)
    assert mats.check_matrix.shape == (3, 1) and mats.edge_check_matrix.shape == (3, 0)


@pytest.mark.parametrize("rounds", [2, 5])
def test_phenomenological_text_gives_the_directly_built_matrices(rounds, tmp_path):
    h = ring_code(6)
    p = np.linspace(0.01, 0.06, 6)
    text = phenomenological_dem(h, rounds, p, 0.02, logical=(0, 3))
    check, obs, pri = phenomenological_matrices(h, rounds, p, 0.02, logical=(0, 3))
    path = tmp_path / "model.dem"
    path.write_text(text)
    for source in (text, path, str(path)):
        mats = detector_error_model_to_check_matrices(source, allow_undecomposed_hyperedges=True)
        assert (mats.check_matrix != check).nnz == 0
        assert (mats.observables_matrix != obs).nnz == 0
        assert np.array_equal(mats.priors, pri)


def test_window_indices():
    h = ring_code(5)
    check, _, _ = phenomenological_matrices(h, 6, 0.01, 0.01)
    dcm = sp.csr_matrix(check)
    # round t: columns [10 t, 10 t + 5) data, [10 t + 5, 10 t + 10) measurement (none in the last round)
    c, d, sc, sd = current_round_inds(dcm, decoding=0, window=3, commit=2, num_checks=5)
    assert (sc, sd) == (slice(0, 10), slice(0, 15))
    assert (c.start, c.stop, d.start, d.stop) == (0, 20, 0, 30)
    c, d, sc, sd = current_round_inds(dcm, decoding=1, window=3, commit=2, num_checks=5)
    assert (sc, sd) == (slice(10, 20), slice(10, 25))
    assert (c.start, c.stop, d.start, d.stop) == (15, 40, 15, 50)  # round 2's detectors also see round 1's measurement errors


def test_window_checker_with_one_window_is_plain_bposd(oracle_built):
    from oracle.window_oracle import WindowOracle
    h = ring_code(6)
    check, obs, pri = phenomenological_matrices(h, 3, 0.05, 0.03)
    rng = np.random.default_rng(3)
    e = (rng.random((20, check.shape[1])) < 0.08).astype(np.uint8)
    shots = np.ascontiguousarray((sp.csr_matrix(check) @ e.T % 2).T.astype(np.uint8))
    w = WindowOracle(check, obs, pri, decodings=1, window=3, commit=3, num_checks=6, max_iter=10, inner="oracle")
    preds, corrs, _ = w.decode_batch(shots)
    plain = oracle_built.BpOracle(sp.csr_matrix(check), error_channel=pri, max_iter=10, bp_method="minimum_sum", ms_scaling_factor=1.0)
    want = plain.bposd_decode_batch(shots, 1, 0, want_llr=False)[0]
    want[~shots.any(axis=1)] = 0
    assert np.array_equal(corrs, want)
    assert np.array_equal(preds[:, 0], (corrs @ obs.toarray()[0]) % 2 == 1)


def _window_fixture(path):
    z = np.load(path, allow_pickle=False)
    cfg = {}
    for k, v in zip(z["config_keys"], z["config_vals"]):
        cfg[str(k)] = str(v) if str(k) in ("bp_method", "osd_method") else (float(v) if "." in str(v) else int(v))
    nd = int(z["num_detectors"])
    unpack = lambda a: np.ascontiguousarray(np.unpackbits(a, axis=1, bitorder="little")[:, :nd])  # noqa: E731
    return dict(text=str(z["dem_text"]), num_checks=int(z["num_checks"]), decodings=int(z["decodings"]), window=int(z["window"]),
                commit=int(z["commit"]), cfg=cfg, shots=unpack(z["shots"]), predictions=z["predictions"], corrections=z["corrections"],
                shots_after=unpack(z["shots_after"]), priors_after=z["priors_after"])


def window_fixtures():
    import glob
    import os
    return sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "window_*.npz")))


@pytest.mark.parametrize("path", window_fixtures(), ids=lambda p: p.split("/")[-1][:-4])
def test_window_checker_with_the_c_oracle_inside_reproduces_the_fixtures(path, oracle_built):
    """The fixtures were made with the real reference decoding each window; the C restatement must give the same."""
    from oracle.window_oracle import WindowOracle
    fx = _window_fixture(path)
    mats = detector_error_model_to_check_matrices(fx["text"], allow_undecomposed_hyperedges=True)
    w = WindowOracle(mats.check_matrix, mats.observables_matrix, mats.priors, decodings=fx["decodings"], window=fx["window"],
                     commit=fx["commit"], num_checks=fx["num_checks"], inner="oracle", **fx["cfg"])
    preds, corrs, after = w.decode_batch(fx["shots"])
    assert np.array_equal(corrs, fx["corrections"])
    assert np.array_equal(preds, fx["predictions"])
    assert np.array_equal(after, fx["shots_after"])
    assert np.array_equal(w.weights, fx["priors_after"])


def _random_dem_ast(rng, depth=0):
    """A random program: list of ("error", p, groups) / ("shift", k) / ("detector", d) / ("obs", l) / ("repeat", count, body)."""
    prog = []
    for _ in range(int(rng.integers(1, 5))):
        kind = rng.choice(["error", "error", "shift", "detector", "obs", "repeat"] if depth < 2 else ["error", "shift", "detector"])
        if kind == "error":
            groups = []
            for _ in range(int(rng.integers(1, 4))):
                dets = sorted(set(int(x) for x in rng.integers(0, 6, size=int(rng.integers(0, 4)))))
                obs = sorted(set(int(x) for x in rng.integers(0, 3, size=int(rng.integers(0, 2)))))
                groups.append((dets, obs))
            prog.append(("error", round(float(rng.uniform(0.001, 0.4)), 6), groups))
        elif kind == "shift":
            prog.append(("shift", int(rng.integers(0, 5))))
        elif kind == "detector":
            prog.append(("detector", int(rng.integers(0, 8))))
        elif kind == "obs":
            prog.append(("obs", int(rng.integers(0, 4))))
        else:
            prog.append(("repeat", int(rng.integers(0, 4)), _random_dem_ast(rng, depth + 1)))
    return prog


def _dem_text(prog, indent=""):
    out = []
    for ins in prog:
        if ins[0] == "error":
            tg = " ^ ".join(" ".join([f"D{d}" for d in dets] + [f"L{o}" for o in obs]) for dets, obs in ins[2])
            out.append(f"{indent}error({ins[1]}) {tg}".rstrip())
        elif ins[0] == "shift":
            out.append(f"{indent}shift_detectors(0.5, 1) {ins[1]}")
        elif ins[0] == "detector":
            out.append(f"{indent}detector(1, 2, 3) D{ins[1]}  # declared")
        elif ins[0] == "obs":
            out.append(f"{indent}logical_observable L{ins[1]}")
        else:
            out.append(f"{indent}repeat {ins[1]} {{")
            out.extend(_dem_text(ins[2], indent + "    "))
            out.append(f"{indent}}}")
    return out


def _dem_flatten(prog, state):
    for ins in prog:
        if ins[0] == "error":
            dets = [[d + state["shift"] for d in g[0]] for g in ins[2]]
            obs = [list(g[1]) for g in ins[2]]
            state["errors"].append((ins[1], dets, obs))
            for g in dets:
                for d in g:
                    state["nd"] = max(state["nd"], d + 1)
            for g in obs:
                for o in g:
                    state["no"] = max(state["no"], o + 1)
        elif ins[0] == "shift":
            state["shift"] += ins[1]
        elif ins[0] == "detector":
            state["nd"] = max(state["nd"], ins[1] + state["shift"] + 1)
        elif ins[0] == "obs":
            state["no"] = max(state["no"], ins[1] + 1)
        else:
            for _ in range(ins[1]):
                _dem_flatten(ins[2], state)


@pytest.mark.parametrize("seed", range(25))
def test_reader_against_an_independent_flattening_of_random_programs(seed):
    rng = np.random.default_rng(seed)
    prog = _random_dem_ast(rng)
    text = "\n".join(_dem_text(prog)) + "\n"
    want = dict(shift=0, errors=[], nd=0, no=0)
    _dem_flatten(prog, want)
    got = parse_dem_text(text)
    assert [(e.probability, e.detectors, e.observables) for e in got.errors] == want["errors"], text
    assert (got.num_detectors, got.num_observables) == (want["nd"], want["no"]), text


def test_window_decoder_restricts_a_serial_order_to_the_occupied_columns():
    """A window matrix with empty columns + schedule='serial' with an explicit order: the order handed to the inner decoder
    is a list (the decoder's own type check, pyx:620-623) of the occupied columns in their old relative order."""
    import scipy.sparse as sp
    from ldpc_amd.ckt_noise.bposd_overlapping_window import _WindowBpOsd
    h = sp.csr_matrix(np.array([[1, 0, 1, 0, 0, 1], [0, 0, 1, 0, 1, 1]], dtype=np.uint8))  # columns 1 and 3 are empty
    w = np.array([0.1, 0.6, 0.1, 0.2, 0.1, 0.1])
    dec = _WindowBpOsd(h, w, dict(schedule="serial", serial_schedule_order=[5, 4, 3, 2, 1, 0], osd_method="osd_0", osd_order=0,
                                 bp_method="minimum_sum", max_iter=0))
    assert list(dec.cols) == [0, 2, 4, 5] and list(dec.static_ones) == [1]
    order = dec.inner.serial_schedule_order
    assert [int(v) for v in order] == [3, 2, 1, 0]  # old columns 5, 4, 2, 0
    assert dec.inner.max_iter == 6  # max_iter = 0 means the FULL width's n
    with pytest.raises(ValueError):
        _WindowBpOsd(h, w, dict(schedule="serial", serial_schedule_order=[0, 1, 2], osd_method="osd_0", osd_order=0))
