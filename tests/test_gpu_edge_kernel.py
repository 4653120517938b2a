"""bp_edge_kernel (min-sum, lane = edge, messages in registers; ldpc_amd/csrc/bp_edge_kernel.h) against the CPU oracle -- which is
pinned to the real reference -- and against the lane = node on-chip kernel, bit for bit: decisions, log-ratios (as bit
patterns), iteration counts, converge flags.  Codes of the family it serves (rows <= 4, columns 1 .. 2 entries): rotated surface
codes, ring / repetition codes, random matrices inside the bounds; both forms (uniform prior: no prior registers; per-column
priors), the corners (priors 0 / 1 -> infinite messages, syndrome bytes above 1, the adaptive scaling factor, rows of weight
1 .. 3, one iteration, no log-ratios, batches that are not a multiple of the pull chunk)."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _bits(x):
    return np.ascontiguousarray(x).view(np.int64) if x.dtype == np.float64 else x


def _run(h, probs, max_iter, alpha, synd, mode, want_llr=True):
    from ldpc_amd.engine import HipBpEngine
    h = sp.csr_matrix(h)
    eng = HipBpEngine(h.indptr, h.indices, h.shape[1], probs, max_iter, 1, alpha)
    eng.set_small_code_kernel(mode)
    out = eng.decode_batch(synd, want_llr=want_llr)
    eng.close()
    return out


def _check(h, probs, max_iter, alpha, synd, oracle, want_llr=True):
    h = sp.csr_matrix(h)
    got = _run(h, probs, max_iter, alpha, synd, 6, want_llr)
    node = _run(h, probs, max_iter, alpha, synd, 3, want_llr)
    want = oracle.BpOracle(h, error_channel=probs, max_iter=max_iter, bp_method="minimum_sum", ms_scaling_factor=alpha).decode_batch(synd)
    for k, name in ((0, "decoding"), (2, "iterations"), (3, "converge")):
        assert np.array_equal(got[k], node[k]), f"{name}: edge kernel vs node kernel"
        assert np.array_equal(got[k], want[k].astype(got[k].dtype)), f"{name}: edge kernel vs oracle"
    if want_llr:
        assert np.array_equal(_bits(got[1]), _bits(node[1])), "log-ratios: edge kernel vs node kernel (bit patterns)"
        assert np.array_equal(_bits(got[1]), _bits(want[1])), "log-ratios: edge kernel vs oracle (bit patterns)"
    else:
        assert got[1] is None
    return got


def _syndromes(h, p, batch, seed):
    rng = np.random.default_rng(seed)
    e = (rng.random((batch, h.shape[1])) < p).astype(np.uint8)
    return np.ascontiguousarray((sp.csr_matrix(h) @ e.T % 2).T.astype(np.uint8))


@pytest.mark.parametrize("d", [3, 5, 7, 11, 15, 21])
@pytest.mark.parametrize("uniform", [True, False])
def test_rotated_surface_codes(d, uniform, oracle_built):
    from ldpc_amd.codes import rotated_surface_code_x
    h = rotated_surface_code_x(d)
    n = h.shape[1]
    probs = np.full(n, 0.05) if uniform else np.random.default_rng(d).uniform(0.01, 0.12, size=n)
    synd = _syndromes(h, 0.05, 333, d)  # 333: not a multiple of any chunk
    synd[0] = 0
    got = _check(h, probs, 30, 0.625, synd, oracle_built)
    assert got[3].any() and (d < 11 or not got[3].all())


@pytest.mark.parametrize("alpha", [0.0, 1.0, 0.9])
def test_scaling_factor_incl_adaptive(alpha, oracle_built):
    from ldpc_amd.codes import rotated_surface_code_x
    h = rotated_surface_code_x(9)
    _check(h, np.full(h.shape[1], 0.08), 12, alpha, _syndromes(h, 0.08, 200, 1), oracle_built)


def test_ring_and_repetition_codes(oracle_built):
    from ldpc_amd.codes import rep_code, ring_code
    for h, p in ((ring_code(200), 0.1), (ring_code(7), 0.2), (rep_code(40), 0.1), (rep_code(2), 0.3)):
        _check(h, np.full(h.shape[1], p), 25, 0.8, _syndromes(h, p, 150, 3), oracle_built)


def _random_family_member(rng):
    """Random matrix with rows of weight 1 .. 4, columns of weight 1 .. 2, every column present."""
    n = int(rng.integers(4, 90))
    deg = rng.integers(1, 3, size=n)
    sockets = np.repeat(np.arange(n), deg)
    rng.shuffle(sockets)
    rows, cur, used = [], [], set()
    for j in sockets:
        if j in used or len(cur) == int(rng.integers(1, 5)):
            if cur:
                rows.append(cur)
            cur, used = [], set()
        cur.append(int(j))
        used.add(int(j))
        if len(cur) == 4:
            rows.append(cur)
            cur, used = [], set()
    if cur:
        rows.append(cur)
    h = np.zeros((len(rows), n), np.uint8)
    for i, r in enumerate(rows):
        h[i, r] = 1
    return sp.csr_matrix(h)


@pytest.mark.parametrize("seed", range(10))
def test_random_members_of_the_family(seed, oracle_built):
    rng = np.random.default_rng(500 + seed)
    h = _random_family_member(rng)
    assert h.sum(axis=1).max() <= 4 and 1 <= h.sum(axis=0).min() and h.sum(axis=0).max() <= 2
    n = h.shape[1]
    probs = np.full(n, float(rng.uniform(0.02, 0.2))) if seed % 2 else rng.uniform(0.01, 0.3, size=n)
    if seed % 3 == 0:
        probs[rng.integers(n)] = 0.5   # prior 0.0
        probs[rng.integers(n)] = 0.7   # negative prior
    synd = _syndromes(h, 0.1, 97, seed)
    if seed % 4 == 1:
        synd[5, rng.integers(h.shape[0])] = 3  # a byte above 1: parity from bit 0, never converges (bp.hpp:236, :300)
        synd[6, rng.integers(h.shape[0])] = 2
    _check(h, probs, int(rng.integers(1, 20)), float(rng.choice([0.0, 0.625, 1.0])), synd, oracle_built, want_llr=bool(seed % 5))


def test_infinite_priors(oracle_built):
    """p = 0 or 1 give priors of +-inf: magnitudes above DBL_MAX never enter the reference's minimum (bp.hpp:240-247), inf - inf = NaN."""
    from ldpc_amd.codes import rotated_surface_code_x, ring_code
    for h in (rotated_surface_code_x(5), ring_code(9)):
        n = h.shape[1]
        for where in ([0], [0, 1, 2, 3, 4], list(range(n))):
            for value in (0.0, 1.0):
                probs = np.full(n, 0.1)
                probs[where] = value
                with np.errstate(all="ignore"):
                    _check(h, probs, 6, 0.75, _syndromes(h, 0.2, 64, len(where)), oracle_built)
    probs = np.full(25, 0.1)
    probs[::2] = 0.0
    probs[1::4] = 1.0
    with np.errstate(all="ignore"):
        _check(rotated_surface_code_x(5), probs, 6, 1.0, _syndromes(rotated_surface_code_x(5), 0.3, 64, 9), oracle_built)


def test_one_iteration_and_small_batches(oracle_built):
    from ldpc_amd.codes import rotated_surface_code_x
    h = rotated_surface_code_x(13)
    for batch in (1, 2, 63, 65):
        _check(h, np.full(h.shape[1], 0.06), 1, 0.625, _syndromes(h, 0.06, batch, batch), oracle_built)
        _check(h, np.full(h.shape[1], 0.06), 40, 0.625, _syndromes(h, 0.06, batch, batch), oracle_built)


def test_large_batch_pulls_in_chunks(oracle_built):
    """A batch big enough for the chunked work counter (several syndromes per pull): the same rows in any split give the same bits."""
    from ldpc_amd.codes import rotated_surface_code_x
    from ldpc_amd.engine import HipBpEngine
    h = sp.csr_matrix(rotated_surface_code_x(7))
    n = h.shape[1]
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, 0.07), 30, 1, 0.625)
    eng.set_small_code_kernel(6)
    s = eng.gen_bsc_syndromes(3, 0.07, shot0=0, shots=700_003, device="cuda:0")
    big = eng.decode_batch(s)
    part = eng.decode_batch(s[123_456:123_456 + 999].contiguous())
    import torch
    for a, b in zip(big, part):
        assert bool(torch.equal(a[123_456:123_456 + 999].view(torch.int64) if a.dtype == torch.float64 else a[123_456:123_456 + 999],
                                b.view(torch.int64) if b.dtype == torch.float64 else b))
    rows = np.sort(np.random.default_rng(1).choice(700_003, size=300, replace=False))
    want = oracle_built.BpOracle(h, error_rate=0.07, max_iter=30, bp_method="minimum_sum", ms_scaling_factor=0.625).decode_batch(s[torch.from_numpy(rows).cuda()].cpu().numpy())
    assert np.array_equal(big[0].cpu().numpy()[rows], want[0]) and np.array_equal(big[2].cpu().numpy()[rows], want[2])
    assert np.array_equal(big[1].cpu().numpy()[rows].view(np.int64), want[1].view(np.int64))


def test_codes_outside_the_family_take_the_other_kernels(oracle_built):
    """Mode 6 = the edge kernel where it applies, else the automatic choice: a column of weight 3, an empty column."""
    h3 = np.array([[1, 1, 0, 0], [1, 0, 1, 0], [1, 0, 0, 1]], np.uint8)
    hempty = np.array([[1, 1, 0, 0], [0, 1, 1, 0]], np.uint8)
    for h in (h3, hempty):
        _check(h, np.full(4, 0.1), 5, 1.0, _syndromes(h, 0.2, 40, 1), oracle_built)


def test_debug_switches_live_in_the_handle(monkeypatch):
    """Environment variables seed the switches when the handle is created; afterwards only set_debug_switch changes them."""
    from ldpc_amd import _lib
    from ldpc_amd.codes import bivariate_bicycle_hx
    from ldpc_amd.engine import HipBpEngine
    h = sp.csr_matrix(bivariate_bicycle_hx())
    n = h.shape[1]
    monkeypatch.setenv("LDPC_HIP_PS_TEAM", "1")
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, 0.05), 20, 0, 1.0)
    monkeypatch.setenv("LDPC_HIP_PS_TEAM", "0")  # too late for this handle: not read again
    synd = _syndromes(h, 0.05, 300, 1)
    a = eng.decode_batch(synd)
    eng.set_debug_switch("PS_TEAM", 0)
    b = eng.decode_batch(synd)
    eng.set_debug_switch("PS_TEAM", -1)
    for x, y in zip(a, b):
        assert np.array_equal(_bits(x), _bits(y))
    with pytest.raises(_lib.LdpcHipError, match="unknown switch"):
        eng.set_debug_switch("NO_SUCH_SWITCH", 1)


# ---- bp_edge8_kernel: rows of weight <= 8 in eight neighbouring lanes, columns of weight <= 4 --------------------------------
def _random_wide_member(rng, max_col):
    """Random matrix with rows of weight 1 .. 8, columns of weight 1 .. max_col, every column present, m small enough for the kernel."""
    n = int(rng.integers(6, 120))
    deg = rng.integers(1, max_col + 1, size=n)
    sockets = np.repeat(np.arange(n), deg)
    rng.shuffle(sockets)
    rows, cur = [], []
    limit = int(rng.integers(1, 9))
    for j in sockets:
        if int(j) in cur or len(cur) == limit:
            if cur:
                rows.append(cur)
            cur, limit = [], int(rng.integers(1, 9))
        cur.append(int(j))
    if cur:
        rows.append(cur)
    h = np.zeros((len(rows), n), np.uint8)
    for i, r in enumerate(rows):
        h[i, r] = 1
    return sp.csr_matrix(h)


@pytest.mark.parametrize("seed", range(12))
def test_wide_kernel_random_members(seed, oracle_built):
    rng = np.random.default_rng(900 + seed)
    max_col = 3 if seed % 2 else 4
    for _ in range(50):
        h = _random_wide_member(rng, max_col)
        if h.shape[0] <= (96 if max_col == 3 else 72) and h.sum(axis=1).max() > 4:
            break
    assert h.sum(axis=1).max() <= 8 and 1 <= h.sum(axis=0).min() and h.sum(axis=0).max() <= max_col
    n = h.shape[1]
    probs = np.full(n, float(rng.uniform(0.02, 0.2))) if seed % 3 else rng.uniform(0.01, 0.3, size=n)
    if seed % 4 == 0:
        probs[rng.integers(n)] = 0.5
        probs[rng.integers(n)] = 0.7
    synd = _syndromes(h, 0.1, 131, seed)
    if seed % 4 == 1:
        synd[5, rng.integers(h.shape[0])] = 3
        synd[6, rng.integers(h.shape[0])] = 2
    _check(h, probs, int(rng.integers(1, 20)), float(rng.choice([0.0, 0.625, 1.0])), synd, oracle_built, want_llr=bool(seed % 5))


@pytest.mark.parametrize("uniform", [True, False])
def test_wide_kernel_bivariate_bicycle(uniform, oracle_built):
    from ldpc_amd.codes import bivariate_bicycle_hx
    h = bivariate_bicycle_hx()
    n = h.shape[1]
    probs = np.full(n, 0.05) if uniform else np.random.default_rng(3).uniform(0.01, 0.1, size=n)
    synd = _syndromes(h, 0.05, 777, 5)
    synd[0] = 0
    got = _check(h, probs, 50, 0.625, synd, oracle_built)
    assert got[3].any() and not got[3].all()
    _check(h, probs, 7, 0.0, synd[:100], oracle_built)


def test_wide_kernel_infinite_priors(oracle_built):
    from ldpc_amd.codes import bivariate_bicycle_hx
    h = bivariate_bicycle_hx()
    n = h.shape[1]
    for where, value in (([0], 0.0), (list(range(0, n, 3)), 1.0), (list(range(n)), 0.0)):
        probs = np.full(n, 0.1)
        probs[where] = value
        with np.errstate(all="ignore"):
            _check(h, probs, 6, 0.75, _syndromes(h, 0.15, 64, len(where)), oracle_built)
