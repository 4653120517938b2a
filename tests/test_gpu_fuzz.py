"""Randomised differential test of the device path against the CPU oracle (which is itself pinned to the real reference):
random sparse matrices (irregular degrees, empty rows and columns, rows heavier than every register bound), random
priors including 0.5 and above, every kernel family (streaming, per-pass, on-chip, serial, OSD, soft syndromes), ragged
batches.  Seeds are fixed: a failure is reproducible."""
import numpy as np
import pytest
import scipy.sparse as sp

from golden_util import bits_equal

pytestmark = pytest.mark.gpu


def _random_code(rng, kind):
    if kind == "tiny":
        m, n, dens = int(rng.integers(1, 9)), int(rng.integers(2, 13)), rng.uniform(0.15, 0.7)
    elif kind == "wide_rows":  # rows far heavier than the 16-entry register bound
        m, n, dens = int(rng.integers(3, 8)), int(rng.integers(40, 90)), rng.uniform(0.3, 0.6)
    elif kind == "tall_cols":
        m, n, dens = int(rng.integers(30, 70)), int(rng.integers(4, 12)), rng.uniform(0.3, 0.7)
    else:
        m, n, dens = int(rng.integers(10, 60)), int(rng.integers(20, 130)), rng.uniform(0.03, 0.15)
    h = (rng.random((m, n)) < dens).astype(np.uint8)
    if kind == "sparse" and m > 3:
        h[int(rng.integers(m))] = 0  # an empty check
        h[:, int(rng.integers(n))] = 0  # an isolated bit
    return sp.csr_matrix(h)


def _priors(rng, n, style):
    if style == 0:
        return np.full(n, float(rng.uniform(0.01, 0.2)))
    p = rng.uniform(0.005, 0.3, size=n)
    if style == 2:
        p[rng.integers(n)] = 0.5
        p[rng.integers(n)] = 0.8
    return p


def _syndromes(rng, h, batch, raw_bytes):
    m, n = h.shape
    e = (rng.random((batch, n)) < 0.1).astype(np.uint8)
    s = np.ascontiguousarray((h @ e.T % 2).T.astype(np.uint8)).reshape(batch, m)
    if raw_bytes and m:
        s[rng.integers(batch), rng.integers(m)] = 3  # a byte > 1: never converges, sign / parity semantics differ
    return s


CASES = [(seed, kind) for seed in range(6) for kind in ("tiny", "wide_rows", "tall_cols", "sparse")]


@pytest.mark.parametrize("seed,kind", CASES)
def test_parallel_schedule_random_codes(seed, kind, oracle_built):
    from ldpc_amd.engine import HipBpEngine
    rng = np.random.default_rng(1000 + seed * 17 + ("tiny", "wide_rows", "tall_cols", "sparse").index(kind))
    h = _random_code(rng, kind)
    m, n = h.shape
    for method, alpha in (("product_sum", 1.0), ("minimum_sum", float(rng.choice([0.0, 0.625, 1.0])))):
        probs = _priors(rng, n, int(rng.integers(3)))
        max_iter = int(rng.integers(1, 12))
        batch = int(rng.choice([1, 63, 64, 65, 130]))
        s = _syndromes(rng, h, batch, raw_bytes=seed % 2 == 0)
        o = oracle_built.BpOracle(h, error_channel=probs, max_iter=max_iter, bp_method=method, ms_scaling_factor=alpha)
        want = o.decode_batch(s)
        eng = HipBpEngine(h.indptr, h.indices, n, probs, max_iter, 0 if method == "product_sum" else 1, alpha)
        for small, handoff in ((-1, -1), (0, 0), (0, 100000), (1, -1), (2, -1), (3, -1), (4, -1), (5, -1)):
            eng.set_small_code_kernel(small)
            eng.set_handoff(handoff)
            got = eng.decode_batch(s)
            tag = f"{kind} seed {seed} {method} small={small} handoff={handoff} m={m} n={n} B={batch} it={max_iter}"
            assert np.array_equal(got[0], want[0]), "decoding: " + tag
            assert np.array_equal(got[3], want[3]) and np.array_equal(got[2], want[2]), "converge / iterations: " + tag
            assert bits_equal(got[1], want[1]), "log-ratios: " + tag


@pytest.mark.parametrize("seed", range(6))
def test_serial_osd_and_soft_random_codes(seed, oracle_built):
    from ldpc_amd.engine import HipBpEngine
    rng = np.random.default_rng(77 + seed)
    h = _random_code(rng, "sparse" if seed % 2 else "tiny")
    m, n = h.shape
    probs = _priors(rng, n, 1)
    max_iter = int(rng.integers(1, 8))
    batch = int(rng.choice([5, 64, 97]))
    s = _syndromes(rng, h, batch, raw_bytes=False)
    for method, alpha in (("product_sum", 1.0), ("minimum_sum", 0.75)):
        o = oracle_built.BpOracle(h, error_channel=probs, max_iter=max_iter, bp_method=method, ms_scaling_factor=alpha)
        eng = HipBpEngine(h.indptr, h.indices, n, probs, max_iter, 0 if method == "product_sum" else 1, alpha)
        # serial schedule with a random order
        order = rng.permutation(n).astype(np.int32)
        want = o.decode_serial_batch(s, order)
        eng.set_schedule("serial", order)
        for serial_kernel in (-1, 0, 1):
            eng.set_serial_kernel(serial_kernel)
            got = eng.decode_batch(s)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3]), serial_kernel
            assert bits_equal(got[1], want[1]), serial_kernel
        # an order that is not a permutation (the reference accepts any n bit numbers)
        dup = rng.integers(0, n, size=n).astype(np.int32)
        want_d = o.decode_serial_batch(s, dup)
        eng.set_schedule("serial", dup)
        for serial_kernel in (0, 1):
            eng.set_serial_kernel(serial_kernel)
            got = eng.decode_batch(s)
            assert np.array_equal(got[0], want_d[0]) and np.array_equal(got[2], want_d[2]) and bits_equal(got[1], want_d[1]), serial_kernel
        eng.set_schedule("parallel")
        # BP + OSD of every kind (syndromes are H e, hence in the image of H)
        for osd_method, osd_order in ((1, 0), (3, int(rng.integers(1, 9))), (2, int(rng.integers(1, 7)))):
            want = o.bposd_decode_batch(s, osd_method, osd_order)
            eng.set_osd(osd_method, osd_order)
            for osd_kernel in (-1, 0, 2):  # 2: the OSD-0 path for matrices beyond LDS (higher orders ignore it)
                eng.set_osd_kernel(osd_kernel)
                got = eng.decode_batch(s, osd=True)
                assert np.array_equal(got[0], want[0]), f"OSD {osd_method}/{osd_order} kernel {osd_kernel} seed {seed} {method} m={m} n={n}"
                assert np.array_equal(got[3], want[3])
            eng.set_osd_kernel(-1)
    # soft syndromes (always serial minimum-sum)
    soft = rng.normal(scale=2.0, size=(batch, m))
    o = oracle_built.BpOracle(h, error_channel=probs, max_iter=max_iter, bp_method="minimum_sum", ms_scaling_factor=0.9)
    eng = HipBpEngine(h.indptr, h.indices, n, probs, max_iter, 1, 0.9)
    want = o.soft_info_decode_batch(soft, 3.0, 1.5)
    for serial_kernel in (0, 1):
        eng.set_serial_kernel(serial_kernel)
        got = eng.soft_info_decode_batch(soft, 3.0, 1.5)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3]), serial_kernel
        assert bits_equal(got[1], want[1]) and bits_equal(got[4], want[4]), serial_kernel


@pytest.mark.parametrize("seed", range(6))
def test_higher_order_osd_across_null_space_sizes(seed, oracle_built):
    """OSD_E / OSD_CS on matrices whose n - rank falls below 128, in (128, 256] and above 256: the three regimes the
    host picks a kernel by (register rows with a 2- or 4-word candidate mask, matrix in LDS)."""
    from ldpc_amd.engine import HipBpEngine
    rng = np.random.default_rng(4242 + seed)
    n = int([90, 200, 300, 420, 500, 511][seed])
    m = int([40, 70, 120, 200, 150, 256][seed])
    h = sp.csr_matrix((rng.random((m, n)) < 3.0 / m).astype(np.uint8))
    probs = rng.uniform(0.01, 0.1, size=n)
    s = _syndromes(rng, h, 70, raw_bytes=False)
    o = oracle_built.BpOracle(h, error_channel=probs, max_iter=3, bp_method="minimum_sum", ms_scaling_factor=0.625)
    eng = HipBpEngine(h.indptr, h.indices, n, probs, 3, 1, 0.625)
    for osd_method, osd_order in ((3, 7), (2, 5), (1, 0)):
        want = o.bposd_decode_batch(s, osd_method, osd_order)
        eng.set_osd(osd_method, osd_order)
        for osd_kernel in (-1, 0, 2):
            eng.set_osd_kernel(osd_kernel)
            got = eng.decode_batch(s, osd=True)
            assert np.array_equal(got[0], want[0]), f"OSD {osd_method}/{osd_order} kernel {osd_kernel} m={m} n={n}"
            assert np.array_equal(got[3], want[3])


@pytest.mark.parametrize("m,n,kernels", [(900, 1700, (-1,)), (400, 900, (-1, 0, 2))])
def test_osd_with_a_workgroup_per_syndrome(m, n, kernels, oracle_built):
    """900 x 1700: [H | s] is 190 KiB bit-packed, more than LDS holds -- OSD runs with H in an HBM scratch slot.
    400 x 900: it would fit LDS three times, i.e. three wavefronts per CU with the one-wavefront kernels (mode 0) -- the
    automatic choice is a workgroup per syndrome with H in LDS; mode 2 puts H in HBM.  All against the oracle."""
    from ldpc_amd._lib import LdpcHipError
    from ldpc_amd.engine import HipBpEngine
    rng = np.random.default_rng(99)
    rows = np.repeat(np.arange(m), 6)
    cols = rng.integers(0, n, size=m * 6)
    h = sp.csr_matrix((np.ones(m * 6, np.uint8), (rows, cols)), shape=(m, n))
    h.data[:] = 1
    h.sum_duplicates()
    h.data[:] = 1
    probs = rng.uniform(0.01, 0.08, size=n)
    e = (rng.random((40, n)) < 0.04).astype(np.uint8)
    s = np.ascontiguousarray((h @ e.T % 2).T.astype(np.uint8))
    o = oracle_built.BpOracle(h, error_channel=probs, max_iter=4, bp_method="minimum_sum", ms_scaling_factor=0.7)
    eng = HipBpEngine(h.indptr, h.indices, n, probs, 4, 1, 0.7)
    for osd_method, osd_order in ((1, 0), (3, 6), (2, 5)):
        want = o.bposd_decode_batch(s, osd_method, osd_order)
        assert not want[3].all(), "the case needs rows that go through OSD"
        eng.set_osd(osd_method, osd_order)
        for kernel in kernels:
            eng.set_osd_kernel(kernel)
            got = eng.decode_batch(s, osd=True)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[3], want[3]), (osd_method, osd_order, kernel)
            assert not np.any((h @ got[0].T % 2).T != s), "every OSD solution satisfies its syndrome"
    if m != 900:
        return
    # the column order and the candidate tables still live in LDS: a 10000 x 20000 matrix is refused, not mis-decoded
    big = sp.csr_matrix((np.ones(20000, np.uint8), (np.arange(20000) % 10000, np.arange(20000))), shape=(10000, 20000))
    eng = HipBpEngine(big.indptr, big.indices, 20000, np.full(20000, 0.05), 2, 1, 1.0)
    eng.set_osd(1, 0)
    s_big = np.zeros((2, 10000), np.uint8)
    s_big[:, 5] = 3  # a byte > 1 never converges, so both rows reach OSD
    with pytest.raises(LdpcHipError, match="150 KiB available"):
        eng.decode_batch(s_big, osd=True)


def test_matrix_without_entries(oracle_built):
    """An all-zero parity-check matrix (nnz = 0): every kernel family must launch (no zero-sized grid) and agree with the oracle."""
    from ldpc_amd.engine import HipBpEngine
    h = sp.csr_matrix(np.zeros((3, 5), np.uint8))
    s = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 1]], np.uint8)
    for method in ("product_sum", "minimum_sum"):
        o = oracle_built.BpOracle(h, error_rate=0.1, max_iter=4, bp_method=method, ms_scaling_factor=0.9)
        want = o.decode_batch(s)
        eng = HipBpEngine(h.indptr, h.indices, 5, np.full(5, 0.1), 4, 0 if method == "product_sum" else 1, 0.9)
        for small, handoff in ((-1, -1), (0, 0), (0, 100000)):
            eng.set_small_code_kernel(small)
            eng.set_handoff(handoff)
            got = eng.decode_batch(s)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3])
            assert bits_equal(got[1], want[1])
        eng.set_schedule("serial")
        ws = o.decode_serial_batch(s, None)
        for serial_kernel in (0, 1):
            eng.set_serial_kernel(serial_kernel)
            got = eng.decode_batch(s)
            assert np.array_equal(got[0], ws[0]) and np.array_equal(got[2], ws[2]) and np.array_equal(got[3], ws[3])
    eng = HipBpEngine(h.indptr, h.indices, 5, np.full(5, 0.1), 4, 1, 0.9)
    soft = np.array([[1.0, 2.0, 0.5], [-1.0, 2.0, 0.5]])
    want = oracle_built.BpOracle(h, error_rate=0.1, max_iter=4, bp_method="minimum_sum", ms_scaling_factor=0.9).soft_info_decode_batch(soft, 3.0, 2.0)
    got = eng.soft_info_decode_batch(soft, 3.0, 2.0)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3])


@pytest.mark.parametrize("seed", range(4))
def test_one_handle_through_a_random_sequence_of_settings(seed, oracle_built):
    """A handle is stateful (workspace reuse, cached tables, schedule levels, OSD settings): drive ONE handle through a random
    sequence of parameter changes and batch sizes and compare every decode with a fresh oracle for the current settings."""
    from ldpc_amd.engine import HipBpEngine
    from ldpc_amd import codes
    rng = np.random.default_rng(500 + seed)
    h = sp.csr_matrix([codes.bivariate_bicycle_hx(), codes.rotated_surface_code_x(5), codes.regular_ldpc_code(96, 3, 6, seed=3),
                       codes.hamming_code(4)][seed])
    m, n = h.shape
    st = dict(p=np.full(n, 0.06), max_iter=6, method=0, alpha=1.0, schedule="parallel", order=None)
    eng = HipBpEngine(h.indptr, h.indices, n, st["p"], st["max_iter"], st["method"], st["alpha"])
    for step in range(40):
        op = int(rng.integers(8))
        if op == 0:
            st["p"] = rng.uniform(0.01, 0.2, size=n) if rng.random() < 0.5 else np.full(n, float(rng.uniform(0.02, 0.12)))
            eng.set_channel(st["p"])
        elif op == 1:
            st["max_iter"], st["method"] = int(rng.integers(1, 10)), int(rng.integers(2))
            st["alpha"] = 1.0 if st["method"] == 0 else float(rng.choice([0.0, 0.625, 0.9]))
            eng.set_params(st["max_iter"], st["method"], st["alpha"])
        elif op == 2:
            if rng.random() < 0.5:
                st["schedule"], st["order"] = "parallel", None
                eng.set_schedule("parallel")
            else:
                st["schedule"] = "serial"
                st["order"] = rng.permutation(n).astype(np.int32) if rng.random() < 0.6 else None
                eng.set_schedule("serial", st["order"])
        elif op == 3:
            eng.set_small_code_kernel(int(rng.choice([-1, 0, 1, 2, 3, 4, 5])))
            eng.set_handoff(int(rng.choice([-1, 0, 2, 100000])))
        elif op == 4:
            eng.set_serial_kernel(int(rng.choice([-1, 0, 1])))
            eng.set_repack(int(rng.choice([-1, 0, 2])))
            eng.set_osd_kernel(int(rng.choice([-1, 0])))
        batch = int(rng.choice([1, 7, 64, 65, 200, 700]))
        e = (rng.random((batch, n)) < 0.07).astype(np.uint8)
        s = np.ascontiguousarray((h @ e.T % 2).T.astype(np.uint8))
        o = oracle_built.BpOracle(h, error_channel=st["p"], max_iter=st["max_iter"], bp_method=("product_sum", "minimum_sum")[st["method"]],
                                  ms_scaling_factor=st["alpha"])
        want_llr = bool(rng.random() < 0.7)
        osd = [None, (1, 0), (3, 4), (2, 3)][int(rng.integers(4))]
        if st["schedule"] == "serial":
            want = list(o.decode_serial_batch(s, st["order"]))
            if osd is not None:
                for b in np.flatnonzero(~want[3]):
                    want[0][b] = o.osdw(s[b], want[1][b], osd[0], osd[1], channel_probs=st["p"])[0]
        elif osd is not None:
            want = o.bposd_decode_batch(s, osd[0], osd[1])
        else:
            want = o.decode_batch(s)
        if osd is not None:
            eng.set_osd(*osd)
        got = eng.decode_batch(s, want_llr=want_llr, osd=osd is not None)
        tag = f"seed {seed} step {step} {st['schedule']} method {st['method']} it {st['max_iter']} B {batch} osd {osd}"
        assert np.array_equal(got[0], want[0]), "decoding: " + tag
        assert np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3]), "iterations / converge: " + tag
        if want_llr:
            assert bits_equal(got[1], want[1]), "log-ratios: " + tag
        else:
            assert got[1] is None


def test_streamed_parallel_schedule_repacks_from_the_previous_histogram(oracle_built):
    """A code kept off the on-chip kernels, a batch of 40 000 where most syndromes converge early: the first decode runs
    plain and leaves its iteration histogram, the second is steered by it (two passes), a fixed first-pass length forces
    two passes, and `set_repack(0)` switches it off -- all four give the same arrays, and those match the oracle."""
    import torch
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    h = sp.csr_matrix(codes.regular_ldpc_code(n=1200, dv=3, dc=6, seed=3))
    n = h.shape[1]
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, 0.03), 24, 1, 0.8)
    eng.set_small_code_kernel(0)
    # 95 % of the shots from mild noise (converge in a few iterations), 5 % from noise the code cannot handle, shuffled
    s = torch.cat([eng.gen_bsc_syndromes(5, 0.02, shot0=0, shots=38000, device="cuda:0"),
                   eng.gen_bsc_syndromes(6, 0.08, shot0=0, shots=2000, device="cuda:0")])
    s = s[torch.randperm(40000, generator=torch.Generator().manual_seed(1)).to(s.device)].contiguous()
    runs = {}
    for tag, repack in (("first", -1), ("steered", -1), ("fixed", 5), ("off", 0)):
        eng.set_repack(repack)
        out = eng.decode_batch(s)
        runs[tag] = [t.cpu().numpy() for t in out]
        runs[tag + "_ms"] = eng.last_kernel_ms()
    for tag in ("steered", "fixed", "off"):
        for a, b in zip(runs["first"], runs[tag]):
            assert bits_equal(a, b) if a.dtype == np.float64 else np.array_equal(a, b), tag
    # without log-ratios, in chunks of 200 tiles, and through the bit-packed entry point: the same decisions
    eng.set_repack(-1)
    lean = eng.decode_batch(s, want_llr=False)
    assert lean[1] is None and np.array_equal(lean[0].cpu().numpy(), runs["first"][0]) and np.array_equal(lean[2].cpu().numpy(), runs["first"][2])
    eng.set_tuning(max_chunk_tiles=200)
    chunked = eng.decode_batch(s)
    eng.set_tuning(max_chunk_tiles=0)
    assert np.array_equal(chunked[0].cpu().numpy(), runs["first"][0]) and bits_equal(chunked[1].cpu().numpy(), runs["first"][1])
    eng.set_observables(sp.identity(n, dtype=np.uint8, format="csr")[:16])
    obs, packed, it8, cv8 = eng.decode_b8(eng.pack_b8(s), want_decoding=True)
    assert np.array_equal(eng.unpack_b8(packed, n).cpu().numpy(), runs["first"][0]) and np.array_equal(it8.cpu().numpy(), runs["first"][2])
    conv = runs["first"][3].astype(bool)
    assert 0.8 < conv.mean() < 0.999 and runs["first"][2][conv].mean() < 8, "the case is meant to converge early, with stragglers"
    assert runs["steered_ms"] < 0.8 * runs["first_ms"], (runs["first_ms"], runs["steered_ms"], conv.mean())
    pick = np.r_[0:64, np.flatnonzero(~conv)[:64]]
    o = oracle_built.BpOracle(h, error_rate=0.03, max_iter=24, bp_method="minimum_sum", ms_scaling_factor=0.8)
    want = o.decode_batch(s.cpu().numpy()[pick])
    assert np.array_equal(runs["steered"][0][pick], want[0]) and np.array_equal(runs["steered"][2][pick], want[2])
    assert np.array_equal(runs["steered"][3][pick].astype(bool), want[3].astype(bool)) and bits_equal(runs["steered"][1][pick], want[1])
    # ordered-statistics post-processing on top of the repacked run (it consumes the scattered log-ratios and flags)
    eng.set_osd(1, 0)
    osd = eng.decode_batch(s, want_llr=False, osd=True)[0].cpu().numpy()
    want_osd = o.bposd_decode_batch(s.cpu().numpy()[pick], 1, 0, want_llr=False)[0]
    assert np.array_equal(osd[pick], want_osd)
    assert np.array_equal(osd[conv], runs["first"][0][conv])



@pytest.mark.parametrize("flood_lanes", [0, 1, 2, 4])
@pytest.mark.parametrize("method,alpha", [(0, 1.0), (1, 0.0)])
def test_lane_compaction_carries_the_decode_on_at_any_cut(method, alpha, flood_lanes, oracle_built):
    """The streamed two-pass decode (ldpc_hip_bp_set_repack(k)): k iterations for all, then the unconverged rows' message state is
    compacted lane by lane into dense tiles and iterations k + 1 ... follow on those.  Any cut must give the arrays of the plain
    run -- including a cut after which nothing is left, one that leaves almost everything, the last possible one, the adaptive
    scaling factor (which depends on the absolute iteration number) and a second pass small enough for the per-pass kernels."""
    import torch
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    h = sp.csr_matrix(codes.regular_ldpc_code(n=600, dv=3, dc=6, seed=9))
    n = h.shape[1]
    max_iter = 14
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, 0.04), max_iter, method, alpha)
    eng.set_small_code_kernel(0)
    # 1: what the first pass leaves finishes a workgroup per syndrome (bp_flood_lane_kernel) instead of in tiles; 2 / 4: after one / three rounds in compacted tiles
    eng.set_debug_switch("FLOOD_LANES", flood_lanes)
    s = torch.cat([eng.gen_bsc_syndromes(5, 0.03, shot0=0, shots=33000, device="cuda:0"),
                   eng.gen_bsc_syndromes(6, 0.09, shot0=0, shots=900, device="cuda:0")])
    s = s[torch.randperm(len(s), generator=torch.Generator().manual_seed(2)).to(s.device)].contiguous()
    eng.set_repack(0)
    plain = [t.cpu().numpy() for t in eng.decode_batch(s)]
    it = plain[2]
    assert 2 < np.median(it) < 9 and (it == max_iter).sum() > 100
    for k in (1, 2, int(np.median(it)), int(np.median(it)) + 2, max_iter - 1):
        eng.set_repack(k)
        got = [t.cpu().numpy() for t in eng.decode_batch(s)]
        for a, b in zip(plain, got):
            assert bits_equal(a, b) if a.dtype == np.float64 else np.array_equal(a, b), k
        lean = eng.decode_batch(s, want_llr=False)
        assert lean[1] is None and np.array_equal(lean[0].cpu().numpy(), plain[0]) and np.array_equal(lean[2].cpu().numpy(), plain[2]), k
    # caller-provided output arrays that hold garbage: the second pass writes its rows THROUGH the list into them (no scatter copies)
    eng.set_repack(4)
    out = tuple(torch.full_like(t, 77) if t.dtype != torch.float64 else torch.full_like(t, float("nan")) for t in eng.decode_batch(s))
    got = [t.cpu().numpy() for t in eng.decode_batch(s, out=out)]
    assert np.array_equal(got[0], plain[0]) and np.array_equal(got[2], plain[2]) and np.array_equal(got[3], plain[3]) and bits_equal(got[1], plain[1])
    rows = np.r_[0:48, np.flatnonzero(it == max_iter)[:48]]
    want = oracle_built.BpOracle(h, error_rate=0.04, max_iter=max_iter, bp_method=method, ms_scaling_factor=alpha).decode_batch(s.cpu().numpy()[rows])
    assert np.array_equal(plain[0][rows], want[0]) and np.array_equal(plain[2][rows], want[2]) and bits_equal(plain[1][rows], want[1])


@pytest.mark.parametrize("code", ["ldpc36_n1200", "irregular", "hamming_heavy"])
def test_rows_a_first_pass_leaves_finish_a_workgroup_per_syndrome(code, oracle_built):
    """bp_flood_lane_kernel on regular, irregular (rows and columns heavier than the register bounds included) and product-sum / min-sum
    decodes, forced after first passes of several lengths: the arrays of the plain run, bit for bit, and the checker's on a sample."""
    import torch
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    if code == "ldpc36_n1200":
        h, p, max_iter = sp.csr_matrix(codes.regular_ldpc_code(n=1200, dv=3, dc=6, seed=3)), 0.055, 20
    elif code == "irregular":
        h, p, max_iter = sp.csr_matrix(codes.irregular_ldpc_code(1500, 750, seed=4)), 0.03, 16
    else:
        h, p, max_iter = sp.csr_matrix(codes.hamming_code(7)), 0.01, 10  # rows of 64 entries: beyond the register bounds
    n = h.shape[1]
    for method, alpha in ((0, 1.0), (1, 0.75)):
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), max_iter, method, alpha)
        eng.set_small_code_kernel(0)
        s = eng.gen_bsc_syndromes(9, p, shot0=0, shots=33000 + 17, device="cuda:0")
        s[5, 0] = 2
        eng.set_repack(0)
        plain = [t.cpu().numpy() for t in eng.decode_batch(s)]
        for k, fl in ((1, 1), (3, 1), (max_iter - 1, 1), (2, 2), (3, 3), (1, 4)):
            eng.set_debug_switch("FLOOD_LANES", fl)
            eng.set_repack(k)
            got = [t.cpu().numpy() for t in eng.decode_batch(s)]
            for a, b in zip(plain, got):
                assert bits_equal(a, b) if a.dtype == np.float64 else np.array_equal(a, b), (code, method, k, fl)
            lean = eng.decode_batch(s, want_llr=False)
            assert lean[1] is None and np.array_equal(lean[0].cpu().numpy(), plain[0]) and np.array_equal(lean[2].cpu().numpy(), plain[2])
        rows = np.r_[0:40, np.flatnonzero(~plain[3].astype(bool))[:24]]
        want = oracle_built.BpOracle(h, error_rate=p, max_iter=max_iter, bp_method=method, ms_scaling_factor=alpha).decode_batch(s.cpu().numpy()[rows])
        assert np.array_equal(plain[0][rows], want[0]) and np.array_equal(plain[2][rows], want[2]) and bits_equal(plain[1][rows], want[1])
        eng.close()
