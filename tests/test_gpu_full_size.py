"""BASELINE.json's configurations at their full sizes, checked through size-independent properties (the CPU oracle only
sees a sample): H x == s for exactly the rows flagged converged, iteration counts consistent with the flags, agreement
with the oracle on a random subset, invariance under the kernel family, and for config 5 that every OSD output solves
its syndrome.  Everything stays in HBM; only flags and small samples come back."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _check_flags_against_syndromes(eng, synd, dec, cv, it, max_iter):
    import torch
    resid = eng.mulvec_batch(dec) ^ synd          # H x + s on the device
    solved = ~resid.any(dim=1).bool()
    cvb = cv.bool()
    assert bool((solved == cvb).all()), "converge flag <=> H x == s must hold for every row (bp.hpp:300-308)"
    assert bool((it[~cvb] == max_iter).all()) and bool((it[cvb] >= 1).all()) and bool((it <= max_iter).all())
    return cvb


def _subset_vs_oracle(oracle_built, h, p, max_iter, method, alpha, synd, dec, it, cv, llr, rows):
    o = oracle_built.BpOracle(h, error_rate=p, max_iter=max_iter, bp_method=method, ms_scaling_factor=alpha)
    s = synd[rows].cpu().numpy()
    want = o.decode_batch(s)
    assert np.array_equal(dec[rows].cpu().numpy(), want[0])
    assert np.array_equal(it[rows].cpu().numpy(), want[2]) and np.array_equal(cv[rows].cpu().numpy().astype(bool), want[3])
    if llr is not None:
        from golden_util import bits_equal
        assert bits_equal(llr[rows].cpu().numpy(), want[1])


@pytest.mark.parametrize("p", [0.05, 0.09])
def test_config2_full_batch(p, oracle_built):
    """(3,6)-regular n = 10 000, product_sum, 50 iterations, B = 65 536 (bench.py's workload)."""
    import torch
    from ldpc_amd.codes import regular_ldpc_code
    from ldpc_amd.engine import HipBpEngine
    h = regular_ldpc_code(10000, 3, 6, seed=1)
    eng = HipBpEngine(h.indptr, h.indices, 10000, np.full(10000, p), 50, 0, 1.0)
    B = 65536
    synd, err = eng.gen_bsc_syndromes(7, p, shot0=0, shots=B, device=torch.device("cuda", 0), want_errors=True)
    dec, llr, it, cv = eng.decode_batch(synd, want_llr=True)
    cvb = _check_flags_against_syndromes(eng, synd, dec, cv, it, 50)
    if p == 0.05:  # below threshold: (almost) everything converges, and to the error that was injected
        assert cvb.float().mean().item() > 0.999
        assert ((dec == err).all(dim=1) == cvb).float().mean().item() > 0.999
        assert 6.0 < it.float().mean().item() < 8.0
    else:          # above threshold: (almost) nothing does
        assert cvb.float().mean().item() < 0.05
    rows = torch.from_numpy(np.random.default_rng(1).choice(B, 24 if p == 0.09 else 96, replace=False)).cuda()
    _subset_vs_oracle(oracle_built, h, p, 50, "product_sum", 1.0, synd, dec, it, cv, llr, rows)
    # the per-pass kernels alone give the same batch (first 16 384 rows = 256 tiles: per-pass from the first iteration)
    d2, _, i2, c2 = eng.decode_batch(synd[:16384].contiguous(), want_llr=False)
    assert bool((d2 == dec[:16384]).all()) and bool((i2 == it[:16384]).all()) and bool((c2 == cv[:16384]).all())


@pytest.mark.parametrize("p", [0.09, 0.05])
def test_config4_per_gpu_share(p, oracle_built):
    """BASELINE configs[3] at the size ONE GPU of its eight decodes: (3,6)-regular n = 10 000, product_sum, 50 iterations, B = 131 072
    (1 048 576 / 8; messages 63 GB).  flag <=> H x == s on every row, an oracle subset, and every row -- log-ratio bits included --
    equal to what two 65 536-row decodes of the same shot stream (shots [0, 65 536) and [65 536, 131 072), the shards two ranks of a
    16-GPU job would hold) give: a rank's results do not depend on how the stream is cut into shards."""
    import torch
    from ldpc_amd.codes import regular_ldpc_code
    from ldpc_amd.engine import HipBpEngine
    dev = torch.device("cuda", 0)
    h = regular_ldpc_code(10000, 3, 6, seed=1)
    eng = HipBpEngine(h.indptr, h.indices, 10000, np.full(10000, p), 50, 0, 1.0)
    B, half = 131072, 65536
    synd = eng.gen_bsc_syndromes(7, p, shot0=0, shots=B, device=dev)
    dec, llr, it, cv = eng.decode_batch(synd, want_llr=True)
    cvb = _check_flags_against_syndromes(eng, synd, dec, cv, it, 50)
    if p == 0.05:
        assert cvb.float().mean().item() > 0.999 and 6.0 < it.float().mean().item() < 8.0
    else:
        assert cvb.float().mean().item() < 0.05
    rows = torch.from_numpy(np.sort(np.random.default_rng(4).choice(B, 24 if p == 0.09 else 96, replace=False))).to(dev)
    _subset_vs_oracle(oracle_built, h, p, 50, "product_sum", 1.0, synd, dec, it, cv, llr, rows)
    for k in range(2):
        part = eng.gen_bsc_syndromes(7, p, shot0=k * half, shots=half, device=dev)
        assert bool(torch.equal(part, synd[k * half:(k + 1) * half])), "the shot stream is counter-based: a shard is a slice of it"
        d2, l2, i2, c2 = eng.decode_batch(part, want_llr=True)
        sl = slice(k * half, (k + 1) * half)
        assert bool(torch.equal(d2, dec[sl])) and bool(torch.equal(i2, it[sl])) and bool(torch.equal(c2, cv[sl]))
        same = (l2.view(torch.int64) == llr[sl].view(torch.int64)) | (l2.isnan() & llr[sl].isnan())  # (any NaN matches any NaN: oracle.bits_equal)
        assert bool(same.all()), "log-ratio bits differ between shard sizes"
        del part, d2, l2, i2, c2
    eng.close()


def test_bench_config4_geometry_under_a_process_group_of_one():
    """`python bench.py --gpus 1 --force-launch --batch-per-gpu 131072`: the line the driver's `--gpus 8` run prints, for the one rank
    this box can hold -- torch.distributed.run, an RCCL group, a 131 072-syndrome shard, the gather of bit-packed rows, per-rank parity."""
    from test_gpu_async_group import _bench
    got = _bench(["--gpus", "1", "--force-launch", "--batch-per-gpu", "131072", "--steps", "2", "--warmup", "1", "--rank-parity", "64",
                  "--secondary", "0", "--cpu-sample", "0", "--host-io", "0"], timeout=1200)
    assert got["n_gpus"] == 1 and got["rccl"]["ranks"] == 1 and got["rccl"]["backend"] == "nccl"
    assert got["config"]["batch_per_gpu"] == 131072 and got["config"]["global_batch"] == 131072
    assert got["gather"]["rows_on_rank0"] == 131072
    assert got["per_rank"]["parity_all_ranks"] is True and got["per_rank"]["parity_rows_per_rank"] == 64
    assert "parity_failed" not in got and got["value"] > 0
    assert 0.2 < got["roofline"]["frac"] <= 1.0
    assert abs(got["value"] - 131072 / (got["ms_per_step"] * 1e-3)) < 1e-6 * got["value"]


def test_config3_full_batch(oracle_built):
    """Rotated surface code d = 21, minimum_sum 0.625, 30 iterations, B = 262 144."""
    import torch
    from ldpc_amd.codes import rotated_surface_code_x
    from ldpc_amd.engine import HipBpEngine
    h = rotated_surface_code_x(21)
    m, n = h.shape
    B = 262144
    for p in (0.05, 0.01):
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), 30, 1, 0.625)
        synd = eng.gen_bsc_syndromes(7, p, shot0=0, shots=B, device=torch.device("cuda", 0))
        dec, llr, it, cv = eng.decode_batch(synd)
        _check_flags_against_syndromes(eng, synd, dec, cv, it, 30)
        rows = torch.from_numpy(np.random.default_rng(2).choice(B, 512, replace=False)).cuda()
        _subset_vs_oracle(oracle_built, h, p, 30, "minimum_sum", 0.625, synd, dec, it, cv, llr, rows)
        eng.set_small_code_kernel(0)  # the streaming kernel on the same batch
        d2, _, i2, c2 = eng.decode_batch(synd[:32768].contiguous(), want_llr=False)
        assert bool((d2 == dec[:32768]).all()) and bool((i2 == it[:32768]).all()) and bool((c2 == cv[:32768]).all())


def test_config5_full_batch_osd_orders(oracle_built):
    """BB [[144,12,12]], product_sum 50 iterations + OSD (0 / CS-10), B = 8 192: every output solves its syndrome and the
    sweep never raises the weight."""
    import torch
    from ldpc_amd.codes import bivariate_bicycle_hx
    from ldpc_amd.engine import HipBpEngine
    h = bivariate_bicycle_hx()
    eng = HipBpEngine(h.indptr, h.indices, 144, np.full(144, 0.05), 50, 0, 1.0)
    synd = eng.gen_bsc_syndromes(7, 0.05, shot0=0, shots=8192, device=torch.device("cuda", 0))
    eng.set_osd(1, 0)
    d0, _, it, cv = eng.decode_batch(synd, osd=True)
    eng.set_osd(3, 10)
    d1, _, it1, cv1 = eng.decode_batch(synd, osd=True)
    for d in (d0, d1):
        assert not bool((eng.mulvec_batch(d) ^ synd).any()), "every BP+OSD output must reproduce its syndrome"
    assert bool((it == it1).all()) and bool((cv == cv1).all())
    assert bool((d0[cv.bool()] == d1[cv.bool()]).all())  # converged rows are BP's own decision either way
    assert bool((d1.sum(dim=1) <= d0.sum(dim=1)).all())   # uniform priors: weight = Hamming weight (osd.hpp:171-177)
    rows = np.random.default_rng(3).choice(8192, 600, replace=False)
    want = oracle_built.BpOracle(h, error_rate=0.05, max_iter=50).bposd_decode_batch(synd.cpu().numpy()[rows], 3, 10)
    assert np.array_equal(d1.cpu().numpy()[rows], want[0])


def test_config2_code_with_osd0(oracle_built):
    """BP + OSD-0 on the headline code itself (5000 x 10000: [H | s] is 980 KiB per syndrome, the elimination runs with H
    in an HBM scratch slot): a few syndromes against the oracle, and every OSD solution satisfies its syndrome."""
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    h = sp.csr_matrix(codes.regular_ldpc_code(10000, 3, 6, seed=1))
    m, n = h.shape
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, 0.09), 5, 0, 1.0)
    s = eng.gen_bsc_syndromes(3, 0.09, shot0=0, shots=6, device="cuda:0").cpu().numpy()
    eng.set_osd(1, 0)
    dec, _, it, cv = eng.decode_batch(s, want_llr=False, osd=True)
    assert not cv.any(), "five iterations at p = 0.09 leave everything to OSD"
    assert not np.any((h @ dec.T % 2).T != s)
    o = oracle_built.BpOracle(h, error_rate=0.09, max_iter=5, bp_method="product_sum")
    want = o.bposd_decode_batch(s[:2], 1, 0, want_llr=False)
    assert np.array_equal(dec[:2], want[0])


@pytest.mark.parametrize("n, unblocked", [(2400, False), (3600, False), (2400, True)])
def test_workgroup_osd_beyond_1024_rows_higher_order(oracle_built, monkeypatch, n, unblocked):
    """OSD_CS / OSD_E / OSD-0 on 1200 x 2400 and 1800 x 3600 matrices: six and eight rows per thread in the blocked elimination
    (its limit is 2048 rows), and the one-pivot-per-step loop larger matrices take (forced here) -- both on the working copy with
    its columns in sorted order, the T planes squeezed out 1024 rows at a time."""
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    h = sp.csr_matrix(codes.regular_ldpc_code(n, 3, 6, seed=4))
    m = h.shape[0]
    assert m > 1024
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, 0.08), 4, 1, 0.75)
    if unblocked:
        eng.set_debug_switch("OSD_UNBLOCKED", 1)
    s = eng.gen_bsc_syndromes(5, 0.08, shot0=0, shots=40, device="cuda:0").cpu().numpy()
    o = oracle_built.BpOracle(h, error_rate=0.08, max_iter=4, bp_method="minimum_sum", ms_scaling_factor=0.75)
    for method, order in ((3, 7), (2, 5), (1, 0)):
        eng.set_osd(method, order)
        dec, _, it, cv = eng.decode_batch(s, want_llr=False, osd=True)
        assert not cv.all()
        assert not np.any((h @ dec.T % 2).T != s)
        want = o.bposd_decode_batch(s[:4], method, order, want_llr=False)
        assert np.array_equal(dec[:4], want[0]) and np.array_equal(cv[:4], want[3])


@pytest.mark.parametrize("method,alpha", [("product_sum", 1.0), ("minimum_sum", 0.0)])
def test_irregular_code_on_the_generic_degree_streamed_kernels(method, alpha, oracle_built):
    """An irregular LDPC code (rows of 3 ... 16 entries, columns of 2 ... 8) forced onto the streamed kernels -- the variants with one
    16-entry row in registers, no register double buffer and 8-wavefront workgroups (bp_decode_kernel<., ., 16, 8, 0>; zero VGPR spills
    since round 4) -- against the CPU checker on every row, log-ratio bits included; batches that take the persistent kernel with its
    hand-off and the per-pass kernels alone; every workgroup size the variant allows (and one it does not: clamped)."""
    from golden_util import bits_equal
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    h = codes.irregular_ldpc_code(600, 300, seed=3)
    n, p = 600, 0.03
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), 16, 0 if method == "product_sum" else 1, alpha)
    eng.set_small_code_kernel(0)
    s = eng.gen_bsc_syndromes(13, p, shot0=0, shots=40000, device="cuda:0")
    sh = s.cpu().numpy()
    rows = np.r_[0:300, 39700:40000]
    want = oracle_built.BpOracle(h, error_rate=p, max_iter=16, bp_method=method, ms_scaling_factor=alpha).decode_batch(sh[rows])
    assert 0.2 < want[3].mean() < 0.999, "the case is meant to mix converged and unconverged rows"
    for waves, handoff in ((0, -1), (4, 0), (8, 64), (16, -1)):
        eng.set_tuning(waves_per_workgroup=waves)
        eng.set_handoff(handoff)
        dec, llr, it, cv = eng.decode_batch(s, want_llr=True)
        _check_flags_against_syndromes(eng, s, dec, cv, it, 16)
        tag = f"waves {waves} handoff {handoff}"
        assert np.array_equal(dec.cpu().numpy()[rows], want[0]) and np.array_equal(it.cpu().numpy()[rows], want[2]), tag
        assert np.array_equal(cv.cpu().numpy()[rows].astype(bool), want[3]) and bits_equal(llr.cpu().numpy()[rows], want[1]), tag
    small = eng.decode_batch(s[:300].contiguous(), want_llr=True)  # five tiles: per-pass kernels from the first iteration
    assert np.array_equal(small[0].cpu().numpy(), want[0][:300]) and bits_equal(small[1].cpu().numpy(), want[1][:300])
    eng.close()


def test_config4_full_batch_one_gpu(oracle_built):
    """BASELINE configs[3]'s WHOLE batch -- 1 048 576 syndromes of the (3,6)-regular n = 10 000 code, product_sum, 50 iterations, p = 0.09 --
    in ONE decode_batch on one GPU: 504 GB of messages, so the memory-driven chunk loop of decode_device (hipMemGetInfo) cuts it into
    several chunks of thousands of tiles.  flag <=> H x == s on every row; rows [0, 131 072) -- the first GPU's share of the eight --
    equal to a decode of that share alone (decisions, iteration counts, flags), an oracle subset from both ends of the batch."""
    import torch
    from ldpc_amd.codes import regular_ldpc_code
    from ldpc_amd.engine import HipBpEngine
    dev = torch.device("cuda", 0)
    p = 0.09
    h = regular_ldpc_code(10000, 3, 6, seed=1)
    eng = HipBpEngine(h.indptr, h.indices, 10000, np.full(10000, p), 50, 0, 1.0)
    B, share = 1048576, 131072
    synd = eng.gen_bsc_syndromes(7, p, shot0=0, shots=B, device=dev)
    dec, _, it, cv = eng.decode_batch(synd, want_llr=False)
    torch.cuda.synchronize()
    tiles, per_tile = B // 64, 2 * 30000 * 64 * 8
    free_b, total_b = torch.cuda.mem_get_info()
    assert tiles * per_tile > total_b, "the batch's messages must not fit the GPU at once (else this is not the chunk loop)"
    for lo in range(0, B, share):  # (H x for 131 072 rows at a time: the residual of the whole batch is another 5 GB)
        sl = slice(lo, lo + share)
        cvb = _check_flags_against_syndromes(eng, synd[sl], dec[sl], cv[sl], it[sl], 50)
        assert cvb.float().mean().item() < 0.05
    d1, _, i1, c1 = eng.decode_batch(synd[:share].contiguous(), want_llr=False)
    assert bool(torch.equal(d1, dec[:share])) and bool(torch.equal(i1, it[:share])) and bool(torch.equal(c1, cv[:share]))
    rows = torch.from_numpy(np.r_[0:6, B // 2 - 3:B // 2 + 3, B - 6:B]).to(dev)
    _subset_vs_oracle(oracle_built, h, p, 50, "product_sum", 1.0, synd, dec, it, cv, None, rows)
    eng.close()


def test_eight_shards_on_one_device_equal_the_single_handle():
    """ldpc_hip_bp_multi with the device listed eight times: eight handles, eight host threads, eight shards of 8 192 rows of configs[1]'s
    stream -- what an eight-GPU node runs in one process, on the one GPU of this box -- against the single-handle decode of the 65 536 rows."""
    import torch
    from ldpc_amd.codes import regular_ldpc_code
    from ldpc_amd.engine import HipBpEngine, HipBpMultiEngine
    dev = torch.device("cuda", 0)
    p = 0.05
    h = regular_ldpc_code(10000, 3, 6, seed=1)
    one = HipBpEngine(h.indptr, h.indices, 10000, np.full(10000, p), 50, 0, 1.0)
    many = HipBpMultiEngine(h.indptr, h.indices, 10000, np.full(10000, p), 50, 0, 1.0, [0] * 8)
    B = 8 * 8192
    synd = one.gen_bsc_syndromes(7, p, shot0=0, shots=B, device=dev)
    a = one.decode_batch(synd, want_llr=True)
    b = many.decode_batch(synd, want_llr=True)
    assert bool(torch.equal(a[0], b[0])) and bool(torch.equal(a[2], b[2])) and bool(torch.equal(a[3], b[3]))
    same = (a[1].view(torch.int64) == b[1].view(torch.int64)) | (a[1].isnan() & b[1].isnan())
    assert bool(same.all())
    assert len(many.last_kernel_ms()) == 8
