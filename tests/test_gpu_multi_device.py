"""ldpc_hip_bp_multi: one decoder object over several GPUs inside one process (`device_ids[ndev]`).

The sharded call must give, row for row and bit for bit, what the single-GPU call gives.  A one-GPU box still runs every
path: a device may be listed twice (two handles, two shards, two host threads on one GPU) and `set_staging(True)` sends
device tensors through the peer-copy / bit-packed route even on the GPU they live on."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _code():
    from ldpc_amd.codes import regular_ldpc_code
    return regular_ldpc_code(1200, 3, 6, seed=2)


def _engines(h, p, max_iter, method, alpha, ids):
    from ldpc_amd.engine import HipBpEngine, HipBpMultiEngine
    n = h.shape[1]
    one = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), max_iter, method, alpha)
    many = HipBpMultiEngine(h.indptr, h.indices, n, np.full(n, p), max_iter, method, alpha, ids)
    return one, many


def _same(a, b):
    import torch
    for x, y in zip(a, b):
        if x is None or y is None:
            assert x is None and y is None
        elif isinstance(x, np.ndarray):
            assert np.array_equal(x.view(np.int64) if x.dtype == np.float64 else x, y.view(np.int64) if y.dtype == np.float64 else y)
        else:
            assert bool(torch.equal(x.view(torch.int64) if x.dtype == torch.float64 else x, y.view(torch.int64) if y.dtype == torch.float64 else y))


@pytest.mark.parametrize("ids", [[0], [0, 0], [0, 0, 0]])
@pytest.mark.parametrize("method,alpha", [(0, 1.0), (1, 0.625)])
def test_host_arrays_sharded_equals_single(ids, method, alpha, oracle_built):
    h = _code()
    p = 0.06
    one, many = _engines(h, p, 30, method, alpha, ids)
    for B in (1, 63, 64, 1000):  # fewer tiles than shards, ragged last tile
        s = one.gen_bsc_syndromes(7, p, shot0=5, shots=B)
        _same(one.decode_batch(s), many.decode_batch(s))
    assert len(many.last_kernel_ms()) == len(ids) and many.last_kernel_ms()[0] > 0
    s = one.gen_bsc_syndromes(7, p, shot0=5, shots=200)
    o = oracle_built.BpOracle(h, error_rate=p, max_iter=30, bp_method=method, ms_scaling_factor=alpha)
    want = o.decode_batch(s)
    got = many.decode_batch(s)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3])


@pytest.mark.parametrize("staged", [False, True])
def test_device_tensors_sharded_equals_single(staged):
    import torch
    h = _code()
    p = 0.06
    one, many = _engines(h, p, 30, 0, 1.0, [0, 0])
    many.set_staging(staged)
    dev = torch.device("cuda", 0)
    for B in (100, 4097):
        s = one.gen_bsc_syndromes(7, p, shot0=0, shots=B, device=dev)
        _same(one.decode_batch(s), many.decode_batch(s))
        _same(one.decode_batch(s, want_llr=False), many.decode_batch(s, want_llr=False))


def test_setters_reach_every_gpu_and_osd_runs_sharded(oracle_built):
    from ldpc_amd.codes import bivariate_bicycle_hx
    h = bivariate_bicycle_hx()
    p = 0.06
    one, many = _engines(h, p, 50, 0, 1.0, [0, 0])
    s = one.gen_bsc_syndromes(7, p, shot0=0, shots=2000)
    _same(one.decode_batch(s, osd0=True), many.decode_batch(s, osd0=True))
    for e in (one, many):
        e.set_osd(3, 10)
        e.set_params(20, 1, 0.75)
        e.set_channel(np.linspace(0.02, 0.08, h.shape[1]))
    a, b = one.decode_batch(s, osd=True), many.decode_batch(s, osd=True)
    _same(a, b)
    assert not a[3].all() and a[3].any()  # the workload has both kinds of rows
    import torch
    many.set_staging(True)
    sd = torch.from_numpy(s).cuda()
    c = many.decode_batch(sd, osd=True)
    assert np.array_equal(c[0].cpu().numpy(), a[0]) and np.array_equal(c[1].cpu().numpy().view(np.int64), a[1].view(np.int64))


def test_every_visible_gpu():
    import torch
    ngpu = torch.cuda.device_count()
    if ngpu < 2:
        pytest.skip("one GPU visible")
    h = _code()
    p = 0.06
    one, many = _engines(h, p, 30, 0, 1.0, list(range(ngpu)))
    s = one.gen_bsc_syndromes(7, p, shot0=0, shots=64 * ngpu * 3 + 17)
    _same(one.decode_batch(s), many.decode_batch(s))
    sd = torch.from_numpy(s).cuda(0)
    got = many.decode_batch(sd)  # rows leave GPU 0 by peer copy, decisions come back bit-packed
    want = one.decode_batch(sd)
    _same(want, got)


def test_bpdecoder_device_ids_keyword():
    from ldpc_amd.bp_decoder import BpDecoder
    from ldpc_amd.bposd_decoder import BpOsdDecoder
    from ldpc_amd.noise_models import generate_bsc_batch
    h = _code()
    err = generate_bsc_batch(1200, 0.05, seed=3, shot0=0, shots=300)
    synd = (err.astype(np.int64) @ h.T.toarray().astype(np.int64) % 2).astype(np.uint8)
    a = BpDecoder(h, error_rate=0.05, max_iter=25, bp_method="ms", ms_scaling_factor=0.8)
    b = BpDecoder(h, error_rate=0.05, max_iter=25, bp_method="ms", ms_scaling_factor=0.8, device_ids=[0, 0])
    assert np.array_equal(a.decode_batch(synd), b.decode_batch(synd))
    assert np.array_equal(a.iter_batch, b.iter_batch) and np.array_equal(a.log_prob_ratios_batch.view(np.int64), b.log_prob_ratios_batch.view(np.int64))
    assert np.array_equal(a.decode(synd[7]), b.decode(synd[7]))
    b.max_iter = 3  # setters reach both handles
    a.max_iter = 3
    assert np.array_equal(a.decode_batch(synd), b.decode_batch(synd)) and np.array_equal(a.converge_batch, b.converge_batch)
    with pytest.raises(ValueError):
        BpDecoder(h, error_rate=0.05, device_ids=[])
    c = BpOsdDecoder(h, error_rate=0.05, max_iter=6, osd_method="osd_0", device_ids=[0, 0])
    d = BpOsdDecoder(h, error_rate=0.05, max_iter=6, osd_method="osd_0")
    assert np.array_equal(c.decode_batch(synd), d.decode_batch(synd))


def test_bad_device_lists_are_errors():
    from ldpc_amd import _lib
    from ldpc_amd.engine import HipBpMultiEngine
    h = _code()
    with pytest.raises(_lib.LdpcHipError, match="device"):
        HipBpMultiEngine(h.indptr, h.indices, 1200, np.full(1200, 0.05), 10, 0, 1.0, [0, 99])


@pytest.mark.parametrize("kind", ["serial_relative", "random", "random_clock_seed"])
def test_stateful_schedules_keep_one_state_over_all_handles(kind):
    """serial_relative / the random serial schedule carry their order (and generator) from call to call (bp.hpp:467-483):
    a SEQUENCE of sharded calls must equal the same sequence on one GPU -- including batches that leave shards without rows."""
    from ldpc_amd.codes import bivariate_bicycle_hx
    h = bivariate_bicycle_hx()
    p = 0.06
    one, many = _engines(h, p, 12, 0, 1.0, [0, 0, 0])
    for e in (one, many):
        if kind == "serial_relative":
            e.set_schedule("serial_relative")
        else:
            e.set_schedule("serial")
            e.set_random_serial(True, 1234)
    if kind == "random_clock_seed":  # seed 0 = every handle reads the clock on its own: the first call must still use ONE generator
        many.set_random_serial(True, 0)
        s = one.gen_bsc_syndromes(7, p, shot0=0, shots=64 * 3 + 5)
        got = many.decode_batch(s)
        orders = [sub.schedule_order() for sub in many.subs]
        assert all(np.array_equal(orders[0], o) for o in orders[1:])
        assert got[3].any()
        return
    for shot0, B in ((0, 200), (300, 1), (400, 64 * 3 + 9), (900, 70)):  # B = 1 and 70: shards without rows
        s = one.gen_bsc_syndromes(7, p, shot0=shot0, shots=B)
        _same(one.decode_batch(s), many.decode_batch(s))
        assert np.array_equal(one.schedule_order(), many.schedule_order())
        assert all(np.array_equal(one.schedule_order(), sub.schedule_order()) for sub in many.subs)


def test_soft_info_on_a_multi_engine_runs_on_its_first_gpu():
    from ldpc_amd.codes import bivariate_bicycle_hx
    h = bivariate_bicycle_hx()
    one, many = _engines(h, 0.05, 20, 1, 0.9, [0, 0])
    rng = np.random.default_rng(5)
    soft = rng.normal(1.0, 0.8, size=(100, h.shape[0]))
    _same(one.soft_info_decode_batch(soft, 2.0, 0.7), many.soft_info_decode_batch(soft, 2.0, 0.7))


def test_two_ranks_over_rccl_when_two_gpus_are_visible():
    """The first box with two GPUs proves the N > 1 path: `python bench.py --gpus 2` as the driver launches
    it -- torch.distributed.run, one rank per GPU, an RCCL group of two, contiguous row shards, per-rank parity against the CPU checker,
    one gather of bit-packed decoded rows + flags onto rank 0 (ldpc_amd/sharding.py).  On the one-GPU boxes of this pool it skips; no
    N > 1 curve has been measured anywhere yet."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    from test_gpu_async_group import _bench
    got = _bench(["--gpus", "2", "--batch-per-gpu", "16384", "--steps", "2", "--warmup", "1", "--rank-parity", "64",
                  "--secondary", "0", "--cpu-sample", "0", "--host-io", "0"], timeout=1500)
    assert got["n_gpus"] == 2 and got["rccl"]["ranks"] == 2 and got["rccl"]["backend"] == "nccl"
    assert got["config"]["batch_per_gpu"] == 16384 and got["config"]["global_batch"] == 32768
    assert got["gather"]["rows_on_rank0"] == 32768
    assert got["per_rank"]["parity_all_ranks"] is True and len(got["per_rank"]["parity_ok"]) == 2 and all(got["per_rank"]["parity_ok"])
    assert "parity_failed" not in got and got["value"] > 0 and got["scaling"] == "weak"
    assert abs(got["value"] - 32768 / (got["ms_per_step"] * 1e-3)) < 1e-6 * got["value"]
    # and the one-process form with a real peer copy between two distinct devices
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine, HipBpMultiEngine
    h = codes.regular_ldpc_code(600, 3, 6, seed=3)
    one = HipBpEngine(h.indptr, h.indices, 600, np.full(600, 0.04), 20, 0, 1.0)
    two = HipBpMultiEngine(h.indptr, h.indices, 600, np.full(600, 0.04), 20, 0, 1.0, device_ids=[0, 1])
    s = one.gen_bsc_syndromes(5, 0.04, shot0=0, shots=9000, device="cuda:0")
    a = [x.cpu().numpy() for x in one.decode_batch(s, want_llr=True)]
    b = [x.cpu().numpy() if hasattr(x, "cpu") else x for x in two.decode_batch(s, want_llr=True)]
    from golden_util import bits_equal
    assert np.array_equal(a[0], b[0]) and bits_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    one.close()
    two.close()
