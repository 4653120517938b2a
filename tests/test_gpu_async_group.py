"""The *_async entry points really are asynchronous, a change of launch stream is ordered after pending work, and the
sharded path runs under a real process group (nccl = RCCL) -- world size 1 on a one-GPU box, through bench.py's own launcher."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _assert_same(got, want):
    """Outputs equal bit for bit; among the log-ratios any NaN matches any NaN (inf - inf in the reference's arithmetic too: the sign /
    payload of a NaN is not a result -- it depends on which of two bit-identical routes a 64-syndrome tile took, oracle.bits_equal)."""
    import torch
    for g, w in zip(got, want):
        if g.dtype == torch.float64:
            assert bool(((g.view(torch.int64) == w.view(torch.int64)) | (g.isnan() & w.isnan())).all())
        else:
            assert bool(torch.equal(g, w))


def _headline_engine(p=0.09, max_iter=50):
    from ldpc_amd.codes import regular_ldpc_code
    from ldpc_amd.engine import HipBpEngine
    h = regular_ldpc_code(10000, 3, 6, seed=1)
    return h, HipBpEngine(h.indptr, h.indices, 10000, np.full(10000, p), max_iter, 0, 1.0)


@pytest.mark.parametrize("p", [0.09, 0.05])
def test_decode_batch_async_returns_before_the_kernels_finish(p):
    """bench.py's workload at B = 65 536 runs for ~0.6 s on the device (p = 0.09; ~0.1 s at p = 0.05): the call must come back long
    before that, with the launch stream still busy, and (second call) without waiting for the hand-off of the last tiles.  At
    p = 0.05 the second call is the TWO-PASS decode (steered by the first call's iteration histogram, whose copy has landed): its
    second pass is sized on the device -- no host round trip for the row count, no event waits."""
    import time
    import torch
    h, eng = _headline_engine(p=p)
    B = 65536
    dev = torch.device("cuda", 0)
    synd = eng.gen_bsc_syndromes(7, p, shot0=0, shots=B, device=dev)
    out = eng.decode_batch(synd, want_llr=True)  # warm-up: allocations, module load; leaves its iteration histogram
    torch.cuda.synchronize()
    plain_ms = eng.last_kernel_ms()
    stream = torch.cuda.current_stream(dev)
    t0 = time.perf_counter()
    eng.decode_batch(synd, want_llr=True, out=out, asynchronous=True)
    t_call = time.perf_counter() - t0
    busy = not stream.query()
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    assert busy, "the launch stream was idle when decode_batch_async returned"
    assert t_call < 0.5 * t_all, f"call took {t_call * 1e3:.1f} ms of {t_all * 1e3:.1f} ms: it waited for the device"  # (a few ms; slack for a shared box)
    second_ms = eng.last_kernel_ms()
    if p == 0.05:
        assert second_ms < 0.97 * plain_ms, f"{second_ms:.1f} ms vs {plain_ms:.1f} ms plain: the steered call should have compacted the live lanes"
    # the results of the asynchronous call are those of the synchronous one, and of a plain single-pass decode
    ref = eng.decode_batch(synd, want_llr=True)
    _assert_same(out, ref)
    eng.set_repack(0)
    _assert_same(out, eng.decode_batch(synd, want_llr=True))
    assert second_ms > (100.0 if p == 0.09 else 30.0)


def test_change_of_stream_waits_for_the_previous_decode(oracle_built):
    """One handle owns one workspace: an asynchronous decode on stream A followed by a decode on stream B must not overlap."""
    import torch
    from ldpc_amd.codes import regular_ldpc_code
    from ldpc_amd.engine import HipBpEngine
    h = regular_ldpc_code(2400, 3, 6, seed=5)
    p = 0.07
    eng = HipBpEngine(h.indptr, h.indices, 2400, np.full(2400, p), 40, 0, 1.0)
    eng.set_small_code_kernel(0)
    dev = torch.device("cuda", 0)
    sa = eng.gen_bsc_syndromes(7, p, shot0=0, shots=8192, device=dev)
    sb = eng.gen_bsc_syndromes(7, p, shot0=8192, shots=8192, device=dev)
    want_a = eng.decode_batch(sa, want_llr=True)
    want_b = eng.decode_batch(sb, want_llr=True)
    torch.cuda.synchronize()
    st_a, st_b = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    for _ in range(3):
        with torch.cuda.stream(st_a):
            got_a = eng.decode_batch(sa, want_llr=True, asynchronous=True)
        with torch.cuda.stream(st_b):
            got_b = eng.decode_batch(sb, want_llr=True, asynchronous=True)
        torch.cuda.synchronize()
        _assert_same(got_a, want_a)
        _assert_same(got_b, want_b)
    o = oracle_built.BpOracle(h, error_rate=p, max_iter=40, bp_method="product_sum")
    chk = o.decode_batch(sa[:32].cpu().numpy())
    assert np.array_equal(want_a[0][:32].cpu().numpy(), chk[0])


def test_large_max_iter_stops_queueing_rounds():
    """The reference's default max_iter = n: the per-pass rounds are queued without waiting, but not all n of them once
    the device has reported that nothing is left to do -- and the results are those of a short max_iter where everything converges."""
    import time
    import torch
    from ldpc_amd.codes import regular_ldpc_code
    from ldpc_amd.engine import HipBpEngine
    h = regular_ldpc_code(2400, 3, 6, seed=5)
    p = 0.03
    dev = torch.device("cuda", 0)
    short = HipBpEngine(h.indptr, h.indices, 2400, np.full(2400, p), 60, 0, 1.0)
    long_ = HipBpEngine(h.indptr, h.indices, 2400, np.full(2400, p), 200000, 0, 1.0)
    for e in (short, long_):
        e.set_small_code_kernel(0)
    s = short.gen_bsc_syndromes(7, p, shot0=0, shots=4096, device=dev)
    a = short.decode_batch(s, want_llr=True)
    assert bool(a[3].bool().all()), "workload of this test: everything converges"
    long_.decode_batch(s, want_llr=True)
    t0 = time.perf_counter()
    b = long_.decode_batch(s, want_llr=True)
    dt = time.perf_counter() - t0
    _assert_same(b, a)
    assert dt < 2.0, f"{dt:.2f} s: 200 000 rounds were queued although the batch converged within 60"


def _bench(args, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_under_a_process_group_of_one():
    """`python bench.py --gpus 1 --force-launch`: torch.distributed.run, init_process_group('nccl'), HipBpEngine under the
    group, gather_rows(eng.pack_b8(dec)) + flags, per-rank parity -- the N > 1 code path on the one GPU this box has."""
    got = _bench(["--gpus", "1", "--force-launch", "--batch-per-gpu", "4096", "--steps", "2", "--warmup", "1", "--cpu-sample", "0",
                  "--rank-parity", "48", "--secondary", "0"])
    assert got["n_gpus"] == 1 and got["rccl"]["ranks"] == 1 and got["rccl"]["backend"] == "nccl"
    assert got["per_rank"]["parity_all_ranks"] is True and got["per_rank"]["parity_rows_per_rank"] == 48
    assert len(got["per_rank"]["kernel_ms"]) == 1 and got["per_rank"]["kernel_ms"][0] > 0
    assert got["gather"]["rows_on_rank0"] == 4096
    assert "parity_failed" not in got


def test_gather_rows_of_the_engine_under_nccl():
    """In-process (a fresh interpreter): world-1 nccl group, decode, pack_b8, gather_rows, unpack -- equal to the decode."""
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from ldpc_amd.codes import regular_ldpc_code
from ldpc_amd.engine import HipBpEngine
from ldpc_amd.sharding import decode_sharded, shard_range
h = regular_ldpc_code(1200, 3, 6, seed=2)
eng = HipBpEngine(h.indptr, h.indices, 1200, np.full(1200, 0.06), 30, 0, 1.0)
s = eng.gen_bsc_syndromes(7, 0.06, shot0=0, shots=1000, device=torch.device("cuda", 0))
(dec8, cv, it), llr = decode_sharded(lambda x: eng.decode_batch(x, want_llr=True), s, 1000, dst=0, pack=eng.pack_b8)
t = torch.ones(1, device="cuda"); dist.all_reduce(t)   # a real collective on the RCCL communicator
want = eng.decode_batch(s, want_llr=True)
assert dec8.shape == (1000, 150) and bool((eng.unpack_b8(dec8, 1200) == want[0]).all())
assert bool((cv == want[3]).all()) and bool((it == want[2]).all()) and bool(torch.equal(llr.view(torch.int64), want[1].view(torch.int64)))
assert shard_range(1000, 0, 1) == (0, 1000) and float(t.item()) == 1.0
dist.destroy_process_group()
print("ok")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_two_ranks_share_the_one_gpu():
    """world_size 2 on hardware: two processes, each with its own HipBpEngine on cuda:0 (RCCL refuses two ranks on one device, so the
    group is gloo and the rows travel as host tensors): shards of one shot stream decoded concurrently, gathered on rank 0, equal -- bits
    of the log-ratios included -- to one decode of all rows (tests/two_rank_gpu_worker.py)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "1"
    import bench  # (free_port: a fixed port can collide on a shared box)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(bench.free_port()),
           os.path.join(ROOT, "tests", "two_rank_gpu_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "two ranks ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_bench_with_two_ranks_on_the_one_gpu():
    """`python bench.py --gpus 2 --share-gpu`: the N > 1 bookkeeping of the bench on hardware -- two shards of the shot stream, barrier,
    max over ranks, gather onto rank 0, per-rank parity and timings -- with both ranks on cuda:0 under a gloo group (a diagnostic: RCCL
    refuses two ranks on one device, and the rate of two processes sharing a GPU means nothing)."""
    got = _bench(["--gpus", "2", "--share-gpu", "--batch-per-gpu", "4096", "--steps", "2", "--warmup", "1", "--rank-parity", "48"])
    assert got["n_gpus"] == 2 and got["rccl"]["ranks"] == 2 and got["rccl"]["backend"] == "gloo"
    assert got["config"]["global_batch"] == 8192 and got["gather"]["rows_on_rank0"] == 8192
    assert got["per_rank"]["parity_ok"] == [True, True] and got["per_rank"]["parity_all_ranks"] is True
    assert len(got["per_rank"]["kernel_ms"]) == 2 and min(got["per_rank"]["kernel_ms"]) > 0
    assert got["cpu_baseline"] == "N = 1 only" and "secondary" not in got and "parity_failed" not in got
    assert abs(got["value"] - 8192 / (got["ms_per_step"] * 1e-3)) < 1e-6 * got["value"]
