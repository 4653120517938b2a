"""The N > 1 path on CPU: world_size-2 gloo process group, batch sharded by rows, one gather of decoded rows.

The per-rank "decoder" here is the CPU checker (this is a TEST of the sharding/gather logic in
ldpc_amd/sharding.py, which is device-agnostic; on the GPU box the same functions run over RCCL).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ldpc_amd.sharding import shard_range


def test_shard_range_covers_everything_once():
    for total in (0, 1, 7, 64, 65, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from ldpc_amd.codes import bivariate_bicycle_hx
        from ldpc_amd.noise_models import generate_bsc_batch
        from ldpc_amd.sharding import decode_sharded, shard_range as sr
        h = bivariate_bicycle_hx()
        lo, hi = sr(total, rank, world)
        err = generate_bsc_batch(144, 0.05, 7, lo, hi - lo)  # this rank's slice of the global shot stream
        synd = (err.astype(np.int64) @ h.T.toarray().astype(np.int64) % 2).astype(np.uint8)
        dec = oracle.BpOracle(h, error_rate=0.05, max_iter=30)

        def decode_fn(s):
            d, l, it, cv = dec.decode_batch(s.numpy())
            return torch.from_numpy(d), torch.from_numpy(l), torch.from_numpy(it), torch.from_numpy(cv.astype(np.uint8))

        (gd, gc, gi), llr_local = decode_sharded(decode_fn, torch.from_numpy(synd), total, dst=0)
        assert llr_local.shape == (hi - lo, 144)
        if rank == 0:
            q.put((gd.numpy(), gc.numpy(), gi.numpy()))
        else:
            assert gd is None and gc is None and gi is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [101, 64])
def test_sharded_decode_equals_single_process(total):
    import oracle
    from ldpc_amd.codes import bivariate_bicycle_hx
    from ldpc_amd.noise_models import generate_bsc_batch
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    h = bivariate_bicycle_hx()
    err = generate_bsc_batch(144, 0.05, 7, 0, total)
    synd = (err.astype(np.int64) @ h.T.toarray().astype(np.int64) % 2).astype(np.uint8)
    wd, _, wi, wc = oracle.BpOracle(h, error_rate=0.05, max_iter=30).decode_batch(synd)
    assert np.array_equal(got[0], wd) and np.array_equal(got[1].astype(bool), wc) and np.array_equal(got[2], wi)
