"""Worker of tests/test_gpu_async_group.py::test_two_ranks_share_the_one_gpu (launched by torch.distributed.run, two ranks):
both ranks drive their OWN HipBpEngine on cuda:0 -- the box has one GPU, and RCCL refuses two ranks on one device, so the group is gloo
and the decoded rows travel as host tensors -- each decodes its shard of one counter-based shot stream, rank 0 gathers
(ldpc_amd.sharding.gather_rows) and compares with ONE decode of all rows.  Not collected by pytest."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldpc_amd.codes import regular_ldpc_code  # noqa: E402
from ldpc_amd.engine import HipBpEngine  # noqa: E402
from ldpc_amd.sharding import gather_rows, shard_range  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    n, p, total = 1200, 0.06, 1001  # (ragged shards: 501 + 500)
    h = regular_ldpc_code(n, 3, 6, seed=2)
    for method, alpha, small in ((0, 1.0, 0), (1, 0.625, -1)):  # streamed product-sum, on-chip min-sum
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), 30, method, alpha)
        eng.set_small_code_kernel(small)
        lo, hi = shard_range(total, rank, world)
        s = eng.gen_bsc_syndromes(7, p, shot0=lo, shots=hi - lo, device=dev)
        dist.barrier()  # both engines decode at the same time on the one device
        dec, llr, it, cv = eng.decode_batch(s, want_llr=True)
        got = [gather_rows(t.cpu(), total, 0) for t in (eng.pack_b8(dec), llr, it, cv)]
        if rank == 0:
            s_all = eng.gen_bsc_syndromes(7, p, shot0=0, shots=total, device=dev)
            want = eng.decode_batch(s_all, want_llr=True)
            assert torch.equal(eng.unpack_b8(got[0].to(dev), n), want[0]), "decisions"
            assert torch.equal(got[1].view(torch.int64), want[1].cpu().view(torch.int64)), "log-ratio bits"
            assert torch.equal(got[2], want[2].cpu()) and torch.equal(got[3], want[3].cpu()), "iterations / flags"
        else:
            assert all(g is None for g in got)
        eng.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("two ranks ok")


if __name__ == "__main__":
    main()
