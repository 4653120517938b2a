"""Multi-GPU data parallelism for batched decoding: one process per GPU, syndromes sharded by rows.

Syndromes are independent (they share only the read-only H and priors), so a batch is
cut into contiguous row ranges, every rank decodes its own range with no communication, and the only
collective is ONE gather of the decoded rows at the end (RCCL over xGMI when the process group is
``nccl``; ``gloo`` in the CPU tests).  The reference has no counterpart -- it is single-threaded
(bp.hpp:129-140: OpenMP is a stub).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous ``[lo, hi)`` row range of rank ``rank``; sizes differ by at most one row."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def gather_rows(local: torch.Tensor, total_rows: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Gather row shards (cut by ``shard_range``) of a ``(rows, ...)`` tensor onto ``dst``.

    Returns the ``(total_rows, ...)`` tensor on ``dst`` and ``None`` elsewhere.  Ragged shards are
    handled by padding every shard to the largest one for the collective.
    """
    rank, world = _world()
    if world == 1:
        return local
    sizes = [shard_range(total_rows, r, world) for r in range(world)]
    biggest = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < biggest:
        filler = torch.zeros((biggest - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad = torch.cat([local, filler], dim=0)
    pad = pad.contiguous()
    bufs: Optional[List[torch.Tensor]] = None
    if rank == dst:
        bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.gather(pad, gather_list=bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


def decode_sharded(decode_fn, syndromes_local: torch.Tensor, total_rows: int, dst: int = 0, pack=None):
    """Decode this rank's shard with ``decode_fn`` and gather ``(decoding, converge, iterations)`` on ``dst``.

    ``pack`` (e.g. ``HipBpEngine.pack_b8``) is applied to the decoded rows before the collective: bit-packed rows
    are 1/8 of the bytes on the wire; ``dst`` then receives ``(total_rows, ceil(n / 8))`` b8 rows.

    ``decode_fn(syndromes) -> (decoding, llr_or_None, iterations, converge)`` is e.g.
    ``HipBpEngine.decode_batch``.  LLRs stay sharded on the rank that produced them (they are 8x the
    size of the decisions and normally consumed locally).
    """
    dec, llr, it, cv = decode_fn(syndromes_local)
    out = (gather_rows(pack(dec) if pack is not None else dec, total_rows, dst), gather_rows(cv, total_rows, dst),
           gather_rows(it, total_rows, dst))
    return out, llr
