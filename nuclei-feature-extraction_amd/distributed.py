"""Index sharding of nuclei over ranks and the gather of feature rows.

The reference's multi-GPU mode is `gpus[rayon_thread_idx % gpu_count]` per batch plus a host-side
`vstack_mut` under a mutex (utils.rs:215-221, main.rs:96-108).  Here: one process per GPU, each rank
extracts a contiguous index shard whose boundaries are multiples of the reference batch size (the
hue statistics couple the nuclei of a batch, color.rs:144-155, so a batch must live on one rank),
with no data-path collective; the rows are then gathered to one rank with a single collective
(`torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import List, Optional, Tuple


def shard_range(n: int, batch_size: int, rank: int, world: int) -> Tuple[int, int]:
    """Nuclei [i0, i1) of `rank`: rank k gets reference batches [k*ceil(nb/R), (k+1)*ceil(nb/R))."""
    if batch_size < 1 or world < 1 or not (0 <= rank < world):
        raise ValueError("bad shard arguments")
    nb = (n + batch_size - 1) // batch_size
    per = (nb + world - 1) // world
    b0 = min(rank * per, nb)
    b1 = min((rank + 1) * per, nb)
    return min(b0 * batch_size, n), min(b1 * batch_size, n)


def shard_counts(n: int, batch_size: int, world: int) -> List[int]:
    return [shard_range(n, batch_size, r, world)[1] - shard_range(n, batch_size, r, world)[0] for r in range(world)]


def gather_rows(local_rows, counts: List[int], dst: int = 0, group=None, out=None):
    """Gathers per-rank row blocks [counts[r], F] to `dst` in rank (= index) order.

    Ragged shards are padded to the largest count for the collective; `out` ([sum(counts), F]) may be
    preallocated on dst.  Returns the gathered matrix on dst, None elsewhere.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(counts) == world and local_rows.shape[0] == counts[rank]
    F = local_rows.shape[1]
    m = max(counts)
    if local_rows.shape[0] != m:
        pad = torch.zeros((m, F), dtype=local_rows.dtype, device=local_rows.device)
        pad[: local_rows.shape[0]] = local_rows
        local_rows = pad
    local_rows = local_rows.contiguous()
    if rank == dst:
        bufs = [torch.empty((m, F), dtype=local_rows.dtype, device=local_rows.device) for _ in range(world)]
        dist.gather(local_rows, bufs, dst=dst, group=group)
        if out is None:
            out = torch.empty((sum(counts), F), dtype=local_rows.dtype, device=local_rows.device)
        o = 0
        for r in range(world):
            out[o:o + counts[r]] = bufs[r][: counts[r]]
            o += counts[r]
        return out
    dist.gather(local_rows, None, dst=dst, group=group)
    return None
