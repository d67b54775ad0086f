"""Multi-GPU plumbing: the batch of independent preview systems is sharded contiguously over the ranks (one process
per GPU); there is NO data-path exchange between instances, so the only collective is ONE gather of the packed
result slab [U | X | status | iter] to rank 0 (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """contiguous instance range [lo, hi) of `rank`"""
    lo = (total * rank) // world
    hi = (total * (rank + 1)) // world
    return lo, hi


def slab_layout(batch, n, X):
    """byte offsets of the four result arrays inside one flat slab (all 8-byte aligned)"""
    off_u = 0
    off_x = off_u + batch * n * 8
    off_s = off_x + batch * X * 8
    off_i = off_s + ((batch * 4 + 7) // 8) * 8
    total = off_i + batch * 2 * 4
    total = ((total + 7) // 8) * 8
    return off_u, off_x, off_s, off_i, total


def alloc_result_slab(batch, n, X, device):
    """One flat byte buffer holding control [b,n] f64, trajectory [b,X] f64, status [b] i32, iter [b,2] i32.
    The typed views alias the slab, so the engine writes straight into what the gather sends (no packing copy)."""
    off_u, off_x, off_s, off_i, total = slab_layout(batch, n, X)
    slab = torch.zeros(total, dtype=torch.uint8, device=device)
    views = dict(
        control=slab[off_u:off_x].view(torch.float64).view(batch, n),
        trajectory=slab[off_x:off_s].view(torch.float64).view(batch, X),
        status=slab[off_s:off_s + batch * 4].view(torch.int32),
        iter=slab[off_i:off_i + batch * 8].view(torch.int32).view(batch, 2),
    )
    return slab, views


def split_slab(slab, batch, n, X):
    off_u, off_x, off_s, off_i, total = slab_layout(batch, n, X)
    return dict(
        control=slab[off_u:off_x].view(torch.float64).view(batch, n),
        trajectory=slab[off_x:off_s].view(torch.float64).view(batch, X),
        status=slab[off_s:off_s + batch * 4].view(torch.int32),
        iter=slab[off_i:off_i + batch * 8].view(torch.int32).view(batch, 2),
    )


def alloc_gather_buffers(slab, rank, world):
    if rank != 0:
        return None
    return [torch.empty_like(slab) for _ in range(world)]


def gather_results(slab, rank, world, bufs, dst=0, force=False):
    """the single collective of the path: every rank's slab -> rank `dst` (force: also with a one-rank group)"""
    if world == 1 and not force:
        return [slab]
    dist.gather(slab, gather_list=bufs if rank == dst else None, dst=dst)
    return bufs
