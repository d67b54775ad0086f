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
    """byte offsets of the four result arrays inside one flat slab (all 8-byte aligned): [U | status | iter | X].  The
    trajectory comes LAST so that the head of the slab -- controls, status, iteration counts: 492 of the 1500 bytes per instance at
    the headline shape -- is a contiguous prefix: the payload a caller may choose to gather instead of the whole slab (X is the
    roll-out of U, `rollout_trajectory` reproduces it on the receiving side)."""
    off_u = 0
    off_s = off_u + batch * n * 8
    off_i = off_s + ((batch * 4 + 7) // 8) * 8
    off_x = off_i + batch * 2 * 4
    off_x = ((off_x + 7) // 8) * 8
    total = off_x + batch * X * 8
    return off_u, off_x, off_s, off_i, total


def head_bytes(batch, n, X):
    """bytes of the [U | status | iter] prefix of a slab"""
    return slab_layout(batch, n, X)[1]


def _views(slab, batch, n, X):
    off_u, off_x, off_s, off_i, total = slab_layout(batch, n, X)
    return dict(
        control=slab[off_u:off_u + batch * n * 8].view(torch.float64).view(batch, n),
        trajectory=slab[off_x:off_x + batch * X * 8].view(torch.float64).view(batch, X),
        status=slab[off_s:off_s + batch * 4].view(torch.int32),
        iter=slab[off_i:off_i + batch * 8].view(torch.int32).view(batch, 2),
    )


def alloc_result_slab(batch, n, X, device):
    """One flat byte buffer holding control [b,n] f64, status [b] i32, iter [b,2] i32, trajectory [b,X] f64.
    The typed views alias the slab, so the engine writes straight into what the gather sends (no packing copy)."""
    total = slab_layout(batch, n, X)[4]
    slab = torch.zeros(total, dtype=torch.uint8, device=device)
    return slab, _views(slab, batch, n, X)


def split_slab(slab, batch, n, X):
    return _views(slab, batch, n, X)


def max_shard(total, world):
    """instances of the largest shard of shard_range(total, ., world)"""
    return max(shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world))


def alloc_shard_slab(total, rank, world, n, X, device):
    """The result slab of `rank`'s shard for a batch that the ranks do NOT divide: dist.gather wants equal pieces, so every rank's slab
    has the layout of the LARGEST shard; the views a rank fills -- and shard_views() reads back on the receiving side -- cover the
    first hi - lo instances of each array (the tail of a smaller shard's arrays stays zero and travels unused: at most one instance
    per array).  -> (slab, views of this rank's instances, capacity)"""
    cap = max_shard(total, world)
    slab, views = alloc_result_slab(cap, n, X, device)
    lo, hi = shard_range(total, rank, world)
    return slab, {k: v[:hi - lo] for k, v in views.items()}, cap


def shard_views(slab, total, rank, world, n, X):
    """the instances of `rank` inside a slab laid out by alloc_shard_slab (receiving side)"""
    lo, hi = shard_range(total, rank, world)
    return {k: v[:hi - lo] for k, v in _views(slab, max_shard(total, world), n, X).items()}


def rollout_trajectory(A, B, d, x0, control):
    """X = Phi x0 + Psi U + xi as the roll-out x_{k+1} = A x_k + B u_k + d (src/LMPC.cpp:282-286) for a batch, torch tensors in natural
    indexing: A (b,nx,nx), B (b,nx,nu), d (b,nx), x0 (b,nx), control (b, N nu) -> (b, (N+1) nx).  What rank 0 runs when only
    the [U | status | iter] head of the slabs was gathered."""
    b, nx = x0.shape
    nu = B.shape[2]
    N = control.shape[1] // nu
    u = control.view(b, N, nu)
    xs = [x0]
    for k in range(N):
        xs.append(torch.bmm(A, xs[-1].unsqueeze(2)).squeeze(2) + torch.bmm(B, u[:, k].unsqueeze(2)).squeeze(2) + d)
    return torch.stack(xs, dim=1).reshape(b, (N + 1) * nx)


def alloc_gather_buffers(slab, rank, world):
    if rank != 0:
        return None
    return [torch.empty_like(slab) for _ in range(world)]


def gather_results(slab, rank, world, bufs, dst=0, force=False, nbytes=None):
    """the single collective of the path: every rank's slab -> rank `dst` (force: also with a one-rank group); nbytes: only that
    prefix of the slab travels (head_bytes: [U | status | iter])"""
    if world == 1 and not force:
        return [slab]
    if nbytes is None:
        dist.gather(slab, gather_list=bufs if rank == dst else None, dst=dst)
    else:
        dist.gather(slab[:nbytes], gather_list=[t[:nbytes] for t in bufs] if rank == dst else None, dst=dst)
    return bufs


class _HostStream:
    """stand-in for a HIP stream where there is none (gloo ranks on CPU in the tests): everything is synchronous"""

    def wait_event(self, ev):
        pass


class _HostEvent:
    def record(self, stream=None):
        pass


class GatherLoop:
    """The N > 1 step loop of bench.py: every step solves the rank's shard into result slab k and sends that slab to rank
    0 with the ONE collective of the path.  With `overlap` (two slabs) the gather of step k runs on a side stream, ordered
    after the solve that filled its slab, and overlaps the solve of step k + 1; a slab is not overwritten before its
    previous gather has finished.  `solve(views, k)` fills the four typed views of slab k (the engine writes straight
    into them); on a CUDA device it must launch on the current stream.  verify() is the end-to-end check of the path."""

    def __init__(self, slabs, rank, world, solve, device, use_dist=True, overlap=False, force_gather=False, payload_bytes=None):
        self.slabs, self.rank, self.world, self.solve = slabs, rank, world, solve
        self.use_dist, self.force = use_dist, force_gather
        self.payload_bytes = payload_bytes  # None: the whole slab [U | status | iter | X]; head_bytes(...): without X
        self.n_slabs = len(slabs)
        self.cuda = device.type == "cuda"
        self.overlap = overlap and self.n_slabs > 1
        if use_dist:
            self.gather_bufs = [alloc_gather_buffers(sl[0], rank, world) for sl in slabs]
        else:
            self.gather_bufs = [[torch.empty_like(sl[0])] for sl in slabs]  # single-GPU self-test: a device copy stands in
        if self.cuda:
            self.cur = torch.cuda.current_stream()
            self.comm = torch.cuda.Stream(device=device) if self.overlap else None
            self.ev_solved = [torch.cuda.Event() for _ in slabs]
            self.ev_sent = [torch.cuda.Event() for _ in slabs]
        else:
            self.cur, self.comm = _HostStream(), (_HostStream() if self.overlap else None)
            self.ev_solved = [_HostEvent() for _ in slabs]
            self.ev_sent = [_HostEvent() for _ in slabs]
        self.step_no = 0
        self.last = 0  # slab of the most recent step

    def send(self, k):
        if self.use_dist:
            gather_results(self.slabs[k][0], self.rank, self.world, self.gather_bufs[k], force=self.force, nbytes=self.payload_bytes)
        else:
            nb = self.payload_bytes if self.payload_bytes is not None else self.slabs[k][0].numel()
            self.gather_bufs[k][0][:nb].copy_(self.slabs[k][0][:nb], non_blocking=True)

    def step(self, communicate=True):
        k = self.step_no % self.n_slabs
        self.step_no += 1
        self.last = k
        if self.overlap and self.step_no > self.n_slabs:
            self.cur.wait_event(self.ev_sent[k])  # this slab's previous gather must be done before it is overwritten
        self.solve(self.slabs[k][1], k)
        if not communicate:
            return
        if self.overlap:
            self.ev_solved[k].record(self.cur)
            if self.cuda:
                with torch.cuda.stream(self.comm):
                    self.comm.wait_event(self.ev_solved[k])
                    self.send(k)
                    self.ev_sent[k].record(self.comm)
            else:
                self.send(k)
        else:
            self.send(k)

    def verify(self):
        """After the loop has drained: every rank checksums the slab it sent last (exact: int64 wrap-around sum of the
        raw bytes), the checksums are all-gathered, and rank 0 compares them with the checksums of what it received.
        Returns (ok, per-rank checksums) on rank 0, (True, None) elsewhere."""
        k = self.last
        nb = self.payload_bytes if self.payload_bytes is not None else self.slabs[k][0].numel()
        mine = self.slabs[k][0][:nb].view(torch.int64).sum().reshape(1)
        if not self.use_dist:
            return bool(torch.equal(self.gather_bufs[k][0][:nb], self.slabs[k][0][:nb])), [int(mine.item())]
        every = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(every, mine)
        if self.rank != 0:
            return True, None
        got = [int(b[:nb].view(torch.int64).sum().item()) for b in self.gather_bufs[k]]
        sent = [int(t.item()) for t in every]
        return got == sent, sent

    def gathered(self):
        """rank 0: the slabs received by the most recent gather, in rank order"""
        return self.gather_bufs[self.last]
