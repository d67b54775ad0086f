"""N > 1 path on CPU: two gloo ranks shard a batch, fill their result slabs and gather them on rank 0 exactly as
bench.py does with RCCL (backend "nccl") on the GPUs.  The per-rank "solve" is the CPU oracle here (no GPU in this
container); what is under test is the partitioning, the slab layout and the single gather."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pyoracle
    from copra_amd import workloads
    from copra_amd.sharding import alloc_gather_buffers, alloc_result_slab, gather_results, shard_range, split_slab
    wl = workloads.double_integrator(total, seed=11)
    lo, hi = shard_range(total, rank, world)
    b = hi - lo
    n, X = 10, 22
    slab, v = alloc_result_slab(b, n, X, torch.device("cpu"))
    ref = pyoracle.lmpc_solve_batch(wl["A"][lo:hi], wl["B"][lo:hi], wl["d"][lo:hi], wl["x0"][lo:hi], wl["N"],
                                    wl["costs"], wl["cstrs"])
    v["control"].copy_(torch.from_numpy(ref["control"]))
    v["trajectory"].copy_(torch.from_numpy(ref["trajectory"]))
    v["status"].copy_(torch.from_numpy(ref["status"]))
    v["iter"].copy_(torch.from_numpy(ref["iter"]))
    bufs = alloc_gather_buffers(slab, rank, world)
    got = gather_results(slab, rank, world, bufs)
    if rank == 0:
        parts = [split_slab(g, b, n, X) for g in got]
        u = torch.cat([p["control"] for p in parts]).numpy()
        st = torch.cat([p["status"] for p in parts]).numpy()
        np.savez(out_path, control=u, status=st)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    from copra_amd.sharding import shard_range
    for total in (1, 7, 64, 65536, 262144):
        for world in (1, 2, 3, 8):
            edges = [shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            for a, b in zip(edges, edges[1:]):
                assert a[1] == b[0]


@pytest.mark.timeout(300)
def test_two_rank_gloo_gather(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    from copra_amd import workloads
    total, world = 32, 2  # equal shards (weak-scaling layout of bench.py)
    out = str(tmp_path / "gathered.npz")
    pyoracle.lib()
    mp.spawn(_worker, args=(world, _free_port(), total, out), nprocs=world, join=True)
    got = np.load(out)
    wl = workloads.double_integrator(total, seed=11)
    ref = pyoracle.lmpc_solve_batch(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    assert (got["status"] == ref["status"]).all()
    assert np.array_equal(got["control"], ref["control"])
