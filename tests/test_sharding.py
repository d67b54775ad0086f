"""N > 1 path on CPU: two gloo ranks run the step loop bench.py runs with RCCL (backend "nccl") on the GPUs -- the SAME
code (copra_amd.sharding.GatherLoop) -- with the CPU oracle standing in for the per-rank solve (no GPU in this
container); under test: the partitioning, the slab layout, the single gather per step, the double-buffer rotation and
the verification of what rank 0 received."""
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


def _worker(rank, world, port, total, out_path, overlap, payload):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pyoracle
    from copra_amd import workloads
    from copra_amd.sharding import GatherLoop, alloc_result_slab, head_bytes, rollout_trajectory, shard_range, split_slab
    wl = workloads.double_integrator(total, seed=11)
    lo, hi = shard_range(total, rank, world)  # contiguous shards of ONE batch, as bench.py --gpus N does
    b = hi - lo
    n, X = 10, 22
    dev = torch.device("cpu")
    slabs = [alloc_result_slab(b, n, X, dev) for _ in range(2 if overlap else 1)]
    calls = []

    def solve_into(v, k):  # stand-in for eng.set_outputs + eng.solve: the CPU oracle fills the slab the gather sends
        x0 = wl["x0"][lo:hi] + 0.01 * len(calls)  # (every step solves something else: a stale slab would be noticed)
        ref = pyoracle.lmpc_solve_batch(wl["A"][lo:hi], wl["B"][lo:hi], wl["d"][lo:hi], x0, wl["N"], wl["costs"], wl["cstrs"])
        v["control"].copy_(torch.from_numpy(ref["control"]))
        v["trajectory"].copy_(torch.from_numpy(ref["trajectory"]))
        v["status"].copy_(torch.from_numpy(ref["status"]))
        v["iter"].copy_(torch.from_numpy(ref["iter"]))
        calls.append(k)

    loop = GatherLoop(slabs, rank, world, solve_into, dev, use_dist=True, overlap=overlap,
                      payload_bytes=head_bytes(b, n, X) if payload == "controls" else None)
    for _ in range(3):
        loop.step()
    dist.barrier()
    ok, sums = loop.verify()
    assert ok
    if rank == 0:
        assert len(sums) == world
        parts = [split_slab(g, b, n, X) for g in loop.gathered()]
        u = torch.cat([p["control"] for p in parts]).numpy()
        st = torch.cat([p["status"] for p in parts]).numpy()
        if payload == "controls":  # X did not travel: rank 0 rolls the gathered controls out (x0 of the LAST step)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
            xt = rollout_trajectory(t(wl["A"]), t(wl["B"]), t(wl["d"]), t(wl["x0"] + 0.01 * (len(calls) - 1)), torch.from_numpy(u)).numpy()
        else:
            xt = torch.cat([p["trajectory"] for p in parts]).numpy()
        np.savez(out_path, control=u, status=st, trajectory=xt, slabs_used=np.array(calls))
    # a corrupted payload must be caught
    if rank == 0:
        loop.gathered()[1].view(torch.int64)[3] += 1
    ok2, _ = loop.verify()
    if rank == 0:
        assert not ok2
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
@pytest.mark.parametrize("overlap,payload", [(False, "full"), (True, "full"), (True, "controls")])
def test_two_rank_gloo_gather(tmp_path, overlap, payload):
    """bench.py's own step loop (copra_amd.sharding.GatherLoop: step / send / verify, one or two result slabs) with two
    gloo ranks on CPU and the oracle standing in for the solve: contiguous shards, ONE gather per step, rank 0 ends up with
    every shard of the LAST step and the checksum verification notices a corrupted payload; payload "controls": only the
    [U | status | iter] head of the slabs travels and rank 0 reproduces X by the roll-out"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    from copra_amd import workloads
    total, world = 32, 2
    out = str(tmp_path / "gathered.npz")
    pyoracle.lib()
    mp.spawn(_worker, args=(world, _free_port(), total, out, overlap, payload), nprocs=world, join=True)
    got = np.load(out)
    wl = workloads.double_integrator(total, seed=11)
    ref = pyoracle.lmpc_solve_batch(wl["A"], wl["B"], wl["d"], wl["x0"] + 0.02, wl["N"], wl["costs"], wl["cstrs"])  # step 3
    assert (got["status"] == ref["status"]).all()
    assert np.array_equal(got["control"], ref["control"])
    ok = ref["status"] == 0  # (payload "controls": the trajectory is the receiving side's roll-out of the gathered controls)
    assert np.abs(got["trajectory"][ok] - ref["trajectory"][ok]).max() <= (0.0 if payload == "full" else 1e-9)
    assert list(got["slabs_used"]) == ([0, 1, 0] if overlap else [0, 0, 0])


def _worker_ragged(rank, world, port, total, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pyoracle
    from copra_amd import workloads
    from copra_amd.sharding import GatherLoop, alloc_shard_slab, head_bytes, max_shard, shard_range, shard_views
    wl = workloads.double_integrator(total, seed=13)
    lo, hi = shard_range(total, rank, world)
    n, X = 10, 22
    dev = torch.device("cpu")
    made = [alloc_shard_slab(total, rank, world, n, X, dev) for _ in range(2)]
    slabs = [(m[0], m[1]) for m in made]

    def solve_into(v, k):
        ref = pyoracle.lmpc_solve_batch(wl["A"][lo:hi], wl["B"][lo:hi], wl["d"][lo:hi], wl["x0"][lo:hi], wl["N"], wl["costs"], wl["cstrs"])
        for key in ("control", "trajectory", "status", "iter"):
            v[key].copy_(torch.from_numpy(ref[key]))

    loop = GatherLoop(slabs, rank, world, solve_into, dev, use_dist=True, overlap=True)
    for _ in range(2):
        loop.step()
    dist.barrier()
    ok, _ = loop.verify()
    assert ok
    if rank == 0:
        parts = [shard_views(g, total, r, world, n, X) for r, g in enumerate(loop.gathered())]
        assert [p["control"].shape[0] for p in parts] == [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
        np.savez(out_path, control=torch.cat([p["control"] for p in parts]).numpy(), status=torch.cat([p["status"] for p in parts]).numpy(),
                 trajectory=torch.cat([p["trajectory"] for p in parts]).numpy(), iter=torch.cat([p["iter"] for p in parts]).numpy(),
                 head=np.array(head_bytes(max_shard(total, world), n, X)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_three_rank_gloo_gather_of_a_batch_the_ranks_do_not_divide(tmp_path):
    """world = 3, batch = 1000 (shards of 333, 333 and 334 instances; round-5 verdict: dist.gather needs equal slabs and bench.py only avoided
    the case by construction): every rank's slab is laid out for the LARGEST shard (alloc_shard_slab), rank 0 takes each rank's own count
    out of what it received (shard_views) -- the whole batch in order, equal to the oracle's"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    from copra_amd import workloads
    total, world = 1000, 3
    out = str(tmp_path / "ragged.npz")
    pyoracle.lib()
    mp.spawn(_worker_ragged, args=(world, _free_port(), total, out), nprocs=world, join=True)
    got = np.load(out)
    wl = workloads.double_integrator(total, seed=13)
    ref = pyoracle.lmpc_solve_batch(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    assert got["control"].shape == (total, 10) and np.array_equal(got["status"], ref["status"]) and np.array_equal(got["iter"], ref["iter"])
    assert np.array_equal(got["control"], ref["control"]) and np.array_equal(got["trajectory"], ref["trajectory"])
