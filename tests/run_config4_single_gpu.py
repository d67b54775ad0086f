#!/usr/bin/env python3
"""BASELINE.json configs[3] on ONE GPU: the seed-2 batch of 262144 CoM preview systems as its eight consecutive shards of 32768,
each through the step loop the ranks of `bench.py --gpus 8` run (copra_amd.sharding.GatherLoop: solve into the slab, ONE gather of
the slab to rank 0) with a one-rank RCCL process group standing in for the eight-rank one (this pool has one GPU per box).
Per shard: the checksum verification of what rank 0 gathered + 16 instances against the CPU oracle (U, X, status, both iteration
counters); over all 262144 instances: the size-independent properties (every instance solved, bounds, roll-out identity).
Prints ONE JSON line; run by tests/test_gpu_parity.py::test_config4_seed2_batch_as_eight_shards_on_one_gpu in a child process
(the RCCL group lives and dies with it)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rel(a, b, floor=1e-3):
    import numpy as np
    return float(np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    import pyoracle
    from copra_amd import BatchLMPC, workloads
    from copra_amd.sharding import GatherLoop, alloc_result_slab, shard_range, split_slab

    world_of_config, per = 8, 32768
    total = world_of_config * per
    nx, nu, N = 6, 3, 20
    n, X = nu * N, nx * (N + 1)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group(backend="nccl", device_id=dev)  # RCCL, one rank

    wl = workloads.com_preview(total, N=N, seed=2)
    eng = BatchLMPC(nx, nu, N, per, wl["costs"], wl["cstrs"])
    slabs = [alloc_result_slab(per, n, X, dev) for _ in range(2)]
    stream = torch.cuda.current_stream().cuda_stream

    def solve_into(v, k):
        eng.set_outputs(v["control"], v["trajectory"], v["status"], v["iter"])
        eng.solve(stream)

    loop = GatherLoop(slabs, 0, 1, solve_into, dev, use_dist=True, overlap=True, force_gather=True)
    up = np.array(wl["cstrs"][1]["upper"])
    vmax = np.array(wl["cstrs"][0]["upper"])[3:]
    out = {"shards": [], "instances": total, "seed": 2}
    worst_u = worst_x = worst_roll = 0.0
    solved = 0
    hist = np.zeros(16, dtype=np.int64)
    dev_s = 0.0
    for g in range(world_of_config):
        lo, hi = shard_range(total, g, world_of_config)
        A, B, d, x0 = wl["A"][lo:hi], wl["B"][lo:hi], wl["d"][lo:hi], wl["x0"][lo:hi]
        eng.set_system(A, B, d, x0)
        loop.step()
        torch.cuda.synchronize()
        dev_s += eng.last_solve_seconds()
        ok_sum, sums = loop.verify()
        part = split_slab(loop.gathered()[0], per, n, X)
        u = part["control"].cpu().numpy()
        x = part["trajectory"].cpu().numpy()
        st = part["status"].cpu().numpy()
        it = part["iter"].cpu().numpy()
        # the oracle on 16 instances of the shard, evenly spaced, plus the shard's hardest instance
        pick = np.union1d(np.linspace(0, per - 1, 16).astype(int), [int(np.argmax(it[:, 0]))])
        ref = pyoracle.lmpc_solve_batch(A[pick], B[pick], d[pick], x0[pick], N, wl["costs"], wl["cstrs"])
        okm = ref["status"] == 0
        eu, ex = rel(u[pick][okm], ref["control"][okm]), rel(x[pick][okm], ref["trajectory"][okm])
        worst_u, worst_x = max(worst_u, eu), max(worst_x, ex)
        # size-independent properties on the whole shard
        uu, xx = u.reshape(per, N, nu), x.reshape(per, N + 1, nx)
        xr = np.einsum("bij,bkj->bki", A, xx[:, :-1]) + np.einsum("bij,bkj->bki", B, uu) + d[:, None, :]
        roll = float(np.abs(xr - xx[:, 1:]).max())
        worst_roll = max(worst_roll, roll)
        solved += int((st == 0).sum())
        hist += np.bincount(np.minimum(it[:, 0], 15), minlength=16)
        out["shards"].append({
            "shard": g, "range": [lo, hi], "gathered_matches_checksum": bool(ok_sum),
            "oracle_instances": int(len(pick)), "status_agree": bool((st[pick] == ref["status"]).all()),
            "iterations_agree": bool((it[pick][okm] == ref["iter"][okm]).all()), "max_rel_u_err": eu, "max_rel_x_err": ex,
            "all_solved": bool((st == 0).all()),
            "control_bounds_hold": bool((uu <= up + 1e-6).all() and (uu >= -up - 1e-6).all()),
            "velocity_bounds_hold": bool((xx[:, :, 3:] <= vmax + 1e-6).all()),
            "x0_is_first_state": bool(np.abs(xx[:, 0] - x0).max() <= 1e-12), "rollout_residual": roll,
            "lane_pass": list(eng.lane_pass_info()), "device_ms": eng.last_solve_seconds() * 1e3})
    out.update({"solved_ok": solved, "max_rel_u_err": worst_u, "max_rel_x_err": worst_x, "max_rollout_residual": worst_roll,
                "active_set_iteration_histogram": {str(k): int(v) for k, v in enumerate(hist) if v},
                "device_ms_all_shards": dev_s * 1e3, "rccl_world_size": dist.get_world_size()})
    eng.close()
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)  # RCCL's banner goes through C stdio: keep the JSON line last
    except Exception:
        pass
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    t0 = time.time()
    main()
