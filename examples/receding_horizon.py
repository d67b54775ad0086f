"""Closed-loop receding-horizon MPC for a batch of systems that share one model, entirely on the device.

Every tick: x0 <- the state predicted for step 1 (a slice of the trajectory the engine wrote into HBM, plus a
disturbance), copra_batch_set_x0 with that DEVICE pointer, copra_batch_solve on the current stream.  No host
synchronisation inside the loop; the factorisation of the shared model happens once, at the first solve
(shared-model fast path).

    python examples/receding_horizon.py [batch] [ticks]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from copra_amd import BatchLMPC, workloads  # noqa: E402
from copra_amd.sharding import alloc_result_slab  # noqa: E402


def run(batch=16384, ticks=50, seed=0, noise=0.01, warm=False, v_max=0.6, u_max=3.0, check=None):
    dev = torch.device("cuda:0")
    wl = workloads.com_preview(batch, v_max=v_max, u_max=u_max)
    nx, nu, N = 6, 3, wl["N"]
    eng = BatchLMPC(nx, nu, N, batch, wl["costs"], wl["cstrs"])
    eng.set_shared_system(wl["A"][0], wl["B"][0], wl["d"][0])
    if warm:  # the active set of tick k, moved one step towards the present, is tried first at tick k + 1
        eng.set_warm_start(True)
    slab, out = alloc_result_slab(batch, nu * N, nx * (N + 1), dev)
    eng.set_outputs(out["control"], out["trajectory"], out["status"], out["iter"])
    x = torch.from_numpy(np.ascontiguousarray(wl["x0"])).to(dev)
    gen = torch.Generator(device=dev).manual_seed(seed)
    stream = torch.cuda.current_stream().cuda_stream
    goal = torch.tensor(workloads.COM_X_GOAL, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters, kernel_s = 0.0, 0.0
    for tick in range(ticks):
        eng.set_x0(x)  # device pointer, used in place
        eng.solve(stream)
        if check is not None:  # (tests: compare this tick with a reference, costs a synchronisation)
            check(tick, wl, x, out)
            iters += float(out["iter"][:, 0].double().mean().item())
            kernel_s += eng.last_solve_seconds()
        # plant: the predicted next state plus a small disturbance (stays on the device, same stream)
        # (position disturbance only: a state outside the velocity bound at step 0 makes the QP infeasible, reference
        #  quirk Q5; an instance whose QP failed keeps its state instead of the NaN the engine flags failures with)
        pred = out["trajectory"][:, nx:2 * nx].clone()
        pred[:, :3] += noise * torch.randn(batch, 3, device=dev, generator=gen, dtype=torch.float64)
        x = torch.where((out["status"] == 0)[:, None], pred, x).contiguous()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = int((out["status"] == 0).sum().item())
    dist = float((x[:, :3] - goal[:3]).norm(dim=1).mean().item())
    return dict(batch=batch, ticks=ticks, seconds=dt, ticks_per_s=ticks / dt, solves_per_s=batch * ticks / dt,
                solved_last_tick=ok, mean_distance_to_goal=dist, warm_start=warm,
                mean_iterations=iters / ticks if check is not None else None,
                mean_kernel_ms=1e3 * kernel_s / ticks if check is not None else None)


if __name__ == "__main__":
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    print(run(b, k))
    print(run(b, k, warm=True))
    for warm in (False, True):  # the tight workload (3..22 active constraints per instance), with per-tick statistics
        print(run(b, k, warm=warm, v_max=0.25, u_max=1.2, check=lambda *a: None))
