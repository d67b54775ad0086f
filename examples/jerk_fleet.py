"""Closed-loop MPC of a fleet of jerk-controlled CoM models (position, velocity, acceleration per axis; nx = 9, nu = 3: the cart-table model of
preview control) with a goal and velocity / jerk limits PER ROBOT, entirely on the device.

Every robot has its own sampling period, goal (copra_batch_set_cost_reference) and limits (copra_batch_set_constraint_rhs,
copra_batch_set_control_bounds); every tick x0 <- the state predicted for step 1 plus a disturbance.  The controller's axes are decoupled, so
the engine solves it with one (robot, axis) per lane -- nothing to declare.

    python examples/jerk_fleet.py [batch] [ticks]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from copra_amd import BatchLMPC, workloads  # noqa: E402


def run(batch=32768, ticks=40, seed=0, noise=0.002):
    dev = torch.device("cuda:0")
    wl = workloads.jerk_preview(batch, nu=3, N=20, v_max=0.5, j_max=15.0)
    nx, nu, N = 9, 3, wl["N"]
    rng = np.random.default_rng(seed)
    # the velocity limit as rows E x_k <= f, so that every robot can have its own right-hand side
    Ev = np.hstack([np.zeros((3, 3)), np.eye(3), np.zeros((3, 3))])
    cstrs = [dict(kind="trajectory", E=Ev, f=[0.5] * 3, ineq=True), wl["cstrs"][1]]
    eng = BatchLMPC(nx, nu, N, batch, wl["costs"], cstrs)
    goals = np.tile(wl["costs"][0]["p"], (batch, 1))
    goals[:, :3] += 0.3 * rng.standard_normal((batch, 3))
    eng.set_cost_reference(0, torch.from_numpy(goals).to(dev))
    eng.set_constraint_rhs(0, np.repeat((0.5 * rng.uniform(0.7, 1.3, batch))[:, None], 3, axis=1))
    jmax = 15.0 * rng.uniform(0.7, 1.3, batch)
    eng.set_control_bounds(-np.repeat(jmax[:, None], nu * N, axis=1), np.repeat(jmax[:, None], nu * N, axis=1))
    A, B, d = (torch.from_numpy(np.ascontiguousarray(np.swapaxes(wl[k], 1, 2) if wl[k].ndim == 3 else wl[k])).to(dev) for k in ("A", "B", "d"))
    x = torch.from_numpy(np.ascontiguousarray(wl["x0"])).to(dev)
    gen = torch.Generator(device=dev).manual_seed(seed)
    U = torch.empty((batch, nu * N), dtype=torch.float64, device=dev)
    X = torch.empty((batch, nx * (N + 1)), dtype=torch.float64, device=dev)
    status = torch.empty(batch, dtype=torch.int32, device=dev)
    iters = torch.empty((batch, 2), dtype=torch.int32, device=dev)
    eng.set_outputs(U, X, status, iters)  # (the results stay on the device)
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for tick in range(ticks):
        eng.set_system(A, B, d, x)
        eng.solve(stream)
        ok = (status == 0)[:, None]  # (a robot whose state violates its own velocity limit has no solution this tick: it coasts)
        x = torch.where(ok, X[:, nx:2 * nx], x) + noise * torch.randn((batch, nx), device=dev, generator=gen, dtype=torch.float64)  # the plant: step 1 + a disturbance
    torch.cuda.synchronize()
    sec = time.perf_counter() - t0
    solved = (status == 0).cpu().numpy()
    dist = float(np.abs(X[:, nx:nx + 3].cpu().numpy()[solved] - goals[solved, :3]).mean())
    out = dict(batch=batch, ticks=ticks, seconds=sec, solves_per_s=batch * ticks / sec, solved_last_tick=int((status == 0).sum().item()),
               axis_solver=bool(eng.axis_solver_ran()), lane_pass=eng.lane_pass_info(), mean_distance_to_goal=dist)
    eng.close()
    return out


if __name__ == "__main__":
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    t = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    print(run(b, t))
