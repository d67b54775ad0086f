"""Closed-loop TRACKING of a moving reference trajectory for a batch of different systems, entirely on the device.

The reference's API writes a reference that changes along the horizon as a full-size TrajectoryCost (M = blkdiag(M0 .. M0), stacked p,
costFunctions.cpp:63-82) and moves it by replacing the cost object.  Here the controller is built once; every tick
  * the window of the reference that the horizon sees goes to the controller that exists: one copy for the whole batch
    (copra_batch_set_cost_reference_all; `per_instance=True`: every instance follows its own reference, a device tensor used in place),
  * x0 <- the state predicted for step 1 plus a disturbance, copra_batch_set_system with the device pointers, copra_batch_solve.
The full-size cost is recognised as a per-step cost with the reference of the step and runs on the headline's kernels.

With `shared_model=True` the batch is a fleet of IDENTICAL plants (copra_batch_set_shared_system): the factorisation is done once, every
instance adds the delta of its own reference to it.

    python examples/tracking.py [batch] [ticks]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from copra_amd import BatchLMPC, workloads  # noqa: E402
from copra_amd.sharding import alloc_result_slab  # noqa: E402


def reference_window(tick, N, T=0.117, speed=0.4):
    """positions on a circle of radius 5 cm around x_init at the height of x_goal, velocities to match: rows k = 0 .. N of the window"""
    t = T * (tick + np.arange(N + 1))
    c = workloads.COM_X_INIT[:3] + np.array([0.0, 0.0, workloads.COM_X_GOAL[2] - workloads.COM_X_INIT[2]])
    r, w = 0.05, speed
    pos = c[None, :] + r * np.stack([np.cos(w * t), np.sin(w * t), 0.0 * t], axis=1)
    vel = r * w * np.stack([-np.sin(w * t), np.cos(w * t), 0.0 * t], axis=1)
    return np.hstack([pos, vel])


def run(batch=32768, ticks=50, seed=0, noise=0.002, per_instance=False, shared_model=False):
    dev = torch.device("cuda:0")
    wl = workloads.com_preview(batch, v_max=0.6, u_max=3.0)
    nx, nu, N = 6, 3, wl["N"]
    track = dict(kind="trajectory", M=np.kron(np.eye(N + 1), np.eye(nx)), p=reference_window(0, N).reshape(-1),
                 weights=np.tile([10.0, 10.0, 10.0, 1.0, 1.0, 1.0], N + 1))
    eng = BatchLMPC(nx, nu, N, batch, [track, wl["costs"][1]], wl["cstrs"])
    slab, out = alloc_result_slab(batch, nu * N, nx * (N + 1), dev)
    eng.set_outputs(out["control"], out["trajectory"], out["status"], out["iter"])
    A, B, d = (torch.from_numpy(np.ascontiguousarray(np.swapaxes(wl[k], 1, 2) if wl[k].ndim == 3 else wl[k])).to(dev) for k in ("A", "B", "d"))
    x = torch.from_numpy(np.ascontiguousarray(wl["x0"])).to(dev)
    gen = torch.Generator(device=dev).manual_seed(seed)
    stream = torch.cuda.current_stream().cuda_stream
    if shared_model:
        eng.set_shared_system(wl["A"][0], wl["B"][0], wl["d"][0])
    own = torch.empty((batch, nx * (N + 1)), dtype=torch.float64, device=dev) if per_instance else None
    phase = torch.rand(batch, 1, device=dev, generator=gen, dtype=torch.float64) * 0.01 if per_instance else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    err = 0.0
    for tick in range(ticks):
        ref = torch.from_numpy(reference_window(tick, N).reshape(-1)).to(dev)
        if per_instance:  # every instance its own reference: the shared window plus its own offset in height, written in place
            own.copy_(ref[None, :].expand(batch, -1))
            own[:, 2::nx] += phase
            eng.set_cost_reference(0, own)
        else:
            eng.set_cost_reference(0, ref)  # one new reference for every instance (1-D: copra_batch_set_cost_reference_all)
        if shared_model:  # a fleet of identical plants: ONE model for the batch (copra_batch_set_shared_system, above), only the states move;
            eng.set_x0(x)  # per-instance references ride on the batch-wide factor (the delta sweep of the shared lane pass)
        else:
            eng.set_system(A, B, d, x)  # device tensors (column-major A, B), used in place
        eng.solve(stream)
        pred = out["trajectory"][:, nx:2 * nx].clone()
        pred[:, :3] += noise * torch.randn(batch, 3, device=dev, generator=gen, dtype=torch.float64)
        x = torch.where((out["status"] == 0)[:, None], pred, x).contiguous()
        if tick == ticks - 1:
            err = float((x[:, :3] - ref[nx:nx + 3][None, :]).norm(dim=1).mean().item())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(batch=batch, ticks=ticks, per_instance_references=per_instance, shared_model=shared_model, seconds=dt, solves_per_s=batch * ticks / dt,
                solved_last_tick=int((out["status"] == 0).sum().item()), lane_pass=eng.lane_pass_info(), mean_position_error_last_tick=err)


if __name__ == "__main__":
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    run(b, 5)  # (module load, LDS opt-in, first-solve set-up: outside the figures below)
    print(run(b, k))
    print(run(b, k, per_instance=True))
    print(run(b, k, per_instance=True, shared_model=True))  # (a fleet of identical plants, every one on its own path)
