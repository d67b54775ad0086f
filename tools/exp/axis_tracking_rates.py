import sys
import numpy as np, torch
from copra_amd import BatchLMPC, workloads
from copra_amd.autospan import autospan_cost
b = 65536
wl = workloads.com_preview(b)
c0 = wl["costs"][0]
ts_ref = np.linspace(0.0, 1.0, wl["N"] + 1)
xref = workloads.COM_X_INIT[None, :] + ts_ref[:, None] * (workloads.COM_X_GOAL - workloads.COM_X_INIT)[None, :]
track_costs = [autospan_cost(dict(c0, p=xref.reshape(-1))), wl["costs"][1]]
own = np.tile(xref.reshape(-1), (b, 1)) + 0.02 * np.random.default_rng(7).standard_normal((b, xref.size))
for opts in (None, dict(no_axis_solver=1)):
  for refs in (False, True):
    eng = BatchLMPC(6, 3, wl["N"], b, track_costs, wl["cstrs"], options=opts)
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    if refs: eng.set_cost_reference(0, torch.from_numpy(own).cuda())
    for _ in range(8): eng.solve()
    eng.synchronize()
    ts = []
    for _ in range(20):
        eng.solve(); eng.synchronize(); ts.append(eng.last_solve_seconds())
    print(opts, "own trajectories" if refs else "one trajectory", "%.1f M solves/s, %.4f ms" % (b / np.median(ts) / 1e6, np.median(ts) * 1e3), eng.axis_solver_ran(), eng.lane_pass_info())
    eng.close()
# ... and a goal per instance (a constant reference)
goals = workloads.COM_X_GOAL[None, :] + 0.05 * np.random.default_rng(5).standard_normal((b, 6))
for opts in (None, dict(no_axis_solver=1)):
    eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"], options=opts)
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    eng.set_cost_reference(0, torch.from_numpy(np.ascontiguousarray(goals)).cuda())
    for _ in range(8): eng.solve()
    eng.synchronize()
    ts = []
    for _ in range(20):
        eng.solve(); eng.synchronize(); ts.append(eng.last_solve_seconds())
    print(opts, "own goals", "%.1f M solves/s, %.4f ms" % (b / np.median(ts) / 1e6, np.median(ts) * 1e3), eng.axis_solver_ran(), eng.lane_pass_info())
    eng.close()
