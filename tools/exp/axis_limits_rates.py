"""every robot its own velocity and actuator limits: the (instance, axis)-per-lane solver against the round-5 pair"""
import numpy as np
from copra_amd import BatchLMPC, workloads
b = 65536
wl = workloads.com_preview(b)
N = wl["N"]
rng = np.random.default_rng(6)
vlim = 0.6 * rng.uniform(0.8, 1.3, b)
ulim = 3.0 * rng.uniform(0.8, 1.3, b)
Ev = np.hstack([np.zeros((3, 3)), np.eye(3)])
cstrs = [dict(kind="trajectory", E=Ev, f=[0.6] * 3, ineq=True), wl["cstrs"][1]]
for opts in (None, dict(no_axis_solver=1)):
    eng = BatchLMPC(6, 3, N, b, wl["costs"], cstrs, options=opts)
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    eng.set_constraint_rhs(0, np.repeat(vlim[:, None], 3, axis=1))
    eng.set_control_bounds(-np.repeat(ulim[:, None], 3 * N, axis=1), np.repeat(ulim[:, None], 3 * N, axis=1))
    for _ in range(8): eng.solve()
    eng.synchronize()
    ts = []
    for _ in range(20):
        eng.solve(); eng.synchronize(); ts.append(eng.last_solve_seconds())
    print(opts, "own limits %.1f M solves/s, %.4f ms" % (b / np.median(ts) / 1e6, np.median(ts) * 1e3), eng.axis_solver_ran(), eng.lane_pass_info())
    eng.close()
