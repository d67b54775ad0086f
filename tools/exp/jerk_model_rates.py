"""The jerk-controlled CoM model (chains of three states per control: nx = 3 nu) on the (instance, axis)-per-lane solver against the general
one-wave kernels it ran on before (option no_axis_solver)"""
import numpy as np
from copra_amd import BatchLMPC, workloads
b = 65536
for nu, N, vm, jm in ((3, 20, 0.6, 20.0), (3, 20, 0.3, 6.0), (3, 12, 0.3, 6.0), (2, 20, 0.3, 6.0)):
    wl = workloads.jerk_preview(b, nu=nu, N=N, v_max=vm, j_max=jm)
    for opts in (None, dict(no_axis_solver=1)):
        eng = BatchLMPC(3 * nu, nu, N, b, wl["costs"], wl["cstrs"], options=opts)
        eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
        for _ in range(8): eng.solve()
        eng.synchronize()
        ts = []
        for _ in range(10):
            eng.solve(); eng.synchronize(); ts.append(eng.last_solve_seconds())
        r = eng.results()
        print("v_max %.1f" % vm, "(%d, %d, N = %d) %-24s %8.1f M solves/s (%.4f ms) axis solver %s %s mean iterations %.2f solved %d" % (3 * nu, nu, N, opts, b / np.median(ts) / 1e6, np.median(ts) * 1e3, eng.axis_solver_ran(), eng.lane_pass_info(), r["iter"][:, 0].mean(), int((r["status"] == 0).sum())))
        eng.close()

# ... and ONE state per control: a velocity-controlled point (the kinematic model of mobile-robot MPC)
for nu, N in ((3, 20), (2, 20)):
    wl = workloads.kinematic_preview(b, nu=nu, N=N)
    for opts in (None, dict(no_axis_solver=1)):
        eng = BatchLMPC(nu, nu, N, b, wl["costs"], wl["cstrs"], options=opts)
        eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
        for _ in range(8): eng.solve()
        eng.synchronize()
        ts = []
        for _ in range(10):
            eng.solve(); eng.synchronize(); ts.append(eng.last_solve_seconds())
        r = eng.results()
        print("kinematic (%d, %d, N = %d) %-24s %8.1f M solves/s (%.4f ms) axis solver %s %s mean iterations %.2f solved %d" % (nu, nu, N, opts, b / np.median(ts) / 1e6, np.median(ts) * 1e3, eng.axis_solver_ran(), eng.lane_pass_info(), r["iter"][:, 0].mean(), int((r["status"] == 0).sum())))
        eng.close()
