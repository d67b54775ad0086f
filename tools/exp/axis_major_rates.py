"""The headline's controller with its states in axis-major order (x = (p_x, v_x, p_y, v_y, p_z, v_z)): the order is seen from the first system"""
import numpy as np
from copra_amd import BatchLMPC, workloads
b = 65536
base = workloads.com_preview(b)
for name, wl in (("(p, v)", base), ("axis-major", workloads.axis_major(base))):
    for opts in (None, dict(no_axis_solver=1)):
        eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"], options=opts)
        eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
        for _ in range(8): eng.solve()
        eng.synchronize()
        ts = []
        for _ in range(20):
            eng.solve(); eng.synchronize(); ts.append(eng.last_solve_seconds())
        print("%-12s %-24s %7.1f M solves/s (%.4f ms) axis solver %s %s" % (name, opts, b / np.median(ts) / 1e6, np.median(ts) * 1e3, eng.axis_solver_ran(), eng.lane_pass_info()))
        eng.close()
