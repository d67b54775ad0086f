import sys
sys.path.insert(0, "tests")
import numpy as np
import test_gpu_parity as T
from copra_amd import BatchLMPC
for N in (10, 16, 24, 30):
    b = 65536
    wl = T._planar_integrator(b, N)
    for opts in (None, dict(no_axis_solver=1)):
        eng = BatchLMPC(4, 2, N, b, wl["costs"], wl["cstrs"], options=opts)
        eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
        for _ in range(6): eng.solve()
        eng.synchronize()
        ts = []
        for _ in range(6):
            eng.solve(); eng.synchronize(); ts.append(eng.last_solve_seconds())
        r = eng.results()
        print("planar N=%d %s: %.2f M solves/s (%.3f ms) axis %s info %s lanes %d layout %s mean iters %.2f drops %.2f max adds %d solved %d" % (N, opts, b / np.median(ts) / 1e6, np.median(ts) * 1e3, eng.axis_solver_ran(), eng.lane_pass_info(), eng.lanes_per_instance(), eng.layout_info(), r["iter"][:, 0].mean(), r["iter"][:, 1].mean(), r["iter"][:, 0].max(), int((r["status"] == 0).sum())))
        eng.close()
