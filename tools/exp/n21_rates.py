import sys
sys.path.insert(0, "oracle")
import numpy as np
import pyoracle
from copra_amd import BatchLMPC, workloads
b = 65536
for N in (20, 21):
    wl = workloads.com_preview(b, N=N)
    for opts in (None, dict(no_axis_solver=1)):
        eng = BatchLMPC(6, 3, N, b, wl["costs"], wl["cstrs"], options=opts)
        eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
        for _ in range(8): eng.solve()
        eng.synchronize()
        ts = []
        for _ in range(20):
            eng.solve(); eng.synchronize(); ts.append(eng.last_solve_seconds())
        r = eng.results()
        bad = 0
        for k in range(0, b, 4099):
            ro = pyoracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], N, wl["costs"], wl["cstrs"])
            bad += int(r["status"][k] != ro["status"] or tuple(r["iter"][k]) != tuple(ro["iter"]) or np.abs(r["control"][k] - ro["control"]).max() > 1e-6 * max(1e-3, np.abs(ro["control"]).max()))
        print("N = %d %-24s %7.1f M solves/s (%.4f ms) axis %s %s sample mismatches %d" % (N, opts, b / np.median(ts) / 1e6, np.median(ts) * 1e3, eng.axis_solver_ran(), eng.lane_pass_info(), bad))
        eng.close()
