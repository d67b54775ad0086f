"""BASELINE configs[1]: the double integrator (2, 1, N = 10) with a control bound, at 4096 and 262 144 instances -- which kernels take it, rate,
parity sample"""
import sys
import numpy as np
sys.path.insert(0, "oracle")
from copra_amd import BatchLMPC, workloads
import pyoracle
for b in (4096, 262144):
    wl = workloads.double_integrator(b)
    for opts in (None, dict(no_axis_solver=1)):
        eng = BatchLMPC(2, 1, wl["N"], b, wl["costs"], wl["cstrs"], options=opts)
        eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
        for _ in range(8): eng.solve()
        eng.synchronize()
        ts = []
        for _ in range(20):
            eng.solve(); eng.synchronize(); ts.append(eng.last_solve_seconds())
        r = eng.results()
        worst = 0.0
        for k in range(0, b, b // 16):
            ro = pyoracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], wl["costs"], wl["cstrs"])
            assert r["status"][k] == ro["status"] and tuple(r["iter"][k]) == tuple(ro["iter"]), (k, r["iter"][k], ro["iter"])
            worst = max(worst, np.abs(r["control"][k] - ro["control"]).max() / max(np.abs(ro["control"]).max(), 1e-3))
        print("batch %6d %-22s %8.1f M solves/s (%.4f ms) axis solver %s lane info %s layout %s mean iters %.2f worst rel %.1e" % (b, opts, b / np.median(ts) / 1e6, np.median(ts) * 1e3, eng.axis_solver_ran(), eng.lane_pass_info(), eng.layout_info().get("lds_bytes"), r["iter"][:, 0].mean(), worst))
        eng.close()
