"""BASELINE config 2 (double integrator nx=2, nu=1, N=10 + control bound): kernel rate (dev tool; GPU box only)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from copra_amd import BatchLMPC, workloads  # noqa: E402

for batch in (4096, 262144):
    wl = workloads.double_integrator(batch)
    eng = BatchLMPC(2, 1, wl["N"], batch, wl["costs"], wl["cstrs"])
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    ts = []
    for _ in range(8):
        eng.solve()
        eng.synchronize()
        ts.append(eng.last_solve_seconds())
    res = eng.results()
    t = float(np.mean(ts[3:]))
    print("config 2, batch %d: kernel %.3f ms -> %.1f M solves/s; ok %d, mean iters %.2f" % (
        batch, t * 1e3, batch / t / 1e6, int((res["status"] == 0).sum()), res["iter"][:, 0].mean()))
