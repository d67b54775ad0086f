"""The headline workload with its three axes COUPLED (one small entry of A in every instance): the one-instance-per-lane pass runs its dense
sweep and roll-out (lmpc_lane.hpp: `axes` false) -- the figure for systems that are not decoupled axis by axis.  GPU box."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from copra_amd import BatchLMPC, workloads  # noqa: E402

b = 65536
wl = workloads.com_preview(b)
for name, eps in (("decoupled axes (the CoM model as it is)", 0.0), ("coupled (A[0, 4] = 1e-3 in every instance)", 1e-3)):
    A = wl["A"].copy()
    A[:, 0, 4] = eps
    eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"])
    eng.set_system(A, wl["B"], wl["d"], wl["x0"])
    ts = []
    for _ in range(12):
        eng.solve()
        eng.synchronize()
        ts.append(eng.last_solve_seconds())
    t = float(np.mean(ts[4:]))
    print("%-45s %.4f ms per solve, %.1f M solves/s, the pass finished %d" % (name, 1e3 * t, b / t / 1e6, eng.lane_pass_info()[1]))
    eng.close()
