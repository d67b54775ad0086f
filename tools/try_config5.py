"""Solve a few instances of the BASELINE config-5 workload on the GPU and compare with the oracle (dev tool)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import torch  # noqa: F401,E402
import pyoracle  # noqa: E402
from copra_amd import BatchLMPC, workloads  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ncheck = int(sys.argv[2]) if len(sys.argv) > 2 else 4
solver = sys.argv[3] if len(sys.argv) > 3 else "default"  # default | quadprog_dense | riccati_ipm
wl = workloads.long_horizon_initial_state(batch)
ist = wl["initial_state"]
eng = BatchLMPC(12, 6, wl["N"], batch, wl["costs"], wl["cstrs"], initial_state=dict(R=ist["R"], r=ist["r"]))
eng.select_solver(solver)
print("solver:", eng.solver())
eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
eng.set_initial_state_bounds(ist["x0lb"], ist["x0ub"])
for rep in range(3):
    t = time.time()
    eng.solve()
    res = eng.results()
    dt = time.time() - t
    print("batch %d: %.4f s wall, kernel %.4f s -> %.0f solves/s" % (batch, dt, eng.last_solve_seconds(),
                                                                    batch / eng.last_solve_seconds()))
st, it = res["status"], res["iter"]
print("status counts", np.bincount(st, minlength=5), "iters mean %.1f max %d drops mean %.1f" % (
    it[:, 0].mean(), it[:, 0].max(), it[:, 1].mean()))
for k in range(ncheck):
    t = time.time()
    ro = pyoracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], wl["costs"], wl["cstrs"],
                             initial_state=dict(R=ist["R"], r=ist["r"], x0lb=ist["x0lb"][k], x0ub=ist["x0ub"][k]))
    dt = time.time() - t
    err = np.abs(res["control"][k] - ro["control"]).max() / max(1.0, np.abs(ro["control"]).max())
    print("inst %d: status %d/%d iter %s/%s  rel u err %.2e  oracle %.3f s" % (k, st[k], ro["status"], it[k],
                                                                                 ro["iter"], err, dt))
