#!/usr/bin/env python3
"""Which side is off?  Device and oracle against the certified extended-precision optimum (tests/truth.py), entry-wise (floor 1e-3),
on the workloads of the parity suite whose Hessians are ill conditioned.  Prints one line per case."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch  # noqa: F401
import pyoracle
import truth
from copra_amd import BatchLMPC, workloads


def report(tag, res, ref, wl, ist=None, x0o=None, ks=None):
    du = dx = ou = ox = 0.0
    for k in ks if ks is not None else range(len(res["status"])):
        io = None if ist is None else dict(R=ist["R"], r=ist["r"], x0lb=ist["x0lb"][k], x0ub=ist["x0ub"][k])
        zg = ref["control"][k] if ist is None else np.concatenate([ref["x0_opt"][k], ref["control"][k]])
        t = truth.solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], wl["costs"], wl["cstrs"], zg, initial_state=io)
        du, dx = max(du, truth.rel(res["control"][k], t["control"])), max(dx, truth.rel(res["trajectory"][k], t["trajectory"]))
        ou, ox = max(ou, truth.rel(ref["control"][k], t["control"])), max(ox, truth.rel(ref["trajectory"][k], t["trajectory"]))
    print("%-46s device U %.2e X %.2e | oracle U %.2e X %.2e" % (tag, du, dx, ou, ox), flush=True)


for vm, um in ((0.6, 3.0), (0.25, 1.2)):
    wl = workloads.com_preview(1024, v_max=vm, u_max=um)
    eng = BatchLMPC(6, 3, wl["N"], 1024, wl["costs"], wl["cstrs"])
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    eng.solve()
    res = eng.results()
    ref = pyoracle.lmpc_solve_batch(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"], nthreads=8)
    report("headline vmax %.2f (1024)" % vm, res, ref, wl)
    eng.close()

for r_diag in (1e-2, 1e-6):
    b = 6
    wl = workloads.long_horizon_initial_state(b, R_diag=r_diag)
    ist = wl["initial_state"]
    ros = [pyoracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], wl["costs"], wl["cstrs"],
                               initial_state=dict(R=ist["R"], r=ist["r"], x0lb=ist["x0lb"][k], x0ub=ist["x0ub"][k])) for k in range(b)]
    ref = dict(control=np.array([r["control"] for r in ros]), trajectory=np.array([r["trajectory"] for r in ros]),
               x0_opt=np.array([r["x0_opt"] for r in ros]))
    for solver in ("default", "quadprog_dense"):
        eng = BatchLMPC(12, 6, wl["N"], b, wl["costs"], wl["cstrs"], initial_state=dict(R=ist["R"], r=ist["r"]))
        eng.select_solver(solver)
        eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
        eng.set_initial_state_bounds(ist["x0lb"], ist["x0ub"])
        eng.solve()
        res = eng.results()
        report("config 5 R=%g %s" % (r_diag, solver), res, ref, wl, ist=ist)
        eng.close()
