"""A velocity limit written as TrajectoryConstraint(E = selection, f) instead of TrajectoryBoundConstraint (GPU box): the plan builder now
recognises rows that select one component, so the controller keeps the compact variant of the Riccati-factor tier and the lane pass's
hand-over.  COPRA_OPTIONS=no_selection_rows=1: the previous classification (dense rows, general variant)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from copra_amd import BatchLMPC, workloads  # noqa: E402
from copra_amd import _capi  # engine options (copra_options_t) instead of the COPRA_* environment variables of earlier rounds

b = 65536
wl = workloads.com_preview(b)
Ev = np.hstack([np.zeros((3, 3)), np.eye(3)])
cstrs = [dict(kind="trajectory", E=Ev, f=[0.6] * 3, ineq=True), wl["cstrs"][1]]
out = {}
for mode in ("dense rows", "selection rows"):
    if mode == "dense rows":
        _capi.OPTIONS["no_selection_rows"] = int("1")
    else:
        _capi.OPTIONS.pop("no_selection_rows", None)
    eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], cstrs)
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    ts = []
    for _ in range(12):
        eng.solve()
        eng.synchronize()
        ts.append(eng.last_solve_seconds())
    out[mode] = (eng.results(), float(np.mean(ts[6:])), eng.layout_info(), eng.lane_pass_info())
    eng.close()
r0, r1 = out["dense rows"][0], out["selection rows"][0]
ok = r0["status"] == 0
for mode in out:
    print("%-15s %.4f ms (%.1f M solves/s), layout %s, lane pass %s" % (mode, out[mode][1] * 1e3, b / out[mode][1] / 1e6, out[mode][2], out[mode][3]))
print("status equal", (r0["status"] == r1["status"]).all(), "iter equal", (r0["iter"] == r1["iter"]).all(), "max |dU|", np.abs(r0["control"][ok] - r1["control"][ok]).max())
