"""A terminal constraint written the reference's way -- a full-size E that is non-zero in the last state only (GPU box): classified as
a per-step row of step N (COPRA_OPTIONS=no_step_rows=1: as a full-size row, the previous behaviour)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from copra_amd import BatchLMPC, workloads  # noqa: E402
from copra_amd import _capi  # engine options (copra_options_t) instead of the COPRA_* environment variables of earlier rounds

b = 65536
wl = workloads.com_preview(b)
N, nx = wl["N"], 6
X = nx * (N + 1)
E = np.zeros((6, X))
E[:3, X - 3:] = np.eye(3)
E[3:, X - 3:] = -np.eye(3)
cstrs = wl["cstrs"] + [dict(kind="trajectory", E=E, f=[0.3] * 6, ineq=True)]  # |v_N| <= 0.3
out = {}
for mode in ("full-size rows", "step rows"):
    if mode == "full-size rows":
        _capi.OPTIONS["no_step_rows"] = int("1")
    else:
        _capi.OPTIONS.pop("no_step_rows", None)
    eng = BatchLMPC(6, 3, N, b, wl["costs"], cstrs)
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    ts = []
    for _ in range(12):
        eng.solve()
        eng.synchronize()
        ts.append(eng.last_solve_seconds())
    out[mode] = (eng.results(), float(np.mean(ts[6:])), eng.layout_info(), eng.lane_pass_info())
    eng.close()
r0, r1 = out["full-size rows"][0], out["step rows"][0]
ok = r0["status"] == 0
for mode in out:
    print("%-15s %.4f ms (%.1f M solves/s), layout %s, lane pass %s" % (mode, out[mode][1] * 1e3, b / out[mode][1] / 1e6, out[mode][2], out[mode][3]))
print("status equal", (r0["status"] == r1["status"]).all(), "iter equal", (r0["iter"] == r1["iter"]).all(), "max |dU|", np.abs(r0["control"][ok] - r1["control"][ok]).max(),
      "mean iterations", r1["iter"][:, 0].mean())
