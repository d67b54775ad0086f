"""Shared-model tick (copra_batch_set_shared_system) with and without the one-instance-per-lane pass in front of the tier (GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from copra_amd import BatchLMPC, workloads  # noqa: E402
from copra_amd import _capi  # engine options (copra_options_t) instead of the COPRA_* environment variables of earlier rounds

b = 65536
for vmax, umax in ((0.6, 3.0), (0.35, 1.8)):
    wl = workloads.com_preview(b, v_max=vmax, u_max=umax)
    out = {}
    for mode in ("off", "on"):
        if mode == "off":
            _capi.OPTIONS["no_lane_pass"] = int("1")
        else:
            _capi.OPTIONS.pop("no_lane_pass", None)
        eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"])
        eng.set_shared_system(wl["A"][5], wl["B"][5], wl["d"][5])
        eng.set_x0(wl["x0"])
        ts = []
        for _ in range(12):
            eng.solve()
            eng.synchronize()
            ts.append(eng.last_solve_seconds())
        out[mode] = (eng.results(), float(np.mean(ts[6:])), eng.lane_pass_info())
        eng.close()
    r0, r1 = out["off"][0], out["on"][0]
    ok = r0["status"] == 0
    print("v_max %.2f: tick %.4f ms -> %.4f ms with the lane pass (%.0f -> %.0f M solves/s); status equal %s, iter equal %s, max |dU| %.2e, |dX| %.2e, pass %s"
          % (vmax, out["off"][1] * 1e3, out["on"][1] * 1e3, b / out["off"][1] / 1e6, b / out["on"][1] / 1e6, (r0["status"] == r1["status"]).all(),
             (r0["iter"] == r1["iter"]).all(), np.abs(r0["control"][ok] - r1["control"][ok]).max(), np.abs(r0["trajectory"][ok] - r1["trajectory"][ok]).max(),
             out["on"][2]))
