"""One-instance-per-lane pass (lmpc_lane.hpp) on the headline workload: results against the first tier alone, kernel times (GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from copra_amd import _capi  # noqa: E402

if "--lib" in sys.argv:  # an experimental build of the library (copra_amd/csrc/variants/*.so)
    _capi.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
    _capi.build_library = lambda force=False: False
    del sys.argv[sys.argv.index("--lib"):sys.argv.index("--lib") + 2]
from copra_amd import BatchLMPC, workloads  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for vmax, umax in ((0.6, 3.0), (0.4, 2.0), (0.25, 1.2)):
    wl = workloads.com_preview(b, v_max=vmax, u_max=umax)
    out = {}
    for mode in ("off", "on"):
        if mode == "off":
            _capi.OPTIONS["no_lane_pass"] = 1
        else:
            _capi.OPTIONS.pop("no_lane_pass", None)
        eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"])
        eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
        ts = []
        for _ in range(12):
            eng.solve()
            eng.synchronize()
            ts.append(eng.last_solve_seconds())
        res = eng.results()
        out[mode] = (res, float(np.mean(ts[6:])), float(np.min(ts)))
        eng.close()
    r0, r1 = out["off"][0], out["on"][0]
    ok = r0["status"] == 0
    print("v_max %.2f: first tier alone %.4f ms (best %.4f), with the lane pass %.4f ms (best %.4f): %.1f -> %.1f M solves/s; status equal %s, iter equal %s, "
          "max |dU| %.2e, max |dX| %.2e, finished at the minimiser %d of %d"
          % (vmax, out["off"][1] * 1e3, out["off"][2] * 1e3, out["on"][1] * 1e3, out["on"][2] * 1e3, b / out["off"][1] / 1e6, b / out["on"][1] / 1e6,
             (r0["status"] == r1["status"]).all(), (r0["iter"] == r1["iter"]).all(),
             np.abs(r0["control"][ok] - r1["control"][ok]).max(), np.abs(r0["trajectory"][ok] - r1["trajectory"][ok]).max(),
             (r1["iter"][:, 0] == 1).sum(), b))
