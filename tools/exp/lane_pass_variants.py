"""Cost of the one-instance-per-lane pass by parts (GPU box): a workload where EVERY instance ends in it (loose bounds), so the solve
is the pass + an empty first tier; COPRA_LANE_DBG switches parts of it off (results are then wrong: timing only)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from copra_amd import BatchLMPC, workloads  # noqa: E402
from copra_amd import _capi  # engine options (copra_options_t) instead of the COPRA_* environment variables of earlier rounds

b = 65536
wl = workloads.com_preview(b, v_max=50.0, u_max=500.0)
_capi.OPTIONS["lane_keep"] = int("1")
for dbg in [int(v) for v in os.environ.get("VARIANTS", "0,1,2,3").split(",")]:
    _capi.OPTIONS["lane_dbg"] = int(str(dbg))
    eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"])
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    ts = []
    for _ in range(12):
        eng.solve()
        eng.synchronize()
        ts.append(eng.last_solve_seconds())
    res = eng.results()
    print("dbg %d: solve %.4f ms (best %.4f), finished at the minimiser %d of %d" % (dbg, np.mean(ts[6:]) * 1e3, np.min(ts) * 1e3, (res["iter"][:, 0] == 1).sum(), b))
    eng.close()
# phase stamps of the pass (lane 0 of every wave): staging | sweep | roll-out | verdict
for pdbg in os.environ.get("PHASE_DBG", "8").split(","):
  _capi.OPTIONS["lane_dbg"] = int(pdbg)
  print("phase stamps with COPRA_LANE_DBG =", pdbg)
  if True:
    eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"])
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    for _ in range(4):
        eng.solve()
    eng.enable_phase_profile(True)
    eng.solve()
    pr = eng.phase_profile()[: b // 64]
    print("cycles per wave: staging %.0f, sweep %.0f (per stage %.0f), roll-out %.0f (per stage %.0f), verdict %.0f, total %.0f"
          % (pr[:, 0].mean(), pr[:, 1].mean(), pr[:, 1].mean() / 20, pr[:, 2].mean(), pr[:, 2].mean() / 21, pr[:, 3].mean(), pr[:, 7].mean()))

