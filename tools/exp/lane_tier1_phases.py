"""Phase cycles of the first tier BEHIND the one-instance-per-lane pass (GPU box): rows of the instances the pass left over.
Stamps of lmpc_fused_ric.hpp: set-up (here: the gather of the stage records) | sweep (skipped) | row norms | roll-out | - | active set | results."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from copra_amd import BatchLMPC, workloads  # noqa: E402
from copra_amd import _capi  # noqa: E402  engine options (copra_options_t) instead of the COPRA_* environment variables of earlier rounds

_capi.OPTIONS["lane_dbg"] = 8

b = 65536
wl = workloads.com_preview(b)
eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"])
eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
for _ in range(3):
    eng.solve()
eng.enable_phase_profile(True)
eng.solve()
eng.solve()
pr = eng.phase_profile()
res = eng.results()
it = res["iter"][:, 0]
sel = (it >= 2) & (np.arange(b) >= 1024)
print("kernel ms", eng.last_solve_seconds() * 1e3, "instances left over", (it >= 2).sum())
names = ("gather+Acl", "(sweep)", "row norms", "roll-out", "-", "active set", "results", "total")
for k, name in enumerate(names):
    print("%-12s mean %9.0f cycles" % (name, pr[sel, k].mean()))
for v in range(2, int(it.max()) + 1):
    s2 = sel & (it == v)
    if s2.any():
        print("iters=%d: %6d instances, active set %8.0f, total %8.0f cycles" % (v, s2.sum(), pr[s2, 5].mean(), pr[s2, 7].mean()))
