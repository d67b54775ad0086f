"""Phase cycles of the first tier BEHIND the one-instance-per-lane pass (GPU box): rows of the instances the pass left over.
Stamps of lmpc_fused_ric.hpp: set-up (here: the gather of the stage records) | sweep (skipped) | row norms | roll-out | - | active set | results."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from copra_amd import BatchLMPC, workloads  # noqa: E402
from copra_amd import _capi  # noqa: E402  engine options (copra_options_t) instead of the COPRA_* environment variables of earlier rounds

# (the pass runs under the phase profile as it does in production: the rows of the instances it finishes stay zero)

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
ran, finished = eng.lane_pass_info()
nw = (b + 63) // 64
print("kernel ms", eng.last_solve_seconds() * 1e3, "| the pass finished", finished, "of", b, "| at the minimiser", int((it == 1).sum()),
      "| iteration counts", np.bincount(it)[1:8].tolist())
pw = pr[:nw]  # rows 0 .. waves-1: the pass's stamps (one row per wave)
for k, name in enumerate(("staging", "sweep", "roll-out", "verdict")):
    print("pass  %-10s mean %9.0f cycles per wave" % (name, pw[:, k].mean()))
print("pass  %-10s mean %9.0f cycles per wave" % ("total", pw[:, 7].mean()))
sel = (pr[:, 7] > 0) & (np.arange(b) >= nw)  # the first tier's rows: the instances the pass left over (the first `waves` rows belong to the pass)
print("tier instances profiled", int(sel.sum()))
names = ("gather+Acl", "(sweep)", "row norms", "roll-out", "-", "active set", "results", "total")
for k, name in enumerate(names):
    print("tier  %-12s mean %9.0f cycles" % (name, pr[sel, k].mean()))
for v in range(2, int(it.max()) + 1):
    s2 = sel & (it == v)
    if s2.any():
        print("tier  iters=%d: %6d instances, active set %8.0f, total %8.0f cycles" % (v, s2.sum(), pr[s2, 5].mean(), pr[s2, 7].mean()))
