"""Shared-model mode of the Riccati-factor tier on the headline shape with general rows next to the bounds (random_controllers.py:
com_preview_with_general_rows -- dense state rows, a mixed row, a control row), device against the oracle on a sample and against
lmpc_shared.hpp (option no_ric_shared) on the whole batch.      python tests/fuzz/fuzz_shared_general_rows.py first count [batch [integrators]]
With `integrators`: the controllers of random_controllers.py: make_integrator instead (double integrators in one to three dimensions, random
horizon, reference trajectories, every kind of row) -- the tier's run-time-horizon builds in shared-model mode."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as oracle  # noqa: E402
import random_controllers as RC  # noqa: E402
from copra_amd import BatchLMPC  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3))) if a.size else 0.0


first, count = int(sys.argv[1]), int(sys.argv[2])
b = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
ns = 32
bad = 0
ninst = 0
npass = 0
ran = []
integrators = len(sys.argv) > 4 and sys.argv[4] in ("integrators", "integrators-refs")
own_refs = len(sys.argv) > 4 and sys.argv[4] == "integrators-refs"  # every instance its own reference for EVERY cost (goals and trajectories)
for seed in range(first, first + count):
    if integrators:
        wl = RC.make_integrator(seed, b)
        cstrs = wl["cstrs"]
    else:
        wl, cstrs = RC.com_preview_with_general_rows(seed, b)
    k = seed % wl["A"].shape[0]
    A, B, d = wl["A"][k], wl["B"][k], wl["d"][k]
    nx, nu = A.shape[0], B.shape[1]
    out = []
    for opts in (None, dict(no_ric_shared=1)):
        eng = BatchLMPC(nx, nu, wl["N"], b, wl["costs"], cstrs, options=opts)
        eng.set_shared_system(A, B, d)
        eng.set_x0(wl["x0"])
        own = {}
        if own_refs:
            rng = np.random.default_rng([seed, 9])
            for t, cost in enumerate(wl["costs"]):
                p0 = np.asarray(cost["p"], dtype=float)
                own[t] = p0[None, :] + 0.05 * rng.standard_normal((b, p0.size))
                eng.set_cost_reference(t, own[t])
        eng.solve()
        out.append(eng.results())
        ran.append(eng.lane_pass_info()[0])
        eng.close()
    r1, r2 = out
    if own_refs:
        rr = [oracle.lmpc_solve(A, B, d, wl["x0"][k], wl["N"], [dict(cost, p=own[t][k]) for t, cost in enumerate(wl["costs"])], cstrs) for k in range(ns)]
        ref = dict(status=np.array([r["status"] for r in rr]), iter=np.array([r["iter"] for r in rr]), control=np.array([r["control"] for r in rr]))
        npass += int(ran[-2])
    else:
        ref = oracle.lmpc_solve_batch(np.tile(A, (ns, 1, 1)), np.tile(B, (ns, 1, 1)), np.tile(d, (ns, 1)), wl["x0"][:ns], wl["N"], wl["costs"], cstrs, nthreads=8)
    ok = (ref["status"] == 0) & (r1["status"][:ns] == 0)
    st = int((r1["status"][:ns] != ref["status"]).sum())
    itd = int((r1["iter"][:ns][ok] != ref["iter"][ok]).any(axis=1).sum())
    ru = rel(r1["control"][:ns][ok], ref["control"][ok])
    good = (r1["status"] == 0) & (r2["status"] == 0)
    st2 = int((r1["status"] != r2["status"]).sum())
    ru2 = rel(r1["control"][good], r2["control"][good])
    ninst += b
    if st or st2 or ru > 1e-6 or ru2 > 1e-6 or itd:
        bad += 1
        print(seed, (nx, nu, wl["N"], wl["forms"]) if integrators else ("dense state rows", "mixed row", "control row")[seed % 3], "oracle: status differ %d iter differ %d relU %.1e | lmpc_shared.hpp: status differ %d relU %.1e"
              % (st, itd, ru, st2, ru2), "mean iterations %.1f" % ref["iter"][:, 0].mean(), "  <<<<<<" if (st or st2 or ru > 1e-4 or ru2 > 1e-4) else "", flush=True)
print("seeds %d..%d: %d mismatching controllers of %d (%d instances)%s" % (first, first + count - 1, bad, count, ninst,
                                                                          "; the shared lane pass (delta sweep) ran on %d" % npass if own_refs else ""))
