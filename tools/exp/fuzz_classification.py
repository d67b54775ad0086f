"""Differential fuzz of this round's plan-builder classifications and of the lane pass (GPU box): random controllers on the (6, 3) and
(2, 1) systems with random mixes of constraint forms; the default build of the plan (selection rows, step rows, lane pass) against the
previous classification (COPRA_NO_SELECTION_ROWS, COPRA_NO_STEP_ROWS, COPRA_NO_LANE_PASS) -- statuses, iteration counts, U."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from copra_amd import BatchLMPC, workloads  # noqa: E402

OLD = ("COPRA_NO_SELECTION_ROWS", "COPRA_NO_STEP_ROWS", "COPRA_NO_LANE_PASS")
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
bad = 0
for seed in range(nseeds):
    rng = np.random.default_rng(1000 + seed)
    com = rng.random() < 0.6
    N = int(rng.integers(5, 21))
    b = int(rng.choice([4096, 8192, 24576]))
    wl = workloads.com_preview(b, N=N, seed=seed) if com else workloads.double_integrator(b, N=N, seed=seed)
    nx, nu = (6, 3) if com else (2, 1)
    X, U = nx * (N + 1), nu * N
    cstrs = []
    forms = []
    if rng.random() < 0.7:
        cstrs.append(wl["cstrs"][-1])  # the workload's control bound
        forms.append("ubound")
    if com and rng.random() < 0.5:
        cstrs.append(wl["cstrs"][0])
        forms.append("xbound")
    vsel = np.zeros((nu, nx))
    vsel[np.arange(nu), nx - nu + np.arange(nu)] = 1.0  # the velocity components
    lim = 0.7 if com else 8.0
    if rng.random() < 0.5:
        cstrs.append(dict(kind="trajectory", E=np.vstack([vsel, -vsel]), f=[lim] * (2 * nu), ineq=True))
        forms.append("selection+-")
    if rng.random() < 0.3:
        cstrs.append(dict(kind="trajectory", E=rng.standard_normal((1, nx)), f=[6.0 if com else 40.0], ineq=True))
        forms.append("dense-x")
    if rng.random() < 0.3:
        cstrs.append(dict(kind="control", G=rng.standard_normal((1, nu)), f=[4.0 if com else 300.0], ineq=True))
        forms.append("dense-u")
    if rng.random() < 0.3:
        cstrs.append(dict(kind="mixed", E=0.3 * rng.standard_normal((1, nx)), G=rng.standard_normal((1, nu)), f=[5.0 if com else 300.0], ineq=True))
        forms.append("mixed")
    if rng.random() < 0.5:  # terminal limit as a full-size E
        E = np.zeros((2 * nu, X))
        E[:nu, X - nu:] = np.eye(nu)
        E[nu:, X - nu:] = -np.eye(nu)
        cstrs.append(dict(kind="trajectory", E=E, f=[0.4 if com else 6.0] * (2 * nu), ineq=True))
        forms.append("terminal-full")
    if rng.random() < 0.3:  # a full-size control row inside one step
        G = np.zeros((1, U))
        k = int(rng.integers(0, N))
        G[0, k * nu:(k + 1) * nu] = rng.standard_normal(nu)
        cstrs.append(dict(kind="control", G=G, f=[3.0 if com else 250.0], ineq=True))
        forms.append("u-full-1step")
    if not cstrs:
        cstrs = [wl["cstrs"][-1]]
        forms = ["ubound"]
    out = {}
    for mode in ("old", "new"):
        for e in OLD:
            os.environ.pop(e, None)
            if mode == "old":
                os.environ[e] = "1"
        os.environ["COPRA_LANE_MIN_BATCH"] = "1"
        eng = BatchLMPC(nx, nu, N, b, wl["costs"], cstrs)
        eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
        eng.solve()
        eng.solve()
        out[mode] = (eng.results(), eng.lane_pass_info(), eng.layout_info())
        eng.close()
    r0, r1 = out["old"][0], out["new"][0]
    ok = r0["status"] == 0
    same = bool((r0["status"] == r1["status"]).all() and (r0["iter"][ok] == r1["iter"][ok]).all())
    du = float(np.abs(r0["control"][ok] - r1["control"][ok]).max() / max(np.abs(r0["control"][ok]).max(), 1e-2)) if ok.any() else 0.0
    flag = "ok " if same and du <= 1e-9 else "BAD"
    bad += flag == "BAD"
    print("%s seed %2d %s N=%2d b=%5d %-60s solved %5.1f %%  lane %s  lds %d -> %d  dU %.1e"
          % (flag, seed, "CoM" if com else "DI ", N, b, "+".join(forms), 100.0 * ok.mean(), out["new"][1], out["old"][2]["lds_bytes"], out["new"][2]["lds_bytes"], du))
print("mismatches:", bad)
