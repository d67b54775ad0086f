"""Differential fuzz of this round's plan-builder classifications and of the lane pass (GPU box): random controllers on the (6, 3) and
(2, 1) systems with random mixes of constraint forms; the default build of the plan (selection rows, step rows, lane pass) against the
previous classification (COPRA_NO_SELECTION_ROWS, COPRA_NO_STEP_ROWS, COPRA_NO_LANE_PASS, COPRA_NO_STAGE_REFS) -- statuses, iteration
counts, U.  Half of the controllers track a reference TRAJECTORY (a full-size TrajectoryCost and / or ControlCost with repeating blocks),
controller-wide or one per instance."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from copra_amd import BatchLMPC, workloads  # noqa: E402
from copra_amd import _capi  # engine options (copra_options_t) instead of the COPRA_* environment variables of earlier rounds

OLD = ("no_selection_rows", "no_step_rows", "no_lane_pass", "no_stage_refs")
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
    costs = list(wl["costs"])
    own_ref = None
    if rng.random() < 0.5:  # a reference trajectory: the state cost as a full-size entry with a stacked reference
        c0 = costs[0]
        M0 = np.atleast_2d(c0["M"])
        r = M0.shape[0]
        pk = np.asarray(c0["p"])[None, :] * np.linspace(0.6, 1.0, N + 1)[:, None] + 0.01 * rng.standard_normal((N + 1, r))
        costs[0] = dict(kind="trajectory", M=np.kron(np.eye(N + 1), M0), p=pk.reshape(-1), weights=np.tile(np.asarray(c0["weights"], dtype=float), N + 1))
        forms.append("xref")
        if rng.random() < 0.5:
            own_ref = np.tile(pk.reshape(-1), (b, 1)) + 0.01 * rng.standard_normal((b, pk.size))
            forms.append("own")
        if rng.random() < 0.4:
            c1 = costs[1]
            uk = 0.1 * rng.standard_normal((N, nu))
            costs[1] = dict(kind="control", N=np.kron(np.eye(N), np.atleast_2d(c1["N"])), p=uk.reshape(-1),
                            weights=np.tile(np.asarray(c1["weights"], dtype=float), N))
            forms.append("uref")
    if com and rng.random() < 0.25:  # a MixedCost reference trajectory (velocity + a share of the control)
        M0, N0 = np.hstack([np.zeros((3, 3)), np.eye(3)]), 0.05 * np.eye(3)
        pm = 0.05 * rng.standard_normal((N, 3))
        costs.append(dict(kind="mixed", M=np.hstack([np.kron(np.eye(N), M0), np.zeros((3 * N, nx))]), N=np.kron(np.eye(N), N0), p=pm.reshape(-1),
                          weights=np.tile([2.0, 3.0, 1.5], N)))
        forms.append("mref")
    out = {}
    for mode in ("old", "new"):
        opts = dict({e: 1 for e in OLD} if mode == "old" else {}, lane_min_batch=1)
        eng = BatchLMPC(nx, nu, N, b, costs, cstrs, options=opts)
        eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
        if own_ref is not None:
            eng.set_cost_reference(0, own_ref)
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
