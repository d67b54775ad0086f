"""Riccati-factor tier: shapes x (pure bound rows | general rows) x (Q1 in registers | Q1 in LDS at a pinned ladder level), device against
the oracle -- which instantiations are healthy?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as oracle  # noqa: E402
from copra_amd import BatchLMPC  # noqa: E402


def integrator(dim, N, b, seed, general):
    rng = np.random.default_rng(seed)
    nx, nu = 2 * dim, dim
    T = rng.uniform(0.08, 0.15, b)
    I = np.eye(dim)
    A = np.zeros((b, nx, nx)); B = np.zeros((b, nx, nu))
    A[:, :dim, :dim] = I; A[:, dim:, dim:] = I; A[:, :dim, dim:] = T[:, None, None] * I
    B[:, :dim, :] = (0.5 * T * T)[:, None, None] * I; B[:, dim:, :] = T[:, None, None] * I
    d = np.zeros((b, nx))
    x0 = np.zeros((b, nx)); x0[:, :dim] = 0.3 * rng.standard_normal((b, dim)); x0[:, dim:] = rng.uniform(-0.2, 0.2, (b, dim))
    goal = np.concatenate([rng.uniform(-1, 1, dim), np.zeros(dim)])
    costs = [dict(kind="trajectory", M=np.eye(nx), p=goal, weights=[10.0] * dim + [1.0] * dim), dict(kind="control", N=np.eye(nu), p=np.zeros(nu), weights=[1e-2] * nu)]
    inf = np.inf
    cstrs = [dict(kind="trajectory_bound", lower=[-inf] * nx, upper=[inf] * dim + [0.3] * dim), dict(kind="control_bound", lower=[-1.5] * nu, upper=[1.5] * nu)]
    if general:
        vsel = np.hstack([np.zeros((dim, dim)), np.eye(dim)])
        cstrs.append(dict(kind="trajectory", E=-vsel, f=[0.3] * dim, ineq=True))
    return nx, nu, A, B, d, x0, costs, cstrs


b = 4096
shapes = ((2, 8), (2, 17), (2, 30), (3, 8), (3, 13), (3, 21), (1, 48)) if len(sys.argv) < 2 else ((2, 8), (1, 48))
for dim, N in shapes:
    for general in (False, True):
        nx, nu, A, B, d, x0, costs, cstrs = integrator(dim, N, b, 5, general)
        pick = np.linspace(0, b - 1, 128).astype(int)
        ref = oracle.lmpc_solve_batch(A[pick], B[pick], d[pick], x0[pick], N, costs, cstrs, nthreads=8)
        ok = ref["status"] == 0
        out = []
        for opts in (dict(no_ladder=1), dict(ric_k=10, no_ladder=1), dict(ric_k=7, no_ladder=1), dict(ric_k=7, no_ladder=1, ric_general=1)):
            try:
                eng = BatchLMPC(nx, nu, N, b, costs, cstrs, options=opts)
            except Exception as ex:  # noqa: BLE001
                out.append("%s: %s" % (opts, str(ex)[:40]))
                continue
            eng.set_system(A, B, d, x0)
            eng.solve()
            res = eng.results()
            info = eng.layout_info()
            eng.close()
            st = int((res["status"][pick] != ref["status"]).sum())
            both = ok & (res["status"][pick] == 0)
            itd = int((res["iter"][pick][both] != ref["iter"][both]).any(axis=1).sum())
            ru = float(np.max(np.abs(res["control"][pick][both] - ref["control"][both]) / np.maximum(np.abs(ref["control"][both]), 1e-3))) if both.any() else 0.0
            tag = "OK " if (st == 0 and itd == 0 and ru <= 1e-6) else "BAD"
            out.append("%s k=%s%s %dB/%d: %s st %d it %d relU %.0e" % ("ric" if info["factor_only"] else "sq", opts.get("ric_k", "-"), "g" if opts.get("ric_general") else "", info["lds_bytes"], info["active_capacity"], tag, st, itd, ru))
        print((nx, nu, N), "general rows" if general else "bound rows  ", "iters %.1f" % ref["iter"][:, 0].mean(), " | ".join(out), flush=True)
