"""Prototype (numpy) of the stage-wise Riccati interior-point solver for long-horizon LMPC / InitialStateLMPC.
Design study for copra_amd/csrc/lmpc_riccati.hpp; compares against the oracle (and the config-5 truth vectors)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

INF = np.inf
BIG = 1e300


def stage_plan(nx, nu, N, costs, cstrs):
    """cost rows / constraint rows per stage from the plain-dict problem description"""
    nz = nx + nu
    X, U = nx * (N + 1), nu * N
    W = np.zeros((N + 1, nz, nz))
    q = np.zeros((N + 1, nz))
    for k in range(N):
        W[k, nx:, nx:] += 1e-6 * np.eye(nu)  # LMPC.cpp:228-230

    def add_cost_row(k, a, p, w):
        W[k] += w * np.outer(a, a)
        q[k] -= w * p * a

    def split_full(row, cols_per_step, steps):
        nzs = np.nonzero(row)[0]
        if nzs.size == 0:
            return 0, np.zeros(cols_per_step)
        s = nzs[0] // cols_per_step
        if (nzs // cols_per_step != s).any():
            raise NotImplementedError("full-size row couples several steps")
        return s, row[s * cols_per_step:(s + 1) * cols_per_step]

    for c in costs:
        p = np.atleast_1d(np.asarray(c["p"], float))
        r = p.shape[0]
        w = c.get("weights")
        w = np.ones(r) if w is None else np.atleast_1d(np.asarray(w, float))
        if w.shape[0] != r:
            w = np.tile(w, r // w.shape[0])
        M = np.atleast_2d(np.asarray(c["M"], float)) if c.get("M") is not None else None
        Nm = np.atleast_2d(np.asarray(c["N"], float)) if c.get("N") is not None else None
        kind = c["kind"]
        full = (M is not None and M.shape[1] == X and X != nx) or (Nm is not None and Nm.shape[1] == U and U != nu)
        if not full:
            steps = {"trajectory": range(N + 1), "target": [N], "control": range(N), "mixed": range(N)}[kind]
            for k in steps:
                for i in range(r):
                    a = np.zeros(nz)
                    if M is not None and kind != "control":
                        a[:nx] = M[i]
                    if Nm is not None and kind in ("control", "mixed"):
                        a[nx:] = Nm[i]
                    add_cost_row(k, a, p[i], w[i])
        else:
            for i in range(r):
                a = np.zeros(nz)
                ks = None
                if kind != "control":
                    ks, a[:nx] = split_full(M[i], nx, N + 1)
                if kind in ("control", "mixed"):
                    ku, a[nx:] = split_full(Nm[i], nu, N)
                    if ks is not None and ks != ku and np.any(a[:nx]) and np.any(a[nx:]):
                        raise NotImplementedError("full-size mixed row couples steps")
                    ks = ku if ks is None or not np.any(a[:nx]) else ks
                add_cost_row(ks, a, p[i], w[i])
    rows = [[] for _ in range(N + 1)]  # (a, f, is_ineq)

    def add_row(k, a, f, ineq):
        rows[k].append((a, f, ineq))

    lb = np.full(U, -BIG)
    ub = np.full(U, BIG)
    for c in cstrs:
        kind = c["kind"]
        ineq = c.get("ineq", True)
        if kind == "control_bound":
            lo, up = np.atleast_1d(np.asarray(c["lower"], float)), np.atleast_1d(np.asarray(c["upper"], float))
            lb = np.tile(lo, N) if lo.shape[0] == nu else lo
            ub = np.tile(up, N) if up.shape[0] == nu else up
            continue
        if kind == "trajectory_bound":
            lo, up = np.atleast_1d(np.asarray(c["lower"], float)), np.atleast_1d(np.asarray(c["upper"], float))
            fullb = lo.shape[0] == X and X != nx
            for bound in (lo, up):  # quirk Q1: lower rows are  x <= lower  too
                for line in range(bound.shape[0] if fullb else nx * (N + 1)):
                    v = bound[line] if fullb else bound[line % nx]
                    if np.isinf(v):
                        continue
                    a = np.zeros(nz)
                    a[line % nx] = 1.0
                    add_row(line // nx, a, v, True)
            continue
        E = np.atleast_2d(np.asarray(c["E"], float)) if c.get("E") is not None else None
        G = np.atleast_2d(np.asarray(c["G"], float)) if c.get("G") is not None else None
        f = np.atleast_1d(np.asarray(c["f"], float))
        r = f.shape[0]
        full = (E is not None and E.shape[1] == X and X != nx) or (G is not None and G.shape[1] == U and U != nu)
        if not full:
            steps = {"trajectory": range(N + 1), "control": range(N), "mixed": range(N)}[kind]
            for k in steps:
                for i in range(r):
                    a = np.zeros(nz)
                    if E is not None and kind != "control":
                        a[:nx] = E[i]
                    if G is not None and kind != "trajectory":
                        a[nx:] = G[i]
                    add_row(k, a, f[i], ineq)
        else:
            for i in range(r):
                a = np.zeros(nz)
                ks = None
                if kind != "control":
                    ks, a[:nx] = split_full(E[i], nx, N + 1)
                if kind != "trajectory":
                    ku, a[nx:] = split_full(G[i], nu, N)
                    ks = ku if ks is None or not np.any(a[:nx]) else ks
                add_row(ks, a, f[i], ineq)
    for k in range(N):
        for j in range(nu):
            if ub[k * nu + j] < BIG and not np.isinf(ub[k * nu + j]):
                a = np.zeros(nz)
                a[nx + j] = 1.0
                add_row(k, a, ub[k * nu + j], True)
            if lb[k * nu + j] > -BIG and not np.isinf(lb[k * nu + j]):
                a = np.zeros(nz)
                a[nx + j] = -1.0
                add_row(k, a, -lb[k * nu + j], True)
    return W, q, rows


def riccati(A, B, H, g, x0_free, H0_extra=None, g0_extra=None):
    """min sum_k 1/2 dz_k' H_k dz_k + g_k' dz_k  s.t. dx+ = A dx + B du (dx_0 = 0 unless x0_free)"""
    N = H.shape[0] - 1
    nx, nu = B.shape
    P = H[N][:nx, :nx].copy()
    p = g[N][:nx].copy()
    Ks, ks = [None] * N, [None] * N
    AB = np.hstack([A, B])
    for k in range(N - 1, -1, -1):
        M = H[k] + AB.T @ P @ AB
        h = g[k] + AB.T @ p
        Muu, Mux, Mxx = M[nx:, nx:], M[nx:, :nx], M[:nx, :nx]
        L = np.linalg.cholesky(Muu)
        Ks[k] = -np.linalg.solve(L.T, np.linalg.solve(L, Mux))
        ks[k] = -np.linalg.solve(L.T, np.linalg.solve(L, h[nx:]))
        P = Mxx + Mux.T @ Ks[k]
        P = 0.5 * (P + P.T)
        p = h[:nx] + Mux.T @ ks[k]
    dz = np.zeros((N + 1, nx + nu))
    if x0_free:
        P0 = P + (H0_extra if H0_extra is not None else 0.0)
        p0 = p + (g0_extra if g0_extra is not None else 0.0)
        dx = -np.linalg.solve(P0, p0)
    else:
        dx = np.zeros(nx)
    for k in range(N):
        du = Ks[k] @ dx + ks[k]
        dz[k, :nx], dz[k, nx:] = dx, du
        dx = A @ dx + B @ du
    dz[N, :nx] = dx
    return dz, P


def solve(A, B, d, x0, N, costs, cstrs, initial_state=None, max_iter=80, verbose=False, delta=1e-9):
    A, B, d, x0 = (np.asarray(v, float) for v in (A, B, d, x0))
    nx, nu = B.shape
    nz = nx + nu
    W, q, rows = stage_plan(nx, nu, N, costs, cstrs)
    x0_free = initial_state is not None
    if x0_free:
        lo, up = np.asarray(initial_state["x0lb"], float), np.asarray(initial_state["x0ub"], float)
        for j in range(nx):
            a = np.zeros(nz)
            a[j] = 1.0
            rows[0].append((a, up[j], True))
            rows[0].append((-a, -lo[j], True))
    # flatten rows
    stage_of, Arows, frows, isineq = [], [], [], []
    for k in range(N + 1):
        for (a, f, ineq) in rows[k]:
            stage_of.append(k)
            Arows.append(a)
            frows.append(f)
            isineq.append(ineq)
    stage_of = np.array(stage_of, int)
    Arows = np.array(Arows).reshape(-1, nz)
    frows = np.array(frows, float)
    isineq = np.array(isineq, bool)
    m = len(frows)
    mi = int(isineq.sum())

    def rollout(x0v, u):
        z = np.zeros((N + 1, nz))
        x = x0v.copy()
        for k in range(N):
            z[k, :nx], z[k, nx:] = x, u[k]
            x = A @ x + B @ u[k] + d
        z[N, :nx] = x
        return z

    H0_extra = g0_extra = None
    if x0_free:
        # J_ref = J_stage + 1/2 x0'(R - P0) x0 + (r - g0)'x0 : P0 = unconstrained cost-to-go Hessian, g0 = dJ_stage/dx0 at 0
        _, P0unc = riccati(A, B, W, np.zeros((N + 1, nz)), False)
        z00 = rollout(np.zeros(nx), np.zeros((N, nu)))
        lam = W[N][:nx, :nx] @ z00[N, :nx] + q[N][:nx]
        for k in range(N - 1, -1, -1):
            gk = W[k] @ z00[k] + q[k]
            lam = gk[:nx] + A.T @ lam
        H0_extra = np.asarray(initial_state["R"], float) - P0unc
        g0_extra = np.asarray(initial_state["r"], float) - lam
    # ---- start: u = 0 clipped into its box rows is not known here; x0 clipped into its bounds
    xs = np.clip(x0, lo, up) if x0_free else x0
    z = rollout(xs, np.zeros((N, nu)))
    res = frows - np.einsum("ij,ij->i", Arows, z[stage_of])
    s = np.where(isineq, np.maximum(res, 1.0), 0.0)
    lam = np.where(isineq, 1.0, 0.0)
    nu_eq = np.zeros(m)
    it = 0
    for it in range(1, max_iter + 1):
        az = np.einsum("ij,ij->i", Arows, z[stage_of])
        rp = np.where(isineq, az + s - frows, 0.0)
        re = np.where(~isineq, az - frows, 0.0)
        mu = float(s[isineq] @ lam[isineq] / max(mi, 1))
        D = np.where(isineq, lam / np.where(isineq, s, 1.0), 1.0 / delta)
        H = W.copy()
        np.add.at(H, stage_of, D[:, None, None] * Arows[:, :, None] * Arows[:, None, :])
        gbase = np.einsum("kij,kj->ki", W, z) + q
        if x0_free:
            gbase[0, :nx] += H0_extra @ z[0, :nx] + g0_extra

        def newton(sig_mu, corr):
            coef = np.where(isineq, (sig_mu - corr) / np.where(isineq, s, 1.0) + D * rp, nu_eq + re / delta)
            g = gbase.copy()
            np.add.at(g, stage_of, coef[:, None] * Arows)
            dz, _ = riccati(A, B, H, g, x0_free, H0_extra, (H0_extra @ np.zeros(nx)) if x0_free else None)
            adz = np.einsum("ij,ij->i", Arows, dz[stage_of])
            ds = np.where(isineq, -rp - adz, 0.0)
            dl = np.where(isineq, (sig_mu - corr - lam * s - lam * ds) / np.where(isineq, s, 1.0), 0.0)
            return dz, ds, dl, adz

        def amax(v, dv):
            neg = isineq & (dv < 0)
            return min(1.0, float((-v[neg] / dv[neg]).min())) if neg.any() else 1.0

        dz, ds, dl, _ = newton(0.0, 0.0)
        a_aff = min(amax(s, ds), amax(lam, dl))
        mu_aff = float((s + a_aff * ds)[isineq] @ (lam + a_aff * dl)[isineq] / max(mi, 1))
        sig = (mu_aff / mu) ** 3 if mu > 0 else 0.0
        dz, ds, dl, adz = newton(sig * mu, ds * dl)
        a = min(amax(s, ds), amax(lam, dl))
        tau = 0.995 if mu > 1e-10 else 0.9999
        a = min(1.0, tau * a) if a < 1.0 else 1.0
        step = np.abs(a * dz).max()
        z_new_u = z[:N, nx:] + a * dz[:N, nx:]
        x0n = z[0, :nx] + a * dz[0, :nx]
        z = rollout(x0n, z_new_u)
        s = s + a * ds
        lam = lam + a * dl
        nu_eq = np.where(~isineq, nu_eq + (re + a * adz) / delta, 0.0)
        if verbose:
            print("%2d mu %.2e rp %.2e re %.2e step %.2e alpha %.3f sig %.1e" % (it, mu, np.abs(rp).max() if m else 0, np.abs(re).max() if m else 0, step, a, sig))
        if mu < 1e-12 and (np.abs(rp).max() if m else 0) < 1e-9 and (np.abs(re).max() if m else 0) < 1e-9 and step < 1e-9 * (1 + np.abs(z).max()):
            break
    U = z[:N, nx:].reshape(-1)
    Xt = z[:, :nx].reshape(-1)
    return dict(control=U, trajectory=Xt, x0_opt=z[0, :nx].copy(), iter=it, mu=mu)


def main():
    import pyoracle
    from copra_amd import workloads
    import fixtures as F
    T = np.load(os.path.join(ROOT, "tests", "golden", "config5_truth.npz"))
    for rdiag in (1e-6, 1e-2):
        wl = workloads.long_horizon_initial_state(6, R_diag=rdiag)
        ist = wl["initial_state"]
        for k in (0, 2, 5):
            io = dict(R=ist["R"], r=ist["r"], x0lb=ist["x0lb"][k], x0ub=ist["x0ub"][k])
            args = (wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], wl["costs"], wl["cstrs"])
            t = time.time()
            r = solve(*args, initial_state=io, verbose=(k == 0 and rdiag == 1e-6))
            t = time.time() - t
            ro = pyoracle.lmpc_solve(*args, initial_state=io)
            eu = np.abs(r["control"] - ro["control"]) / (1 + np.abs(ro["control"]))
            msg = "R=%g inst %d: ipm iters %d (%.1fs) vs oracle: U %.3e X %.3e x0 %.2e" % (
                rdiag, k, r["iter"], t, eu.max(), np.abs(r["trajectory"] - ro["trajectory"]).max(),
                np.abs(r["x0_opt"] - ro["x0_opt"]).max())
            if rdiag == 1e-6:
                ut = T["control_%d" % k]
                msg += "   vs TRUTH: U %.3e X %.3e" % ((np.abs(r["control"] - ut) / (1 + np.abs(ut))).max(),
                                                     np.abs(r["trajectory"] - T["trajectory_%d" % k]).max())
            print(msg, flush=True)
    # LMPC fixtures at N = 300 and the nine-class problem
    for system in ("bounded", "ineq", "mixed", "eq"):
        for xcost in ("target", "trajectory", "mixed"):
            pb = getattr(F, system + "_system")(xcost, N=300)
            args = (pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"])
            try:
                r = solve(*args)
            except NotImplementedError as e:
                print(system, xcost, "not stage-wise:", e)
                continue
            ro = pyoracle.lmpc_solve(*args)
            eu = np.abs(r["control"] - ro["control"]) / (1 + np.abs(ro["control"]))
            print("%s/%s N=300: ipm iters %d, oracle status %d iters %s: U %.3e X %.3e" % (
                system, xcost, r["iter"], ro["status"], ro["iter"], eu.max(),
                (np.abs(r["trajectory"] - ro["trajectory"]) / (1 + np.abs(ro["trajectory"]))).max()), flush=True)
    for full in (False, True):
        for isl in (False, True):
            pb = F.nine_class_problem(150) if not full else F.initial_state_problem(True)
            ist = dict(R=10.0 * np.eye(2), r=np.array([0.1, -0.2]), x0lb=pb["x0"] - 0.05, x0ub=pb["x0"] + 0.05) if isl else None
            args = (pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"])
            try:
                r = solve(*args, initial_state=ist)
            except NotImplementedError as e:
                print("nine classes full=%s: not stage-wise: %s" % (full, e))
                continue
            ro = pyoracle.lmpc_solve(*args, initial_state=ist)
            eu = np.abs(r["control"] - ro["control"]) / (1 + np.abs(ro["control"]))
            print("nine classes full=%s is=%s N=%d: ipm iters %d, oracle %d %s: U %.3e" % (full, isl, pb["N"], r["iter"], ro["status"], ro["iter"], eu.max()), flush=True)


if __name__ == "__main__":
    main()
