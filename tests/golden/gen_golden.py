#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- SELF-GENERATED, CROSS-VALIDATED golden vectors (NOT reference-generated: the
reference cannot be built or imported in this container, see oracle/copra_oracle.h).

Two independent computations must agree before a vector is written:
  (1) an independent numpy restatement of the condensed QP, vectorised and using the CLOSED FORM
      Psi_{i,j} = A^(i-1-j) B, Phi_i = A^i, xi_i = sum_{k<i} A^k d  (not the recursion the oracle / kernels use),
      with Q = 1e-6 I + (M Psi + N)' W (M Psi + N) etc. written as whole-horizon matrix products;
  (2) the QP solved by Lawson-Hanson least-distance programming on scipy.optimize.nnls (a different algorithm family
      from Goldfarb-Idnani), then polished by an exact KKT solve on the identified active set and checked against
      the KKT conditions (stationarity, primal/dual feasibility, complementarity) to 1e-9.
The committed vectors are then used to pin BOTH the C oracle and the HIP kernels (tests/test_golden.py).
Run:  python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np
import scipy.linalg as sla
from scipy.optimize import nnls

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def preview_closed_form(A, B, d, N):
    nx, nu = B.shape
    X, U = nx * (N + 1), nu * N
    Phi = np.zeros((X, nx))
    Psi = np.zeros((X, U))
    xi = np.zeros(X)
    pw = [np.linalg.matrix_power(A, k) for k in range(N + 1)]
    for i in range(N + 1):
        Phi[i * nx:(i + 1) * nx] = pw[i]
        for k in range(i):
            xi[i * nx:(i + 1) * nx] += pw[k] @ d
        for j in range(i):
            Psi[i * nx:(i + 1) * nx, j * nu:(j + 1) * nu] = pw[i - 1 - j] @ B
    return Phi, Psi, xi


def _blockdiag(M, reps, add_cols=0):
    r, c = M.shape
    out = np.zeros((r * reps, c * (reps + add_cols)))
    for i in range(reps):
        out[i * r:(i + 1) * r, i * c:(i + 1) * c] = M
    return out


def build_qp(A, B, d, x0, N, costs, cstrs):
    """whole-horizon restatement of LMPC::updateSystem + makeQPForm (src/LMPC.cpp:225-280)"""
    A, B, d, x0 = (np.asarray(v, dtype=float) for v in (A, B, d, x0))
    nx, nu = B.shape
    X, U = nx * (N + 1), nu * N
    Phi, Psi, xi = preview_closed_form(A, B, d, N)
    xfree = Phi @ x0 + xi
    Q = 1e-6 * np.eye(U)
    c = np.zeros(U)
    for cf in costs:
        kind = cf["kind"]
        p = np.atleast_1d(np.asarray(cf["p"], dtype=float))
        w = np.ones(len(p)) if cf.get("weights") is None else np.atleast_1d(np.asarray(cf["weights"], dtype=float))
        if len(w) != len(p):
            w = np.tile(w, len(p) // len(w))
        M = None if cf.get("M") is None else np.atleast_2d(np.asarray(cf["M"], dtype=float))
        Nn = None if cf.get("N") is None else np.atleast_2d(np.asarray(cf["N"], dtype=float))
        if kind == "trajectory":
            if M.shape[1] == nx:
                M, p, w = _blockdiag(M, N + 1), np.tile(p, N + 1), np.tile(w, N + 1)
            T, off = M @ Psi, M @ xfree - p
        elif kind == "target":
            S = np.zeros((nx, X))
            S[:, N * nx:] = np.eye(nx)
            T, off = M @ S @ Psi, M @ S @ xfree - p
        elif kind == "control":
            if Nn.shape[1] == nu:
                Nn, p, w = _blockdiag(Nn, N), np.tile(p, N), np.tile(w, N)
            T, off = Nn, -p
        elif kind == "mixed":
            if M.shape[1] == nx:
                M, Nn = _blockdiag(M, N, 1), _blockdiag(Nn, N)
                p, w = np.tile(p, N), np.tile(w, N)
            T, off = M @ Psi + Nn, M @ xfree - p
        Q = Q + T.T @ (w[:, None] * T)
        c = c + T.T @ (w * off)
    Aeq, beq, Ain, bin_ = [], [], [], []
    lb, ub = np.full(U, -np.finfo(float).max), np.full(U, np.finfo(float).max)
    for cs in cstrs:
        kind = cs["kind"]
        if kind == "control_bound":
            lo, up = np.atleast_1d(cs["lower"]).astype(float), np.atleast_1d(cs["upper"]).astype(float)
            lb, ub = (np.tile(lo, N), np.tile(up, N)) if len(lo) == nu else (lo, up)
            continue
        if kind == "trajectory_bound":
            lo, up = np.atleast_1d(cs["lower"]).astype(float), np.atleast_1d(cs["upper"]).astype(float)
            if len(lo) == nx:
                lo, up = np.tile(lo, N + 1), np.tile(up, N + 1)
            rows_lo = [i for i in range(X) if lo[i] != -np.inf]
            rows_up = [i for i in range(X) if up[i] != np.inf]
            # reference quirk Q1: lower rows keep the orientation of upper rows (constraints.cpp:289-296)
            Am = np.vstack([Psi[rows_lo], Psi[rows_up]]) if rows_lo or rows_up else np.zeros((0, U))
            bm = np.concatenate([lo[rows_lo] - xfree[rows_lo], up[rows_up] - xfree[rows_up]])
            Ain.append(Am)
            bin_.append(bm)
            continue
        f = np.atleast_1d(np.asarray(cs["f"], dtype=float))
        E = None if cs.get("E") is None else np.atleast_2d(np.asarray(cs["E"], dtype=float))
        G = None if cs.get("G") is None else np.atleast_2d(np.asarray(cs["G"], dtype=float))
        if kind == "trajectory":
            if E.shape[1] == nx:
                E, f = _blockdiag(E, N + 1), np.tile(f, N + 1)
            Am, bm = E @ Psi, f - E @ xfree
        elif kind == "control":
            if G.shape[1] == nu:
                G, f = _blockdiag(G, N), np.tile(f, N)
            Am, bm = G, f
        elif kind == "mixed":
            if E.shape[1] == nx:
                E, G, f = _blockdiag(E, N, 1), _blockdiag(G, N), np.tile(f, N)
            Am, bm = E @ Psi + G, f - E @ xfree
        if cs.get("ineq", True):
            Ain.append(Am)
            bin_.append(bm)
        else:
            Aeq.append(Am)
            beq.append(bm)
    stack = lambda L, n: np.vstack(L) if L else np.zeros((0, n))
    cat = lambda L: np.concatenate(L) if L else np.zeros(0)
    return dict(Q=Q, c=c, Aeq=stack(Aeq, U), beq=cat(beq), Aineq=stack(Ain, U), bineq=cat(bin_), lb=lb, ub=ub,
                Phi=Phi, Psi=Psi, xi=xi)


def solve_qp_ldp(Q, c, Aeq, beq, Aineq, bineq, lb, ub):
    """min 1/2 x'Qx + c'x s.t. Aeq x = beq, Aineq x <= bineq, lb <= x <= ub by least-distance programming (NNLS),
    then an exact KKT polish.  Returns (x, ok)."""
    n = len(c)
    rows, rhs = [Aineq], [bineq]
    fin_u = np.isfinite(ub) & (ub < 1e300)
    fin_l = np.isfinite(lb) & (lb > -1e300)
    I = np.eye(n)
    rows += [I[fin_u], -I[fin_l]]
    rhs += [ub[fin_u], -lb[fin_l]]
    G = np.vstack(rows)
    h = np.concatenate(rhs)
    # equalities: eliminate through a null-space parametrisation x = xp + Z t
    if len(beq):
        keep = np.linalg.norm(Aeq, axis=1) > 0  # identically-zero rows (EqSystem) carry no information
        Ae, be = Aeq[keep], beq[keep]
        if not np.allclose(beq[~keep], 0.0, atol=1e-12):
            return None, False
        xp = np.linalg.lstsq(Ae, be, rcond=None)[0]
        Z = sla.null_space(Ae)
    else:
        xp, Z = np.zeros(n), np.eye(n)
    Qz, cz = Z.T @ Q @ Z, Z.T @ (Q @ xp + c)
    Gz, hz = G @ Z, h - G @ xp
    L = np.linalg.cholesky(Qz)
    Gy = np.linalg.solve(L, Gz.T).T  # G L^-T
    y0 = np.linalg.solve(L, cz)  # L^-1 c
    hy = hz + Gy @ y0
    # Lawson-Hanson LDP is stated for  G y >= h : negate our  Gy y <= hy
    E = np.vstack([-Gy.T, -hy[None, :]])
    fvec = np.zeros(E.shape[0])
    fvec[-1] = 1.0
    if E.shape[1]:
        u, rn = nnls(E, fvec, maxiter=50 * E.shape[1])
        r = E @ u - fvec
        if np.linalg.norm(r) < 1e-12:
            return None, False  # infeasible
        y = -r[:-1] / r[-1]
        act = np.where(u > 0)[0]  # NNLS's own active set (exact zeros elsewhere)
    else:
        y = np.zeros(Z.shape[1])
        act = np.zeros(0, dtype=int)
    t = np.linalg.solve(L.T, y - y0)
    x = xp + Z @ t
    # KKT polish on the active set
    for _ in range(5):
        Aact = np.vstack([Aeq, G[act]]) if len(act) or len(beq) else np.zeros((0, n))
        bact = np.concatenate([beq, h[act]])
        # rows may be dependent (zero rows): least squares on the KKT system
        K = np.block([[Q, Aact.T], [Aact, np.zeros((len(bact), len(bact)))]])
        sol = np.linalg.lstsq(K, np.concatenate([-c, bact]), rcond=None)[0]
        xk, lam = sol[:n], sol[n:]
        lam_in = lam[len(beq):]
        viol = G @ xk - h
        if (lam_in >= -1e-9).all() and (viol <= 1e-9).all():
            stat = np.abs(Q @ xk + c + Aact.T @ lam).max()
            if stat < 1e-7 * (1 + np.abs(c).max()):
                return xk, True
        # drop negative multipliers / add violated rows and retry
        act = np.array(sorted((set(act[lam_in >= -1e-9]) | set(np.where(viol > 1e-9)[0]))), dtype=int)
    return x, False


def make_case(name, pb):
    qp = build_qp(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"])
    x, ok = solve_qp_ldp(qp["Q"], qp["c"], qp["Aeq"], qp["beq"], qp["Aineq"], qp["bineq"], qp["lb"], qp["ub"])
    assert ok, name
    traj = qp["Phi"] @ np.asarray(pb["x0"], float) + qp["Psi"] @ x + qp["xi"]
    return dict(control=x, trajectory=traj, Q=qp["Q"], c=qp["c"], Aeq=qp["Aeq"], beq=qp["beq"], Aineq=qp["Aineq"],
                bineq=qp["bineq"], lb=qp["lb"], ub=qp["ub"])


def main():
    import fixtures as F
    from copra_amd import workloads
    cases = {}
    # BASELINE config 1: systems.h double integrator, N = 10, CPU plumbing case
    wl = workloads.double_integrator(4, N=10, seed=0)
    for b in range(4):
        pb = dict(A=wl["A"][b], B=wl["B"][b], d=wl["d"][b], x0=wl["x0"][b], N=10, costs=wl["costs"], cstrs=wl["cstrs"])
        cases["dint_%d" % b] = (pb, make_case("dint", pb))
    # BASELINE config 3 (headline shape), both constraint settings
    for tag, (vm, um) in (("com", (0.6, 3.0)), ("comtight", (0.25, 1.2))):
        wl = workloads.com_preview(6, seed=1, v_max=vm, u_max=um)
        for b in range(6):
            pb = dict(A=wl["A"][b], B=wl["B"][b], d=wl["d"][b], x0=wl["x0"][b], N=20, costs=wl["costs"],
                      cstrs=wl["cstrs"])
            cases["%s_%d" % (tag, b)] = (pb, make_case(tag, pb))
    # the reference's fixtures at a short horizon: every cost class x every constraint class
    for system in ("bounded", "ineq", "mixed", "eq"):
        for xcost in ("target", "trajectory", "mixed"):
            pb = getattr(F, system + "_system")(xcost, N=12)
            cases["%s_%s" % (system, xcost)] = (pb, make_case(system, pb))
    pb = F.initial_state_problem(False)
    cases["nine_classes"] = (pb, make_case("nine", pb))
    pb = F.com_walk_problem()
    cases["com_walk"] = (pb, make_case("walk", pb))
    out = {}
    for name, (pb, sol) in cases.items():
        for k, v in sol.items():
            out["%s/%s" % (name, k)] = v
        out["%s/x0" % name] = np.asarray(pb["x0"], float)
        out["%s/A" % name] = np.asarray(pb["A"], float)
        out["%s/B" % name] = np.asarray(pb["B"], float)
        out["%s/d" % name] = np.asarray(pb["d"], float)
    np.savez_compressed(os.path.join(HERE, "golden_lmpc.npz"), **out)
    print("wrote %d cases" % len(cases))


if __name__ == "__main__":
    main()
