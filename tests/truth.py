"""Extended-precision TRUTH for a parity comparison (test infrastructure, independent of the oracle's and the kernels' arithmetic).

Why: the north-star bar is "within 1e-6 relative of copra's CPU QuadProgDense path".  Where the condensed Hessian is ill conditioned
(cond 1e6 ... 1e12) two valid FP64 evaluations of the SAME formulas differ by more than that on entries that nearly vanish at the
optimum, and a comparison HIP-vs-oracle cannot say which side is off.  This module evaluates the QP the reference DEFINES

    LMPC::updateSystem / makeQPForm            src/LMPC.cpp:225-280
    PreviewSystem::updateSystem                src/PreviewSystem.cpp:57-74
    {Trajectory,Target,Control,Mixed}Cost      src/costFunctions.cpp:44-215
    the five constraint classes                src/constraints.cpp:45-367   (quirk Q1 of TrajectoryBoundConstraint kept)
    InitialStateLMPC::makeQPForm               src/InitialStateLMPC.cpp:77-122
    bounds as rows [I; -I]                     src/QuadProgSolver.cpp:59-69

from the primary float64 data in EXTENDED precision -- x87 80-bit `numpy.longdouble` (64-bit mantissa, eps 1.1e-19) by default,
`mpmath` at 50 digits with arith="mp" --, takes the constraints a float64 solution holds as the candidate active set, solves the
KKT system on it in the same arithmetic and CERTIFIES the result there: stationarity, every inactive constraint satisfied, every
multiplier of an active inequality non-negative.  The QP is strictly convex, so a certified point is THE optimum.
tests/test_oracle.py checks the longdouble evaluation against the 50-digit one (<= 1e-15) and the config-5 fixtures of
tests/golden/gen_truth_config5.py, so that the GPU tests can hold the device to 1e-6 of the truth, entry by entry."""
import numpy as np

LD = np.longdouble


class Arith:
    """the two arithmetics behind one interface: conversion of float64 data, zeros, and back to float64"""

    def __init__(self, kind="longdouble", dps=50):
        self.kind = kind
        if kind == "mp":
            import mpmath
            self.mp = mpmath
            mpmath.mp.dps = dps
            self._to = np.frompyfunc(lambda v: mpmath.mpf(float(v)), 1, 1)
            self._flt = np.frompyfunc(float, 1, 1)
        else:
            assert np.finfo(LD).nmant >= 63, "numpy.longdouble is not an extended type on this host"

    def cv(self, a):
        a = np.asarray(a, dtype=np.float64)
        return self._to(a) if self.kind == "mp" else a.astype(LD)

    def zeros(self, *shape):
        if self.kind == "mp":
            out = np.empty(shape, dtype=object)
            out[...] = self.mp.mpf(0)
            return out
        return np.zeros(shape, dtype=LD)

    def f64(self, a):
        return np.asarray(self._flt(a), dtype=np.float64) if self.kind == "mp" else np.asarray(a, dtype=np.float64)

    def eye(self, n):
        return self.cv(np.eye(n))


def lu_solve(Ain, Bin):
    """Gaussian elimination with partial pivoting, vectorised over rows; works on longdouble and on object (mpf) arrays"""
    A = Ain.copy()
    B = Bin.copy().reshape(Bin.shape[0], -1)
    n = A.shape[0]
    for k in range(n):
        p = k + int(np.argmax(np.abs(A[k:, k])))
        if p != k:
            A[[k, p]] = A[[p, k]]
            B[[k, p]] = B[[p, k]]
        fac = A[k + 1:, k] / A[k, k]
        nz = np.nonzero(fac != 0)[0]
        if nz.size:
            A[k + 1 + nz, k + 1:] -= np.outer(fac[nz], A[k, k + 1:])
            B[k + 1 + nz] -= np.outer(fac[nz], B[k])
    X = B.copy()
    for k in range(n - 1, -1, -1):
        if k + 1 < n:
            X[k] = X[k] - A[k, k + 1:].dot(X[k + 1:])
        X[k] = X[k] / A[k, k]
    return X.reshape(Bin.shape)


def _weights(c, rows):
    w = c.get("weights")
    w = np.ones(rows) if w is None else np.atleast_1d(np.asarray(w, dtype=np.float64))
    if w.shape[0] != rows:  # CostFunction::weights tiling (costFunctions.h:54-67)
        assert rows % w.shape[0] == 0
        w = np.tile(w, rows // w.shape[0])
    return w


def build_qp(A, B, d, x0, N, costs, cstrs, initial_state=None, ar=None):
    """the dense QP of one instance; variables U (LMPC) or [x0; U] (InitialStateLMPC) -- min 1/2 z'Hz + g'z,
    Aeq z = beq, Ain z <= bin, lb <= z <= ub"""
    ar = ar or Arith()
    A, B, d, x0 = ar.cv(A), ar.cv(B), ar.cv(d), ar.cv(x0)
    nx, nu = B.shape
    X, U = nx * (N + 1), nu * N
    Ph, xi, G = [ar.eye(nx)], [ar.zeros(nx)], [B]
    for i in range(1, N + 1):
        Ph.append(A.dot(Ph[-1]))
        xi.append(A.dot(xi[-1]) + d)
        if i > 1:
            G.append(A.dot(G[-1]))
    Phi, Psi, xiv = ar.zeros(X, nx), ar.zeros(X, U), ar.zeros(X)
    for i in range(N + 1):
        Phi[i * nx:(i + 1) * nx] = Ph[i]
        xiv[i * nx:(i + 1) * nx] = xi[i]
        for j in range(i):
            Psi[i * nx:(i + 1) * nx, j * nu:(j + 1) * nu] = G[i - 1 - j]
    Q = ar.zeros(U, U)
    Q[np.arange(U), np.arange(U)] = ar.cv(np.full(U, 1e-6))  # LMPC::updateSystem (src/LMPC.cpp:228-230)
    E, f = ar.zeros(nx, U), ar.zeros(U)

    def acc(tmp, mphi, mxi_minus_p, w):
        nonlocal Q, E, f
        wt = w[:, None] * tmp
        Q += tmp.T.dot(wt)
        E += mphi.T.dot(wt)
        f += mxi_minus_p.dot(wt)

    for c in costs:
        p = np.atleast_1d(np.asarray(c["p"], dtype=np.float64))
        rows = p.shape[0]
        w, p = ar.cv(_weights(c, rows)), ar.cv(p)
        M = ar.cv(np.atleast_2d(c["M"])) if c.get("M") is not None else None
        Nm = ar.cv(np.atleast_2d(c["N"])) if c.get("N") is not None else None
        kind = c["kind"]
        if kind == "trajectory":
            if M.shape[1] == X:  # full-size entry: one product (costFunctions.cpp:65-71)
                acc(M.dot(Psi), M.dot(Phi), M.dot(xiv) - p, w)
            else:
                for i in range(N + 1):
                    acc(M.dot(Psi[i * nx:(i + 1) * nx]), M.dot(Ph[i]), M.dot(xi[i]) - p, w)
        elif kind == "target":
            acc(M.dot(Psi[N * nx:]), M.dot(Ph[N]), M.dot(xi[N]) - p, w)
        elif kind == "control":
            if Nm.shape[1] == U:
                Q += Nm.T.dot(w[:, None] * Nm)
                f += -(p * w).dot(Nm)
            else:
                mat, vec = Nm.T.dot(w[:, None] * Nm), -(p * w).dot(Nm)
                for i in range(N):
                    Q[i * nu:(i + 1) * nu, i * nu:(i + 1) * nu] += mat
                    f[i * nu:(i + 1) * nu] += vec
        elif kind == "mixed":
            if M.shape[1] == X:  # full-size entry (costFunctions.cpp:197-203): M over the whole trajectory, N over all controls
                acc(M.dot(Psi) + Nm, M.dot(Phi), M.dot(xiv) - p, w)
            else:
                for i in range(N):
                    tmp = M.dot(Psi[i * nx:(i + 1) * nx])
                    tmp[:, i * nu:(i + 1) * nu] += Nm
                    acc(tmp, M.dot(Ph[i]), M.dot(xi[i]) - p, w)
        else:
            raise NotImplementedError(kind)

    rows_eq, rhs_eq, rows_in, rhs_in = [], [], [], []  # rows as (Y | A), right-hand side z  (b = z - Y x0)
    lbU, ubU = np.full(U, -np.finfo(np.float64).max), np.full(U, np.finfo(np.float64).max)  # LMPC.cpp:207-208

    def put(c, Y, Am, z):
        ineq = c.get("ineq", True)
        (rows_in if ineq else rows_eq).append(np.hstack([Y, Am]))
        (rhs_in if ineq else rhs_eq).append(z)

    for c in cstrs:
        kind = c["kind"]
        if kind == "trajectory":
            Em, fm = ar.cv(np.atleast_2d(c["E"])), ar.cv(np.atleast_1d(c["f"]))
            if Em.shape[1] == X:
                put(c, Em.dot(Phi), Em.dot(Psi), fm - Em.dot(xiv))
            else:
                for i in range(N + 1):
                    put(c, Em.dot(Ph[i]), Em.dot(Psi[i * nx:(i + 1) * nx]), fm - Em.dot(xi[i]))
        elif kind == "control":
            Gm, fm = ar.cv(np.atleast_2d(c["G"])), ar.cv(np.atleast_1d(c["f"]))
            if Gm.shape[1] == U:
                put(c, ar.zeros(Gm.shape[0], nx), Gm, fm)
            else:
                for i in range(N):
                    blk = ar.zeros(Gm.shape[0], U)
                    blk[:, i * nu:(i + 1) * nu] = Gm
                    put(c, ar.zeros(Gm.shape[0], nx), blk, fm)
        elif kind == "mixed":
            Em, Gm, fm = ar.cv(np.atleast_2d(c["E"])), ar.cv(np.atleast_2d(c["G"])), ar.cv(np.atleast_1d(c["f"]))
            if Em.shape[1] == X:  # full-size entry (constraints.cpp:199-204): E over the whole trajectory, G over all controls
                put(c, Em.dot(Phi), Em.dot(Psi) + Gm, fm - Em.dot(xiv))
                continue
            for i in range(N):
                blk = Em.dot(Psi[i * nx:(i + 1) * nx])
                blk[:, i * nu:(i + 1) * nu] += Gm
                put(c, Em.dot(Ph[i]), blk, fm - Em.dot(xi[i]))
        elif kind == "trajectory_bound":
            # quirk Q1 (constraints.cpp:284-315): first all finite LOWER components, step-major, written as Psi_row U <= lower - ...
            # (not negated), then all finite upper components
            lo, up = np.atleast_1d(np.asarray(c["lower"], float)), np.atleast_1d(np.asarray(c["upper"], float))
            for bound in (lo, up):
                idx = [j for j in range(nx) if np.isfinite(bound[j])]
                for i in range(N + 1):
                    for j in idx:
                        r = i * nx + j
                        put(dict(ineq=True), Phi[r:r + 1], Psi[r:r + 1], ar.cv([bound[j]]) - xiv[r:r + 1])
        elif kind == "control_bound":
            lo, hi = np.asarray(c["lower"], float).reshape(-1), np.asarray(c["upper"], float).reshape(-1)
            # (per-step entry tiled over the steps, or a full-size entry taken as it is: constraints.cpp:333-357)
            lbU, ubU = (lo, hi) if lo.size == U else (np.tile(lo, N), np.tile(hi, N))
        else:
            raise NotImplementedError(kind)

    def stack(rows, rhs):
        if not rows:
            return ar.zeros(0, nx + U), ar.zeros(0)
        return np.vstack(rows), np.concatenate(rhs)

    YAe, ze = stack(rows_eq, rhs_eq)
    YAi, zi = stack(rows_in, rhs_in)
    if initial_state is None:
        g = E.T.dot(x0) + f  # costFunctions.cpp:80
        return dict(H=Q, g=g, Aeq=YAe[:, nx:], beq=ze - YAe[:, :nx].dot(x0), Ain=YAi[:, nx:], bin=zi - YAi[:, :nx].dot(x0),
                    lb=lbU, ub=ubU, Phi=Phi, Psi=Psi, xi=xiv, x0=x0, nx0=0, ar=ar)
    ist = initial_state
    n = nx + U
    H = ar.zeros(n, n)
    H[:nx, :nx] = ar.cv(ist["R"]) + E.dot(lu_solve(Q, E.T.copy()))  # InitialStateLMPC.cpp:113-118
    H[:nx, nx:] = E
    H[nx:, :nx] = E.T
    H[nx:, nx:] = Q
    g = np.concatenate([ar.cv(ist["r"]), f])
    lb = np.concatenate([np.asarray(ist["x0lb"], float), lbU])
    ub = np.concatenate([np.asarray(ist["x0ub"], float), ubU])
    return dict(H=H, g=g, Aeq=YAe, beq=ze, Ain=YAi, bin=zi, lb=lb, ub=ub, Phi=Phi, Psi=Psi, xi=xiv, x0=None, nx0=nx, ar=ar)


def certify(qp, z_guess, tol_act=3e-6, tiny=1e-17):
    """active set from the float64 solution z_guess -> KKT solve in the QP's arithmetic -> optimality certificate.
    Returns dict(z (float64), control, trajectory, x0_opt, stationarity, min_mult, min_inactive_slack, n_active)."""
    ar = qp["ar"]
    H, g = qp["H"], qp["g"]
    n = H.shape[0]
    zg = ar.cv(z_guess)
    lb, ub = qp["lb"], qp["ub"]
    fin_lb, fin_ub = np.abs(lb) < 1e300, np.abs(ub) < 1e300
    lbx, ubx = ar.cv(np.where(fin_lb, lb, 0.0)), ar.cv(np.where(fin_ub, ub, 0.0))
    sl = ar.f64(qp["bin"] - qp["Ain"].dot(zg)) if qp["Ain"].shape[0] else np.zeros(0)
    act_in = set(np.nonzero(sl < tol_act)[0])
    act_lb = set(np.nonzero(fin_lb & (z_guess - np.where(fin_lb, lb, 0.0) < tol_act))[0])
    act_ub = set(np.nonzero(fin_ub & (np.where(fin_ub, ub, 0.0) - z_guess < tol_act))[0])
    pinned = act_lb & act_ub  # lb == ub (InitialStateLMPC's default x0 bounds): one equality row
    for _attempt in range(8):
        a_in, a_lb, a_ub = sorted(act_in), sorted(act_lb - pinned), sorted(act_ub - pinned)
        a_pin = sorted(pinned)
        rows, rhs = [qp["Aeq"]], [qp["beq"]]
        for j in a_pin:
            e = ar.zeros(1, n)
            e[0, j] = ar.cv(1.0)
            rows.append(e)
            rhs.append(ubx[[j]])
        neq = qp["Aeq"].shape[0] + len(a_pin)
        rows.append(qp["Ain"][a_in])
        rhs.append(qp["bin"][a_in])
        for j in a_ub:
            e = ar.zeros(1, n)
            e[0, j] = ar.cv(1.0)
            rows.append(e)
            rhs.append(ubx[[j]])
        for j in a_lb:
            e = ar.zeros(1, n)
            e[0, j] = ar.cv(-1.0)
            rows.append(e)
            rhs.append(-lbx[[j]])
        Cm, bv = np.vstack(rows), np.concatenate(rhs)
        m = Cm.shape[0]
        K = ar.zeros(n + m, n + m)
        K[:n, :n] = H
        K[:n, n:] = Cm.T
        K[n:, :n] = Cm
        sol = lu_solve(K, np.concatenate([-g, bv]))
        z, lam = sol[:n], sol[n:]
        lam_in = ar.f64(lam[neq:])
        stat = float(np.max(np.abs(ar.f64(H.dot(z) + g + Cm.T.dot(lam)))))
        sl_all = ar.f64(qp["bin"] - qp["Ain"].dot(z)) if qp["Ain"].shape[0] else np.zeros(0)
        zf = ar.f64(z)
        viol = [i for i in range(len(sl_all)) if sl_all[i] < -tiny and i not in act_in]
        vlb = [j for j in range(n) if fin_lb[j] and ar.f64(z[j] - lbx[j]) < -tiny and j not in act_lb]
        vub = [j for j in range(n) if fin_ub[j] and ar.f64(ubx[j] - z[j]) < -tiny and j not in act_ub]
        neg = [i for i in range(len(lam_in)) if lam_in[i] < 0]
        if not viol and not vlb and not vub and not neg:
            nx0 = qp["nx0"]
            if qp.get("Phi") is not None:
                x0 = z[:nx0] if nx0 else qp["x0"]
                Xt = ar.f64(qp["Phi"].dot(x0) + qp["Psi"].dot(z[nx0:]) + qp["xi"])
            else:  # (a plain dense QP: no preview system behind it)
                Xt = None
            inact = [sl_all[i] for i in range(len(sl_all)) if i not in act_in]
            return dict(z=zf, control=zf[nx0:], x0_opt=zf[:nx0], trajectory=Xt, stationarity=stat,
                        min_mult=float(lam_in.min()) if len(lam_in) else 0.0,
                        min_inactive_slack=float(min(inact)) if inact else np.inf,
                        n_active=(len(a_in), len(a_lb), len(a_ub), len(a_pin)), attempts=_attempt + 1)
        # repair a weakly active guess: drop negative multipliers, add violated rows
        nin, nub = len(a_in), len(a_ub)
        for i in neg:
            if i < nin:
                act_in.discard(a_in[i])
            elif i < nin + nub:
                act_ub.discard(a_ub[i - nin])
            else:
                act_lb.discard(a_lb[i - nin - nub])
        act_in |= set(viol)
        act_lb |= set(vlb)
        act_ub |= set(vub)
    raise RuntimeError("no certified active set")


def solve(A, B, d, x0, N, costs, cstrs, z_guess, initial_state=None, arith="longdouble"):
    """certified optimum of one instance; z_guess: a float64 solution (U, or [x0; U] for InitialStateLMPC) that identifies the
    active set -- the oracle's, or the device's"""
    qp = build_qp(A, B, d, x0, N, costs, cstrs, initial_state=initial_state, ar=Arith(arith))
    return certify(qp, np.asarray(z_guess, dtype=np.float64))


def solve_dense_qp(Q, c, Aeq, beq, Aineq, bineq, XL, XU, z_guess, arith="longdouble"):
    """certified optimum of a dense QP in SolverInterface form (include/SolverInterface.h:54-80) from a float64 point that identifies
    the active set; raises RuntimeError when no active set near the guess can be certified"""
    ar = Arith(arith)
    n = len(c)
    qp = dict(H=ar.cv(Q), g=ar.cv(c), Aeq=ar.cv(np.reshape(Aeq, (-1, n))), beq=ar.cv(beq), Ain=ar.cv(np.reshape(Aineq, (-1, n))),
              bin=ar.cv(bineq), lb=np.asarray(XL, float), ub=np.asarray(XU, float), Phi=None, nx0=0, ar=ar)
    return certify(qp, np.asarray(z_guess, dtype=np.float64))


def rel(a, b, floor=1e-3):
    """the parity measure of tests/test_gpu_parity.py: max_i |a_i - b_i| / max(|b_i|, floor)"""
    return float(np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), floor)))
