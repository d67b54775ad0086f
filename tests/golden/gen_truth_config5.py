#!/usr/bin/env python3
"""Extended-precision TRUTH vectors for BASELINE config 5 (InitialStateLMPC, nx=12, nu=6, N=50, R = 1e-6 I).

Why: with the SURVEY-specified R = 1e-6 I the Hessian of InitialStateLMPC::makeQPForm
(src/InitialStateLMPC.cpp:77-122)   H = [[R + E Q^-1 E', E], [E', Q]]   has the Schur complement R = 1e-6 against entries
of 1e2 (cond ~ 1e10+), so two valid FP64 evaluation orders of the SAME formulas (the reference / oracle: LU
`.inverse()`; the device: (E J)(E J)') differ by ~1e-5 in U.  This script establishes which side is nearer the
mathematical optimum of the problem the reference DEFINES:

  * every matrix of the QP (Psi, Phi, Q, E, f, the constraint rows Y | A, z) is evaluated from the primary float64
    data (A, B, M, W, p, G, ...) in 60-digit arithmetic (mpmath), including E Q^-1 E' through a 60-digit LU solve;
  * the constraints the CPU oracle's solution holds to within 3e-6 are taken as the candidate active set (its own
    iterate leaves ACTIVE bounds slack by up to 5e-7 at this conditioning), the KKT system on it is solved in 60-digit
    arithmetic, and the result is CERTIFIED in the same arithmetic: every inactive constraint strictly satisfied,
    every multiplier of an active inequality >= 0  (the QP is strictly convex, so this is THE optimum);
  * the float64 roundings of x0*, U*, X* = Phi x0* + Psi U* + xi are written to tests/golden/config5_truth.npz
    together with the oracle's error against them.

SELF-GENERATED (the reference holds no numeric vectors and cannot be built here); independent of the oracle's and
the kernels' arithmetic.  Run:  python tests/golden/gen_truth_config5.py   (about 3 minutes)
"""
import os
import sys
import time

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

mp.mp.dps = 60
_mpf = np.frompyfunc(lambda v: mp.mpf(float(v)), 1, 1)
_flt = np.frompyfunc(float, 1, 1)


def M_(a):
    """float64 array -> object array of mpf (exact: every double is a 60-digit mpf)"""
    return _mpf(np.asarray(a, dtype=np.float64))


def F_(a):
    return np.asarray(_flt(a), dtype=np.float64)


def zeros(*shape):
    out = np.empty(shape, dtype=object)
    out[...] = mp.mpf(0)
    return out


def lu_solve(Ain, Bin):
    """Gaussian elimination with partial pivoting on object arrays; B may have several columns"""
    A = Ain.copy()
    B = Bin.copy().reshape(Bin.shape[0], -1)
    n = A.shape[0]
    for k in range(n):
        p = k + int(np.argmax([abs(v) for v in A[k:, k]]))
        if p != k:
            A[[k, p]] = A[[p, k]]
            B[[k, p]] = B[[p, k]]
        piv = A[k, k]
        fac = A[k + 1:, k] / piv
        nz = np.nonzero(fac != 0)[0]
        if nz.size:
            A[k + 1 + nz, k:] -= np.outer(fac[nz], A[k, k:])
            B[k + 1 + nz] -= np.outer(fac[nz], B[k])
    X = zeros(*B.shape)
    for k in range(n - 1, -1, -1):
        X[k] = (B[k] - (A[k, k + 1:].dot(X[k + 1:]) if k + 1 < n else 0)) / A[k, k]
    return X.reshape(Bin.shape)


def build_mp(wl, k):
    """The QP of InitialStateLMPC for instance k of the config-5 workload, in mpf.  Variables [x0; U]."""
    A, B, d = M_(wl["A"][k]), M_(wl["B"][k]), M_(wl["d"][k])
    N = wl["N"]
    nx, nu = 12, 6
    X, U = nx * (N + 1), nu * N
    # PreviewSystem::updateSystem (src/PreviewSystem.cpp:57-74), closed form Psi_ij = A^(i-1-j) B
    G = [B]
    Ph = [M_(np.eye(nx))]
    xi = [zeros(nx)]
    for i in range(1, N + 1):
        Ph.append(A.dot(Ph[-1]))
        xi.append(A.dot(xi[-1]) + d)
        if i > 1:
            G.append(A.dot(G[-1]))
    Phi = zeros(X, nx)
    Psi = zeros(X, U)
    xiv = zeros(X)
    for i in range(N + 1):
        Phi[i * nx:(i + 1) * nx] = Ph[i]
        xiv[i * nx:(i + 1) * nx] = xi[i]
        for j in range(i):
            Psi[i * nx:(i + 1) * nx, j * nu:(j + 1) * nu] = G[i - 1 - j]
    Q = zeros(U, U)
    for i in range(U):
        Q[i, i] = mp.mpf(1e-6)  # LMPC::updateSystem (src/LMPC.cpp:228-230)
    E = zeros(nx, U)
    f = zeros(U)
    for c in wl["costs"]:
        w = M_(np.broadcast_to(np.asarray(c["weights"], dtype=float), (np.atleast_1d(c["p"]).shape[0],)))
        p = M_(np.atleast_1d(c["p"]))
        if c["kind"] == "trajectory":  # costFunctions.cpp:63-82, per-step entry
            Mm = M_(c["M"])
            for i in range(N + 1):
                tmp = Mm.dot(Psi[i * nx:(i + 1) * nx])  # r x U
                wt = w[:, None] * tmp
                Q += tmp.T.dot(wt)
                E += (Mm.dot(Ph[i])).T.dot(wt)
                f += (Mm.dot(xi[i]) - p).dot(wt)
        elif c["kind"] == "control":  # costFunctions.cpp:139-158
            Nm = M_(c["N"])
            mat = Nm.T.dot(w[:, None] * Nm)
            vec = -(p * w).dot(Nm)
            for i in range(N):
                Q[i * nu:(i + 1) * nu, i * nu:(i + 1) * nu] += mat
                f[i * nu:(i + 1) * nu] += vec
        else:
            raise NotImplementedError(c["kind"])
    ist = wl["initial_state"]
    QiEt = lu_solve(Q, E.T.copy())  # Q^-1 E'  (U x nx)
    n = nx + U
    H = zeros(n, n)
    H[:nx, :nx] = M_(ist["R"]) + E.dot(QiEt)  # InitialStateLMPC.cpp:113-118
    H[:nx, nx:] = E
    H[nx:, :nx] = E.T
    H[nx:, nx:] = Q
    g = zeros(n)
    g[:nx] = M_(ist["r"])
    g[nx:] = f
    rows_eq, rhs_eq, rows_in, rhs_in = [], [], [], []
    for c in wl["cstrs"]:
        if c["kind"] == "mixed":  # constraints.cpp:197-226
            Em, Gm, fm = M_(c["E"]), M_(c["G"]), M_(np.atleast_1d(c["f"]))
            for i in range(N):
                blk = zeros(Em.shape[0], n)
                blk[:, :nx] = Em.dot(Ph[i])
                blk[:, nx:] = Em.dot(Psi[i * nx:(i + 1) * nx])
                blk[:, nx + i * nu:nx + (i + 1) * nu] += Gm
                rows_in.append(blk)
                rhs_in.append(fm - Em.dot(xi[i]))
        elif c["kind"] == "trajectory":  # full-size entry, constraints.cpp:68-73
            Em, fm = M_(c["E"]), M_(np.atleast_1d(c["f"]))
            assert Em.shape[1] == X
            blk = zeros(Em.shape[0], n)
            blk[:, :nx] = Em.dot(Phi)
            blk[:, nx:] = Em.dot(Psi)
            (rows_in if c.get("ineq", True) else rows_eq).append(blk)
            (rhs_in if c.get("ineq", True) else rhs_eq).append(fm - Em.dot(xiv))
        elif c["kind"] == "control_bound":
            lb = np.concatenate([ist["x0lb"][k], np.tile(np.asarray(c["lower"], float), N)])
            ub = np.concatenate([ist["x0ub"][k], np.tile(np.asarray(c["upper"], float), N)])
        else:
            raise NotImplementedError(c["kind"])
    Aeq, beq = np.vstack(rows_eq), np.concatenate(rhs_eq)
    Ain, bin_ = np.vstack(rows_in), np.concatenate(rhs_in)
    return dict(H=H, g=g, Aeq=Aeq, beq=beq, Ain=Ain, bin=bin_, lb=M_(lb), ub=M_(ub), Phi=Phi, Psi=Psi, xi=xiv, nx=nx)


def certify(qp, z_guess, tol_act=3e-6):
    """active set from z_guess -> KKT solve in mpf -> optimality certificate"""
    H, g = qp["H"], qp["g"]
    n = H.shape[0]
    zg = M_(z_guess)
    sl = F_(qp["bin"] - qp["Ain"].dot(zg))
    act_in = np.nonzero(sl < tol_act)[0]
    act_lb = np.nonzero(z_guess - F_(qp["lb"]) < tol_act)[0]
    act_ub = np.nonzero(F_(qp["ub"]) - z_guess < tol_act)[0]
    for _attempt in range(6):
        rows = [qp["Aeq"], qp["Ain"][act_in]]
        rhs = [qp["beq"], qp["bin"][act_in]]
        for j in act_ub:
            e = zeros(1, n)
            e[0, j] = mp.mpf(1)
            rows.append(e)
            rhs.append(qp["ub"][[j]])
        for j in act_lb:
            e = zeros(1, n)
            e[0, j] = mp.mpf(-1)
            rows.append(e)
            rhs.append(-qp["lb"][[j]])
        Cm, bv = np.vstack(rows), np.concatenate(rhs)
        m = Cm.shape[0]
        K = zeros(n + m, n + m)
        K[:n, :n] = H
        K[:n, n:] = Cm.T
        K[n:, :n] = Cm
        sol = lu_solve(K, np.concatenate([-g, bv]))
        z, lam = sol[:n], sol[n:]
        neq = qp["Aeq"].shape[0]
        lam_in = lam[neq:]
        # certificate
        stat = max(abs(v) for v in (H.dot(z) + g + Cm.T.dot(lam)))
        sl_all = qp["bin"] - qp["Ain"].dot(z)
        viol = [i for i in range(len(sl_all)) if sl_all[i] < -mp.mpf(10) ** -40 and i not in set(act_in)]
        vlb = [j for j in range(n) if z[j] - qp["lb"][j] < -mp.mpf(10) ** -40 and j not in set(act_lb)]
        vub = [j for j in range(n) if qp["ub"][j] - z[j] < -mp.mpf(10) ** -40 and j not in set(act_ub)]
        neg = [i for i in range(len(lam_in)) if lam_in[i] < 0]
        if not viol and not vlb and not vub and not neg:
            return dict(z=z, lam=lam, act_in=act_in, act_lb=act_lb, act_ub=act_ub, stationarity=float(stat),
                        min_mult=float(min(lam_in)) if len(lam_in) else 0.0,
                        min_inactive_slack=float(min([sl_all[i] for i in range(len(sl_all)) if i not in set(act_in)])))
        # repair the guess (weakly active constraints): drop negative multipliers, add violated rows
        print("   repairing active set: %d violated rows, %d/%d violated bounds, %d negative multipliers"
              % (len(viol), len(vlb), len(vub), len(neg)), flush=True)
        nin, nub = len(act_in), len(act_ub)
        drop_in = [i for i in neg if i < nin]
        drop_ub = [i - nin for i in neg if nin <= i < nin + nub]
        drop_lb = [i - nin - nub for i in neg if i >= nin + nub]
        act_in = np.array(sorted((set(act_in) - {act_in[i] for i in drop_in}) | set(viol)), dtype=int)
        act_ub = np.array(sorted((set(act_ub) - {act_ub[i] for i in drop_ub}) | set(vub)), dtype=int)
        act_lb = np.array(sorted((set(act_lb) - {act_lb[i] for i in drop_lb}) | set(vlb)), dtype=int)
    raise RuntimeError("no certified active set")


def main():
    import pyoracle
    from copra_amd import workloads
    b = 6
    wl = workloads.long_horizon_initial_state(b, R_diag=1e-6)
    ist = wl["initial_state"]
    pick = [0, 2, 5]
    out = dict(instances=np.array(pick), batch=b, r_diag=1e-6)
    for k in pick:
        t0 = time.time()
        io = dict(R=ist["R"], r=ist["r"], x0lb=ist["x0lb"][k], x0ub=ist["x0ub"][k])
        ro = pyoracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], wl["costs"], wl["cstrs"],
                                 initial_state=io)
        assert ro["status"] == 0
        zo = np.concatenate([ro["x0_opt"], ro["control"]])
        qp = build_mp(wl, k)
        print("instance %d: mp build %.0f s" % (k, time.time() - t0), flush=True)
        qo = pyoracle.lmpc_build(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], wl["costs"], wl["cstrs"],
                                 initial_state=io)
        Hf = F_(qp["H"])
        print("   oracle-vs-mp  H: max abs diff %.3e (top-left block %.3e, |H| max %.3e)   c: %.3e   Aineq: %.3e"
              % (np.abs(Hf - qo["Q"]).max(), np.abs(Hf[:12, :12] - qo["Q"][:12, :12]).max(), np.abs(Hf).max(),
                 np.abs(F_(qp["g"]) - qo["c"]).max(), np.abs(F_(qp["Ain"]) - qo["Aineq"]).max()), flush=True)
        cert = certify(qp, zo)
        z = cert["z"]
        Xt = qp["Phi"].dot(z[:12]) + qp["Psi"].dot(z[12:]) + qp["xi"]
        zt, Xf = F_(z), F_(Xt)
        eu = np.abs(zo[12:] - zt[12:])
        print("   certified: stationarity %.1e, min multiplier %.3e, min inactive slack %.3e, |active| = %d+%d+%d"
              % (cert["stationarity"], cert["min_mult"], cert["min_inactive_slack"], len(cert["act_in"]),
                 len(cert["act_lb"]), len(cert["act_ub"])), flush=True)
        print("   oracle vs truth: max|dU| %.3e  max|dU|/(1+|U|) %.3e  max|dx0| %.3e  max|dX| %.3e   (%.0f s)"
              % (eu.max(), (eu / (1 + np.abs(zt[12:]))).max(), np.abs(zo[:12] - zt[:12]).max(),
                 np.abs(ro["trajectory"] - Xf).max(), time.time() - t0), flush=True)
        out["x0_opt_%d" % k] = zt[:12]
        out["control_%d" % k] = zt[12:]
        out["trajectory_%d" % k] = Xf
        out["active_ineq_%d" % k] = cert["act_in"]
        out["active_lb_%d" % k] = cert["act_lb"]
        out["active_ub_%d" % k] = cert["act_ub"]
        out["min_mult_%d" % k] = cert["min_mult"]
        out["oracle_err_control_%d" % k] = eu.max()
        out["H_topleft_%d" % k] = Hf[:12, :12]
    np.savez_compressed(os.path.join(HERE, "config5_truth.npz"), **out)
    print("wrote config5_truth.npz")


if __name__ == "__main__":
    main()
