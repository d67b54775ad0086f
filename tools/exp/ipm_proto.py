"""Prototype: dense Mehrotra predictor-corrector IPM on the condensed QP of config 5 vs the oracle's GI solution."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import pyoracle
from copra_amd import workloads
import scipy.linalg as sla


def ipm(H, g, Aeq, beq, C, b, iters=60, tol=1e-13, verbose=False):
    n = H.shape[0]
    m = C.shape[0]
    me = Aeq.shape[0]
    z = np.zeros(n)
    # start: unconstrained minimiser subject to equalities
    K = np.block([[H, Aeq.T], [Aeq, np.zeros((me, me))]])
    sol = np.linalg.solve(K, np.concatenate([-g, beq]))
    z = sol[:n]; nu = sol[n:]
    s = b - C @ z
    viol = max(0.0, -s.min())
    s = s + viol + 1.0
    lam = np.ones(m)
    # Mehrotra-ish init
    for it in range(iters):
        rd = H @ z + g + Aeq.T @ nu + C.T @ lam
        rp = C @ z + s - b
        re = Aeq @ z - beq
        mu = s @ lam / m
        if verbose:
            print(it, "mu %.2e rd %.2e rp %.2e re %.2e" % (mu, np.abs(rd).max(), np.abs(rp).max(), np.abs(re).max() if me else 0))
        if mu < tol and np.abs(rd).max() < 1e-9 and np.abs(rp).max() < 1e-11:
            break
        D = lam / s
        Hh = H + C.T @ (D[:, None] * C)
        cf = sla.cho_factor(Hh)
        # Schur on equalities
        Y = sla.cho_solve(cf, Aeq.T)
        S = Aeq @ Y

        def solve(rc):
            # rc = complementarity residual target: s*dlam + lam*ds = -rc
            # ds = -rp - C dz ; dlam = (-rc - lam*ds)/s
            rhs = -rd - C.T @ ((-rc + lam * rp) / s)
            t = sla.cho_solve(cf, rhs)
            if me:
                dnu = np.linalg.solve(S, Aeq @ t + re)
                dz = t - Y @ dnu
            else:
                dnu = np.zeros(0); dz = t
            ds = -rp - C @ dz
            dl = (-rc - lam * ds) / s
            return dz, dnu, ds, dl
        dz, dnu, ds, dl = solve(s * lam)
        def amax(v, dv):
            neg = dv < 0
            return min(1.0, (-v[neg] / dv[neg]).min()) if neg.any() else 1.0
        ap = amax(s, ds); ad = amax(lam, dl)
        mua = (s + ap * ds) @ (lam + ad * dl) / m
        sig = (mua / mu) ** 3
        dz, dnu, ds, dl = solve(s * lam + ds * dl - sig * mu)
        tau = max(0.995, 1 - mu) if mu < 1 else 0.995
        tau = 1 - min(0.005, mu**0.5 * 0.005) if False else 0.995 if mu > 1e-8 else 0.99999
        ap = min(1.0, tau * amax(s, ds) / 1.0) if amax(s, ds) < 1 else 1.0
        ad = min(1.0, tau * amax(lam, dl)) if amax(lam, dl) < 1 else 1.0
        z = z + ap * dz; s = s + ap * ds
        nu = nu + ad * dnu; lam = lam + ad * dl
    return z, lam, s, it


b = 6
for rdiag in (1e-6, 1e-2):
    wl = workloads.long_horizon_initial_state(b, R_diag=rdiag)
    ist = wl["initial_state"]
    for k in range(3):
        io = dict(R=ist["R"], r=ist["r"], x0lb=ist["x0lb"][k], x0ub=ist["x0ub"][k])
        args = (wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], wl["costs"], wl["cstrs"])
        ro = pyoracle.lmpc_solve(*args, initial_state=io)
        qp = pyoracle.lmpc_build(*args, initial_state=io)
        n = qp["nvar"]
        I = np.eye(n)
        fl = np.isfinite(qp["lb"]) & (qp["lb"] > -1e300)
        fu = np.isfinite(qp["ub"]) & (qp["ub"] < 1e300)
        C = np.vstack([qp["Aineq"], I[fu], -I[fl]])
        bb = np.concatenate([qp["bineq"], qp["ub"][fu], -qp["lb"][fl]])
        t = time.time()
        z, lam, s, it = ipm(qp["Q"], qp["c"], qp["Aeq"], qp["beq"], C, bb, verbose=(k == 0))
        zo = np.concatenate([ro["x0_opt"], ro["control"]])
        err = np.abs(z - zo)
        print("R=%g inst %d: ipm iters %d, oracle iters %s, max|dz| %.3e rel %.3e  x0 err %.2e   nact %d  cond(H) %.2e"
              % (rdiag, k, it, ro["iter"], err.max(), (err / (1 + np.abs(zo))).max(), err[:12].max(), (lam > s).sum(),
                 np.linalg.cond(qp["Q"])))
