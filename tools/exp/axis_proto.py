"""Prototype (numpy, scalar loops) of the per-(instance, axis) active-set solver of lmpc_axis.hpp.

On decoupled axes the condensed QP of LMPC::solve (LMPC.cpp:79-101) separates into nu independent QPs over one chain each
(nxa = nx / nu states, one control).  qpgen2's run on the whole problem is an interleaving of the axes' own runs: the
counters add up.  Each axis is solved by the Goldfarb-Idnani iteration in RANGE-SPACE (Schur) form on the Riccati factor:
    y = Q^-1 n+      backward + forward closed-loop recursion (ric_factor.hpp)
    g = N' y,  S = N' Q^-1 N  (kept explicitly, q x q),  r = S^-1 g,  z = Q^-1 (n+ - N r)
    t1 = min u_i / r_i (r_i > 0),  t2 = -s / (n+' Q^-1 n+ - g' r)
Same pick rule, same step lengths, same iterates (up to rounding) as qpgen2's J / R form -- checked here against the oracle:
status, both iteration counters, U.

    python tools/exp/axis_proto.py [batch] [v_max] [u_max]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402
from copra_amd import workloads  # noqa: E402

VSMALL = None


def vsmall():
    # qpgen2: smallest v with 1 + 0.1 v > 1 and 1 + 0.2 v > 1
    v = 1e-60
    while True:
        v = v + v
        t = 1.0 + 0.1 * v
        t2 = 1.0 + 0.2 * v
        if t > 1.0 and t2 > 1.0:
            return v


class Axis:
    def __init__(self, A, B, d, x0, H, h, HN, hN, N, rows, ub, lb, qmax=64):
        self.A, self.B, self.d, self.x0 = A, B, d, x0
        self.N = N
        self.nx = A.shape[0]
        self.rows = rows  # list of (k, e (nx), g, f, idx)
        self.ub, self.lb = ub, lb
        self.qmax = qmax
        nx = self.nx
        # sweep
        P = HN.copy()
        p = hN.copy()
        self.K = np.zeros((N, nx))
        self.kv = np.zeros(N)
        self.minv = np.zeros(N)
        AB = np.hstack([A, B.reshape(nx, 1)])
        self.bad = False
        for k in range(N - 1, -1, -1):
            M = H + AB.T @ P @ AB
            m = h + AB.T @ (P @ d + p)
            muu = M[nx, nx]
            if not muu > 0:
                self.bad = True
            mi = 1.0 / muu
            self.minv[k] = mi
            self.K[k] = -mi * M[nx, :nx]
            self.kv[k] = -mi * m[nx]
            P = M[:nx, :nx] + np.outer(M[:nx, nx], self.K[k])
            p = m[:nx] + M[:nx, nx] * self.kv[k]
        # roll-out
        self.U = np.zeros(N)
        x = x0.copy()
        for k in range(N):
            self.U[k] = self.K[k] @ x + self.kv[k]
            x = A @ x + B * self.U[k] + d

    def states(self):
        X = np.zeros((self.N + 1, self.nx))
        x = self.x0.copy()
        X[0] = x
        for k in range(self.N):
            x = self.A @ x + self.B * self.U[k] + self.d
            X[k + 1] = x
        return X

    # constraint table: index order of qpgen2 within this axis: rows (by idx), then ub_k, then lb_k
    def normal(self, c):
        """-> (nu [N], nxinj [N+1][nx]) of n+ (qpgen2 orientation a'x >= b)"""
        nu = np.zeros(self.N)
        ninj = np.zeros((self.N + 1, self.nx))
        kind, k = c[0], c[1]
        if kind == "ub":
            nu[k] = -1.0
        elif kind == "lb":
            nu[k] = 1.0
        else:
            _, k, e, g, f, idx = c
            ninj[k] = -e
            if k < self.N:
                nu[k] = -g
        return nu, ninj

    def qinv(self, nu, ninj):
        """y = Q^-1 n, nQn, and the closed-loop states xi of y"""
        N, nx = self.N, self.nx
        mu = np.zeros(nx)
        t = np.zeros(N)
        nqn = 0.0
        for k in range(N - 1, -1, -1):
            mu = mu + ninj[k + 1]
            s = nu[k] + self.B @ mu
            t[k] = s * self.minv[k]
            nqn += s * t[k]
            Acl = self.A + np.outer(self.B, self.K[k])
            mu = Acl.T @ mu + self.K[k] * nu[k]
        # (an injection at step 0 multiplies x0, which is fixed: no component)
        y = np.zeros(N)
        xi = np.zeros((N + 1, nx))
        for k in range(N):
            y[k] = self.K[k] @ xi[k] + t[k]
            xi[k + 1] = self.A @ xi[k] + self.B * y[k]
        return y, nqn, xi

    def dot(self, c, y, xi):
        """n_c+' y"""
        kind, k = c[0], c[1]
        if kind == "ub":
            return -y[k]
        if kind == "lb":
            return y[k]
        _, k, e, g, f, idx = c
        v = -(e @ xi[k])
        if k < self.N:
            v -= g * y[k]
        return v

    def slack(self, c, X):
        kind, k = c[0], c[1]
        if kind == "ub":
            return self.ub[k] - self.U[k]
        if kind == "lb":
            return self.U[k] - self.lb[k]
        _, k, e, g, f, idx = c
        v = f - e @ X[k]
        if k < self.N:
            v -= g * self.U[k]
        return v

    def norm2(self, c):
        kind, k = c[0], c[1]
        if kind in ("ub", "lb"):
            return 1.0
        nu, ninj = self.normal(c)
        # |Psi_k' e|^2 by the open-loop adjoint
        lam = np.zeros(self.nx)
        tot = 0.0
        for j in range(self.N - 1, -1, -1):
            lam = lam + ninj[j + 1]
            comp = nu[j] + self.B @ lam
            tot += comp * comp
            lam = self.A.T @ lam
        return tot

    def solve(self, max_iter=200):
        vs = VSMALL
        cons = [("row",) + tuple(r) for r in sorted(self.rows, key=lambda r: r[4])]
        cons += [("ub", k) for k in range(self.N)] + [("lb", k) for k in range(self.N)]
        norms = [np.sqrt(self.norm2(c)) for c in cons]
        act = []  # indices into cons
        lam = []
        S = np.zeros((0, 0))
        it_main = it_drop = 0
        if self.bad:
            return 2, 0, 0
        while True:
            if it_main >= max_iter:
                return 3, it_main, it_drop
            it_main += 1
            X = self.states()
            best, bi, bs = 0.0, -1, 0.0
            for i, c in enumerate(cons):
                s = self.slack(c, X)
                if abs(s) < vs:
                    s = 0.0
                if i in act:
                    s = 0.0
                # the twin of an active bound on a pinned variable: never violated (gi_core.hpp)
                if c[0] in ("ub", "lb"):
                    k = c[1]
                    pinned = (self.ub[k] - self.lb[k]) <= 1e-12 * max(1.0, abs(self.ub[k]))
                    twin = ("lb", k) if c[0] == "ub" else ("ub", k)
                    if pinned and any(cons[a][0] == twin[0] and cons[a][1] == k for a in act):
                        s = 0.0
                with np.errstate(invalid="ignore", divide="ignore"):
                    ratio = s / norms[i]
                if ratio < best:
                    best, bi, bs = ratio, i, s
            if bi < 0:
                return 0, it_main, it_drop
            p = bi
            sp = bs
            self._lam_p = 0.0
            while True:
                nu, ninj = self.normal(cons[p])
                y, nqn, xi = self.qinv(nu, ninj)
                q = len(act)
                g = np.array([self.dot(cons[a], y, xi) for a in act])
                if q:
                    r = np.linalg.solve(S, g)
                else:
                    r = np.zeros(0)
                # z = Q^-1 (n+ - N r)
                nu2, ninj2 = nu.copy(), ninj.copy()
                for a, ra in zip(act, r):
                    na, nia = self.normal(cons[a])
                    nu2 -= ra * na
                    ninj2 -= ra * nia
                z, _, xiz = self.qinv(nu2, ninj2)
                zn = nqn - (g @ r if q else 0.0)
                zz = z @ z
                # t1
                t1, l = np.inf, -1
                for j in range(q):
                    if r[j] > 0.0:
                        tj = lam[j] / r[j]
                        if tj < t1:
                            t1, l = tj, j
                if abs(zz) <= vs:
                    if l < 0:
                        return 1, it_main, it_drop
                    for j in range(q):
                        lam[j] -= t1 * r[j]
                    lam_p = getattr(self, "_lam_p", 0.0) + t1
                    self._lam_p = lam_p
                    drop = True
                    full = False
                else:
                    tt = -sp / zn
                    full = True
                    if l >= 0 and t1 < tt:
                        tt = t1
                        full = False
                    self.U = self.U + tt * z
                    for j in range(q):
                        lam[j] -= tt * r[j]
                    self._lam_p = getattr(self, "_lam_p", 0.0) + tt
                    drop = not full
                if full:
                    if q >= self.qmax:
                        return 4, it_main, it_drop
                    # S grows: [[S, g], [g', nqn]]
                    S2 = np.zeros((q + 1, q + 1))
                    S2[:q, :q] = S
                    S2[:q, q] = g
                    S2[q, :q] = g
                    S2[q, q] = nqn
                    S = S2
                    act.append(p)
                    lam.append(self._lam_p)
                    self._lam_p = 0.0
                    break
                # partial step: recompute the slack of p, drop l
                if abs(zz) > vs:
                    X = self.states()
                    sp = self.slack(cons[p], X)
                del act[l]
                del lam[l]
                S = np.delete(np.delete(S, l, 0), l, 1)
                it_drop += 1
                if it_drop > max_iter:
                    return 3, it_main, it_drop


def axis_problem(wl, b, a, nu):
    A, B, d, x0 = wl["A"][b], wl["B"][b], wl["d"][b], wl["x0"][b]
    nx = A.shape[0]
    N = wl["N"]
    sx = [i for i in range(nx) if i % nu == a]
    Aa = A[np.ix_(sx, sx)]
    Ba = B[sx, a]
    da = d[sx]
    x0a = x0[sx]
    nz = nx + nu
    H = np.zeros((nz, nz))
    h = np.zeros(nz)
    HN = np.zeros((nx, nx))
    hN = np.zeros(nx)
    for c in wl["costs"]:
        w = np.asarray(c["weights"], float)
        p = np.asarray(c["p"], float)
        if c["kind"] == "trajectory":
            M = np.asarray(c["M"], float)
            MN = np.hstack([M, np.zeros((M.shape[0], nu))])
            H += MN.T @ (w[:, None] * MN)
            h += -(MN.T @ (w * p))
            HN += M.T @ (w[:, None] * M)
            hN += -(M.T @ (w * p))
        elif c["kind"] == "control":
            Nn = np.asarray(c["N"], float)
            MN = np.hstack([np.zeros((Nn.shape[0], nx)), Nn])
            H += MN.T @ (w[:, None] * MN)
            h += -(MN.T @ (w * p))
        else:
            raise NotImplementedError
    for c in range(nx, nz):
        H[c, c] += 1e-6
    sz = sx + [nx + a]
    Ha = H[np.ix_(sz, sz)]
    ha = h[sz]
    HNa = HN[np.ix_(sx, sx)]
    hNa = hN[sx]
    rows = []
    ub = np.full(N, np.inf)
    lb = np.full(N, -np.inf)
    idx = 0
    for c in wl["cstrs"]:
        if c["kind"] == "trajectory_bound":
            up = np.asarray(c["upper"], float)
            lo = np.asarray(c["lower"], float)
            assert np.all(np.isinf(lo))
            fin = [i for i in range(nx) if np.isfinite(up[i])]
            for k in range(N + 1):
                for i in fin:
                    if i % nu == a:
                        e = np.zeros(len(sx))
                        e[sx.index(i)] = 1.0
                        rows.append((k, e, 0.0, up[i], idx))
                    idx += 1
        elif c["kind"] == "control_bound":
            ub[:] = np.asarray(c["upper"], float)[a]
            lb[:] = np.asarray(c["lower"], float)[a]
        else:
            raise NotImplementedError
    return Axis(Aa, Ba, da, x0a, Ha, ha, HNa, hNa, N, rows, ub, lb)


def main():
    global VSMALL
    VSMALL = vsmall()
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    v_max = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
    u_max = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
    wl = workloads.com_preview(batch, v_max=v_max, u_max=u_max)
    ref = pyoracle.lmpc_solve_batch(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    nu = 3
    bad = 0
    worst = 0.0
    qhist = {}
    ithist = {}
    for b in range(batch):
        st, im, idr = 0, 1, 0
        U = np.zeros((wl["N"], nu))
        for a in range(nu):
            ax = axis_problem(wl, b, a, nu)
            s, i0, i1 = ax.solve()
            st = max(st, s)
            im += i0 - 1
            idr += i1
            U[:, a] = ax.U
            ithist[i0 - 1 + i1] = ithist.get(i0 - 1 + i1, 0) + 1
        ok = st == ref["status"][b] and (st != 0 or (im == ref["iter"][b][0] and idr == ref["iter"][b][1]))
        if st == 0:
            err = np.max(np.abs(U.reshape(-1) - ref["control"][b]) / np.maximum(np.abs(ref["control"][b]), 1e-3))
            worst = max(worst, err)
            ok = ok and err < 1e-6
        if not ok:
            bad += 1
            if bad <= 10:
                print("instance", b, "axis solver", (st, im, idr), "oracle", (ref["status"][b], tuple(ref["iter"][b])))
    print("batch %d v_max %.2f u_max %.2f: %d differ, worst rel err of U %.2e, mean oracle iters %.2f" % (batch, v_max, u_max, bad, worst, ref["iter"][:, 0].mean()))
    print("per-axis (adds + drops) histogram:", dict(sorted(ithist.items())))


if __name__ == "__main__":
    main()
