"""Published worked examples of strictly convex QPs -- third-party numeric pins of the solver (round-3 verdict: "parity unpinned"
rested on one 3-variable problem).  The reference's own tests hold no numeric vector at the QP boundary (tests/TestSolvers.cpp:25-33
checks SI_solve == true and SI_fail() == 0 only), eigen-quadprog is absent and unpinned; what CAN be pinned from outside is the
algorithm (Goldfarb-Idnani as qpgen2 realises it) on problems whose answers are in print.

Every entry is in SolverInterface form (include/SolverInterface.h:54-80):  min 1/2 x'Qx + c'x,  Aeq x = beq,  Aineq x <= bineq,
XL <= x <= XU, with `x_star` / `f_star` as PUBLISHED (digits as printed) and, where the source prints them, the iteration counts
of qpgen2.  tests/test_oracle.py checks each one three ways: the oracle against the published digits, the oracle against an
independent least-distance (NNLS) solve + exact KKT polish, and the optimality certificate of tests/truth.py; the emulator
(tests/test_emu_kernels.py) and the device (tests/test_gpu_parity.py) run the same list through plug-in point 1.

Sources (all public, cited from memory of the printed problem data; the KKT certificate in the tests does not depend on the
citation being exact -- a wrong digit here would fail it):
  r_solve_qp      R package `quadprog`, help page of solve.QP (the qpgen2 code eigen-quadprog wraps): $solution, $value, $iterations
  goldfarb_idnani D. Goldfarb, A. Idnani, "A numerically stable dual method for solving strictly convex quadratic programs",
                  Math. Programming 27 (1983), the paper's numerical example; also the self-test shipped with QuadProg++
  quadprogpp      QuadProg++ (L. Di Gaspero) documentation example: the same G with an equality added
  matlab_ineq     MathWorks documentation of `quadprog`, "Quadratic program with linear constraints"
  matlab_eq       ... "Quadratic program with linear equality constraint"
  nocedal_wright  J. Nocedal, S. Wright, Numerical Optimization (2nd ed.), Example 16.4 (active-set method)
  cvxopt_doc      CVXOPT user's guide, "Quadratic Programming" example
  scilab_qld      Scilab documentation of `qld` -- the Problem fixture of the reference's tests/systems.h:9-38
  hs21, hs35, hs76  W. Hock, K. Schittkowski, Test Examples for Nonlinear Programming Codes (1981), problems 21, 35 (Beale), 76
"""
import numpy as np

BIG = np.finfo(float).max


def _qp(Q, c, Aeq=None, beq=None, Aineq=None, bineq=None, XL=None, XU=None, x_star=None, f_star=None, digits=6, iterations=None):
    Q = np.asarray(Q, dtype=float)
    n = Q.shape[0]
    return dict(Q=Q, c=np.asarray(c, dtype=float),
                Aeq=np.zeros((0, n)) if Aeq is None else np.atleast_2d(np.asarray(Aeq, dtype=float)),
                beq=np.zeros(0) if beq is None else np.atleast_1d(np.asarray(beq, dtype=float)),
                Aineq=np.zeros((0, n)) if Aineq is None else np.atleast_2d(np.asarray(Aineq, dtype=float)),
                bineq=np.zeros(0) if bineq is None else np.atleast_1d(np.asarray(bineq, dtype=float)),
                XL=np.full(n, -BIG) if XL is None else np.asarray(XL, dtype=float),
                XU=np.full(n, BIG) if XU is None else np.asarray(XU, dtype=float),
                x_star=np.asarray(x_star, dtype=float), f_star=f_star, tol=0.6 * 10.0 ** (-digits), iterations=iterations)


PUBLISHED = {
    # t(Amat) x >= bvec with Amat = [[-4,-3,0],[2,1,0],[0,-2,1]] (column per constraint) -> Aineq = -t(Amat), bineq = -bvec
    "r_solve_qp": _qp(np.eye(3), [0.0, -5.0, 0.0],
                      Aineq=-np.array([[-4.0, -3.0, 0.0], [2.0, 1.0, 0.0], [0.0, -2.0, 1.0]]), bineq=[8.0, -2.0, 0.0],
                      x_star=[0.4761905, 1.0476190, 2.0952381], f_star=-2.380952, digits=7, iterations=(3, 0)),
    # min 1/2 x'Gx + a'x, G = [[4,-2],[-2,4]], a = (6, 0); x1 >= 0, x2 >= 0, x1 + x2 >= 2  ->  x = (0.5, 1.5), f = 6.5
    "goldfarb_idnani": _qp([[4.0, -2.0], [-2.0, 4.0]], [6.0, 0.0], Aineq=[[-1.0, 0.0], [0.0, -1.0], [-1.0, -1.0]],
                           bineq=[0.0, 0.0, -2.0], x_star=[0.5, 1.5], f_star=6.5, digits=10),
    # the same with x1 + x2 = 3  ->  x = (1, 2), f = 12
    "quadprogpp": _qp([[4.0, -2.0], [-2.0, 4.0]], [6.0, 0.0], Aeq=[[1.0, 1.0]], beq=[3.0],
                      Aineq=[[-1.0, 0.0], [0.0, -1.0], [-1.0, -1.0]], bineq=[0.0, 0.0, -2.0], x_star=[1.0, 2.0], f_star=12.0, digits=10),
    # H = [1 -1; -1 2], f = [-2; -6], A = [1 1; -1 2; 2 1], b = [2; 2; 3]  ->  x = (0.6667, 1.3333), fval = -8.2222
    "matlab_ineq": _qp([[1.0, -1.0], [-1.0, 2.0]], [-2.0, -6.0], Aineq=[[1.0, 1.0], [-1.0, 2.0], [2.0, 1.0]], bineq=[2.0, 2.0, 3.0],
                       x_star=[0.6667, 1.3333], f_star=-8.2222, digits=4),
    # the same H, f with Aeq = [1 1], beq = 0  ->  x = (-0.8, 0.8), fval = -1.6
    "matlab_eq": _qp([[1.0, -1.0], [-1.0, 2.0]], [-2.0, -6.0], Aeq=[[1.0, 1.0]], beq=[0.0], x_star=[-0.8, 0.8], f_star=-1.6, digits=4),
    # min (x1 - 1)^2 + (x2 - 2.5)^2; x1 - 2x2 + 2 >= 0, -x1 - 2x2 + 6 >= 0, -x1 + 2x2 + 2 >= 0, x >= 0  ->  x = (1.4, 1.7)
    "nocedal_wright": _qp(2.0 * np.eye(2), [-2.0, -5.0], Aineq=[[-1.0, 2.0], [1.0, 2.0], [1.0, -2.0]], bineq=[2.0, 6.0, 2.0],
                          XL=[0.0, 0.0], x_star=[1.4, 1.7], f_star=-6.45, digits=10),
    # min 2 x1^2 + x2^2 + x1 x2 + x1 + x2; x >= 0, x1 + x2 = 1  ->  x = (0.25, 0.75), objective 1.875
    "cvxopt_doc": _qp([[4.0, 1.0], [1.0, 2.0]], [1.0, 1.0], Aeq=[[1.0, 1.0]], beq=[1.0], XL=[0.0, 0.0], x_star=[0.25, 0.75],
                      f_star=1.875, digits=7),
    # tests/systems.h:9-38 (values as SURVEY.md 8c quotes the Scilab documentation)
    "scilab_qld": _qp(np.eye(6), [1.0, 2.0, 3.0, 4.0, 5.0, 6.0],
                      Aeq=[[1, -1, 1, 0, 3, 1], [-1, 0, -3, -4, 5, 6], [2, 5, 3, 0, 1, 0]], beq=[1.0, 2.0, 3.0],
                      Aineq=[[0, 1, 0, 1, 2, -1], [-1, 0, 2, 1, 1, 0]], bineq=[-1.0, 2.5],
                      XL=[-1000, -10000, 0, -1000, -1000, -1000], XU=[10000, 100, 1.5, 100, 100, 1000],
                      x_star=[1.7975426, -0.3381487, 0.1633880, -4.9884023, 0.6054943, -3.1155623], f_star=-14.843248, digits=6),
    # HS21: min 0.01 x1^2 + x2^2 - 100; 10 x1 - x2 >= 10, 2 <= x1 <= 50, -50 <= x2 <= 50  ->  x = (2, 0), f = -99.96 (constant dropped: 0.04)
    "hs21": _qp([[0.02, 0.0], [0.0, 2.0]], [0.0, 0.0], Aineq=[[-10.0, 1.0]], bineq=[-10.0], XL=[2.0, -50.0], XU=[50.0, 50.0],
                x_star=[2.0, 0.0], f_star=0.04, digits=10),
    # HS35 (Beale): min 9 - 8x1 - 6x2 - 4x3 + 2x1^2 + 2x2^2 + x3^2 + 2x1x2 + 2x1x3; x1 + x2 + 2x3 <= 3, x >= 0
    #   ->  x = (4/3, 7/9, 4/9), f = 1/9 (with the constant 9: here f_star = 1/9 - 9)
    "hs35": _qp([[4.0, 2.0, 2.0], [2.0, 4.0, 0.0], [2.0, 0.0, 2.0]], [-8.0, -6.0, -4.0], Aineq=[[1.0, 1.0, 2.0]], bineq=[3.0],
                XL=[0.0, 0.0, 0.0], x_star=[1.3333333, 0.7777778, 0.4444444], f_star=1.0 / 9.0 - 9.0, digits=7),
    # HS76: min x1^2 + 0.5x2^2 + x3^2 + 0.5x4^2 - x1x3 + x3x4 - x1 - 3x2 + x3 - x4;
    #   x1 + 2x2 + x3 + x4 <= 5, 3x1 + x2 + 2x3 - x4 <= 4, x2 + 4x3 >= 1.5, x >= 0  ->  x = (0.2727273, 2.090909, 0, 0.5454545), f = -4.681818
    "hs76": _qp([[2.0, 0.0, -1.0, 0.0], [0.0, 1.0, 0.0, 0.0], [-1.0, 0.0, 2.0, 1.0], [0.0, 0.0, 1.0, 1.0]], [-1.0, -3.0, 1.0, -1.0],
                Aineq=[[1.0, 2.0, 1.0, 1.0], [3.0, 1.0, 2.0, -1.0], [0.0, -1.0, -4.0, 0.0]], bineq=[5.0, 4.0, -1.5], XL=[0.0] * 4,
                x_star=[0.2727273, 2.090909, 0.0, 0.5454545], f_star=-4.681818, digits=6),
}


def objective(qp, x):
    return 0.5 * x @ qp["Q"] @ x + qp["c"] @ x


def random_qp(rng, kind):
    """one random strictly convex QP of the randomized differential (tests/test_oracle.py): `kind` in
      generic     random constraints, some active at the optimum
      degenerate  duplicated rows, a row that is a positive combination of two others, a bound equal to a row
      pinned      lb == ub on some variables (what InitialStateLMPC's default x0 bounds are)
      equality    a few equality rows, one of them identically zero (EqSystem's zero rows, tests/systems.h:187-229)
      infeasible  contradicting rows  a'x <= -1, -a'x <= -1   (SI_fail() == 1 expected)
      not_pd      an indefinite Q  (SI_fail() == 2 expected)"""
    n = int(rng.integers(2, 13))
    G = rng.standard_normal((n + 2, n))
    Q = G.T @ G / n + 10.0 ** rng.uniform(-4, 0) * np.eye(n)
    c = rng.standard_normal(n) * 10.0 ** rng.uniform(-1, 1)
    mi = int(rng.integers(0, 2 * n + 1))
    Aineq = rng.standard_normal((mi, n))
    x_in = rng.standard_normal(n)  # a point the inequality rows leave strictly feasible
    bineq = Aineq @ x_in + rng.uniform(0.05, 2.0, mi)
    XL, XU = np.full(n, -BIG), np.full(n, BIG)
    bounded = rng.random(n) < 0.5
    XL[bounded] = x_in[bounded] - rng.uniform(0.05, 1.5, bounded.sum())
    XU[bounded] = x_in[bounded] + rng.uniform(0.05, 1.5, bounded.sum())
    Aeq, beq = np.zeros((0, n)), np.zeros(0)
    if kind == "degenerate" and mi >= 2:
        Aineq = np.vstack([Aineq, Aineq[0], 0.5 * Aineq[0] + 2.0 * Aineq[1]])
        bineq = np.concatenate([bineq, [bineq[0]], [0.5 * bineq[0] + 2.0 * bineq[1]]])
        j = int(rng.integers(0, n))
        e = np.zeros(n)
        e[j] = 1.0
        XU[j] = x_in[j] + 0.3
        Aineq, bineq = np.vstack([Aineq, e]), np.concatenate([bineq, [XU[j]]])
    elif kind == "pinned":
        pins = rng.random(n) < 0.3
        XL[pins] = XU[pins] = x_in[pins]
    elif kind == "equality":
        me = int(rng.integers(1, max(2, n // 2)))
        Aeq = np.vstack([rng.standard_normal((me, n)), np.zeros((1, n))])
        beq = np.concatenate([Aeq[:me] @ x_in, [0.0]])
    elif kind == "infeasible":
        a = rng.standard_normal(n)
        Aineq, bineq = np.vstack([Aineq, a, -a]), np.concatenate([bineq, [-1.0, -1.0]])
    elif kind == "not_pd":
        Q = Q - (np.linalg.eigvalsh(Q)[0] + 0.5) * np.eye(n)  # smallest eigenvalue -0.5
    return dict(Q=Q, c=c, Aeq=Aeq, beq=beq, Aineq=Aineq, bineq=bineq, XL=XL, XU=XU, kind=kind)
