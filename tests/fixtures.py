"""The reference's own test fixtures as data (constants only): tests/systems.h of jrl-umi3218/copra.

Each function returns the plain-dict problem description used by the oracle, the emulator and the HIP engine.
"""
import numpy as np

INF = np.inf


def scilab_problem():
    """tests/systems.h:11-38 (Scilab qld example).  Known answer computed from the KKT system of the active set
    {3 equalities, inequality #0} and confirmed with scipy (SURVEY.md 8c / BASELINE.md 5)."""
    return dict(
        Q=np.eye(6), c=np.array([1, 2, 3, 4, 5, 6.]),
        Aeq=np.array([[1, -1, 1, 0, 3, 1], [-1, 0, -3, -4, 5, 6], [2, 5, 3, 0, 1, 0.]]), beq=np.array([1, 2, 3.]),
        Aineq=np.array([[0, 1, 0, 1, 2, -1], [-1, 0, 2, 1, 1, 0.]]), bineq=np.array([-1, 2.5]),
        XL=np.array([-1000, -10000, 0, -1000, -1000, -1000.]), XU=np.array([10000, 100, 1.5, 100, 100, 1000.]),
        x_star=np.array([1.7975426035, -0.3381487238, 0.1633880281, -4.9884022703, 0.6054943277, -3.1155623387]),
        f_star=-14.84324774)


def _falling_mass(N):
    T, mass = 0.005, 5.0  # systems.h:43-44
    A = np.array([[1, T], [0, 1.]])
    B = np.array([[0.5 * T * T / mass], [T / mass]])
    d = np.array([(-9.81 / 2.) * T * T, -9.81 * T])
    return T, mass, A, B, d


def _costs(xcost, xd, N=None):
    M = np.eye(2)
    wx, wu = [10.0, 10000.0], [1e-4]
    if xcost == "mixed":  # TestLMPC.cpp:182-183: MixedCost(M, Zero(2,1), xd) and MixedCost(Zero(1,2), N, ud)
        return [dict(kind="mixed", M=M, N=np.zeros((2, 1)), p=xd, weights=wx),
                dict(kind="mixed", M=np.zeros((1, 2)), N=[[1.0]], p=[2.0], weights=wu)]
    return [dict(kind=xcost, M=M, p=xd, weights=wx), dict(kind="control", N=[[1.0]], p=[2.0], weights=wu)]


def bounded_system(xcost="target", N=300):
    """systems.h:42-90 + TestLMPC.cpp:36-260"""
    T, mass, A, B, d = _falling_mass(N)
    cstrs = [dict(kind="trajectory_bound", lower=[-INF, -INF], upper=[INF, 0.0]),
             dict(kind="control_bound", lower=[-INF], upper=[200.0])]
    return dict(A=A, B=B, d=d, x0=np.array([0.0, -5.0]), N=N, costs=_costs(xcost, [0.0, -1.0]), cstrs=cstrs,
                xd=np.array([0.0, -1.0]), u_upper=200.0, v_upper=0.0)


def ineq_system(xcost="target", N=300):
    """systems.h:94-137 + TestLMPC.cpp:266-360"""
    T, mass, A, B, d = _falling_mass(N)
    cstrs = [dict(kind="trajectory", E=[[0.0, 1.0]], f=[0.0]), dict(kind="control", G=[[1.0]], f=[200.0])]
    return dict(A=A, B=B, d=d, x0=np.array([0.0, -5.0]), N=N, costs=_costs(xcost, [0.0, -1.0]), cstrs=cstrs,
                xd=np.array([0.0, -1.0]), u_upper=200.0, v_upper=0.0)


def mixed_system(xcost="target", N=300):
    """systems.h:141-182 + TestLMPC.cpp:415-587: E x_k + G u_k <= p with E = [0 1], G = 1, p = 200"""
    T, mass, A, B, d = _falling_mass(N)
    cstrs = [dict(kind="mixed", E=[[0.0, 1.0]], G=[[1.0]], f=[200.0])]
    return dict(A=A, B=B, d=d, x0=np.array([0.0, -5.0]), N=N, costs=_costs(xcost, [0.0, -1.0]), cstrs=cstrs,
                xd=np.array([0.0, -1.0]), E=np.array([[0.0, 1.0]]), G=np.array([[1.0]]), p=200.0)


def eq_system(xcost="target", N=300):
    """systems.h:187-229 + TestLMPC.cpp:593-771: position pinned to x0 = 0 -> u_k = m g = 49.05"""
    T, mass, A, B, d = _falling_mass(N)
    E = np.zeros((2, 2))
    E[0, 0] = 1.0
    x0 = np.zeros(2)
    cstrs = [dict(kind="trajectory", E=E, f=x0, ineq=False)]
    return dict(A=A, B=B, d=d, x0=x0, N=N, costs=_costs(xcost, [0.0, 0.0]), cstrs=cstrs, xd=np.zeros(2),
                u_expected=mass * 9.81)


def initial_state_problem(full_size):
    """tests/TestLMPC_InitialState.cpp:29-130: ALL nine cost / constraint classes on A = ones(2,2), B = ones(2,1),
    N = 10, either with per-step entries or with full-size entries produced by autoSpan()."""
    from copra_amd.autospan import autospan_cost, autospan_cstr
    xDim, uDim, factor, N = 2, 1, 10.0, 10
    U, X = (N, N + 1) if full_size else (1, 1)
    ones = np.ones
    costs = [dict(kind="trajectory", M=ones((1, xDim)), p=factor * ones(1 * X)),
             dict(kind="target", M=ones((1, xDim)), p=factor * ones(1)),
             dict(kind="control", N=ones((1, uDim)), p=factor * ones(1 * U)),
             dict(kind="mixed", M=ones((1, xDim)), N=ones((1, uDim)), p=factor * ones(1 * U))]
    cstrs = [dict(kind="trajectory", E=ones((1, xDim)), f=factor * ones(1 * X)),
             dict(kind="control", G=ones((1, uDim)), f=factor * ones(1 * U)),
             dict(kind="mixed", E=ones((1, xDim)), G=ones((1, uDim)), f=factor * ones(1 * U)),
             dict(kind="trajectory_bound", lower=-INF * ones(xDim * X), upper=INF * ones(xDim * X)),
             dict(kind="control_bound", lower=-3.0 * ones(uDim * U), upper=3.0 * ones(uDim * U))]
    costs = [autospan_cost(c) for c in costs]
    for c in costs:
        c["weights"] = np.ones(np.atleast_1d(c["p"]).shape[0])  # ->weight(1)
    cstrs = [autospan_cstr(c) for c in cstrs]
    combi = ones((xDim + uDim, xDim + uDim))
    return dict(A=combi[:xDim, :xDim], B=combi[:xDim, xDim:], d=np.zeros(xDim), x0=np.zeros(xDim), N=N,
                costs=costs, cstrs=cstrs)


def com_walk_problem():
    """binding/python/tests/pyTests.py:341-443 (test_dynamic_walk): CoM system nx=6, nu=3, N=10, T=0.117 with a
    full-size 66x30 ControlConstraint polytope and a TargetCost.  G / h are regenerated from the 7 + 6 distinct
    rows the listing repeats (values copied as data)."""
    T = 0.11699999999999999
    A = np.eye(6)
    A[:3, 3:] = T * np.eye(3)
    B = np.zeros((6, 3))
    B[:3] = 0.006844499999999999 * np.eye(3)
    B[3:] = T * np.eye(3)
    x_init = np.array([1.5842778860957882, 0.3422260214935311, 2.289067474385933, 0., 0., 0.])
    x_goal = np.array([1.627772868473883, 0.4156386515475985, 2.3984423755527136, 0.06745225960685897,
                       0.3882830795737303, 0.06845759848745198])
    blkA = np.array([[-1, 9.946646523934742, -4.870790074510924],
                     [-18.826459196882055, 3.4468275392859393, -1],
                     [-1.374181960437557, -8.028252906078723, -1],
                     [-9.936224732113594, 5.000580301294253, -1],
                     [8.750597187695343, -4.7538557382857105, -1],
                     [18.65430148414319, 1, -5.084871935334947],
                     [2.1775137248880574e-15, -5.443784312220143e-16, 1]])
    hA = np.array([47.76613348420254, 9.80665, 9.80665, 9.80665, 9.806649999999998, 49.86555936465245,
                   9.806649999999994])
    blkB = np.array([[8.750597218241072, -4.753855754313641, -1],
                     [-1.3741819771739483, -8.028252929943818, -1],
                     [-18.82645925631406, 3.4468275193254927, -1],
                     [1.1824397341134247, 7.4638136184143935, -1],
                     [14.006645137157978, -2.2569159229140494, -1],
                     [2.1775137248880578e-15, -5.443784312220144e-16, 1]])
    hB = np.array([9.806650000000007, 9.806650000000008, 9.80665, 9.806650000000001, 9.806650000000007,
                   9.806649999999996])
    N = 10
    G = np.zeros((66, 30))
    h = np.zeros(66)
    r = 0
    for s in range(6):
        G[r:r + 7, 3 * s:3 * s + 3] = blkA
        h[r:r + 7] = hA
        r += 7
    for s in range(6, 10):
        G[r:r + 6, 3 * s:3 * s + 3] = blkB
        h[r:r + 6] = hB
        r += 6
    assert r == 66
    costs = [dict(kind="target", M=np.eye(6), p=-x_goal)]  # pyTests.py:436 (sic: -x_goal)
    cstrs = [dict(kind="control", G=G, f=h)]
    return dict(A=A, B=B, d=np.zeros(6), x0=x_init, N=N, costs=costs, cstrs=cstrs)


def random_dense_qp(rng, n, meq, mineq, tight=0.3):
    """strictly convex QP around a feasible point: equalities, inequalities (some active) and box bounds"""
    M = rng.standard_normal((n, n))
    Q = M @ M.T / n + 0.5 * np.eye(n)
    c = rng.standard_normal(n)
    xf = 0.2 * rng.standard_normal(n)
    Aeq = rng.standard_normal((meq, n))
    Ain = rng.standard_normal((mineq, n))
    return dict(Q=Q, c=c, Aeq=Aeq, beq=Aeq @ xf, Aineq=Ain, bineq=Ain @ xf + tight * rng.random(mineq),
                XL=xf - 0.5 * rng.random(n) - 0.05, XU=xf + 0.5 * rng.random(n) + 0.05)


def nine_class_problem(N):
    """All four cost classes and all five constraint classes (per-step entries) on the falling-mass system of
    systems.h:42-90 -- a well-conditioned counterpart of initial_state_problem() for long horizons"""
    T, mass, A, B, d = _falling_mass(N)
    costs = [dict(kind="trajectory", M=np.eye(2), p=[0.0, -1.0], weights=[10.0, 100.0]),
             dict(kind="target", M=[[1.0, 0.5]], p=[0.2], weights=[50.0]),
             dict(kind="control", N=[[1.0]], p=[2.0], weights=[1e-3]),
             dict(kind="mixed", M=[[0.0, 1.0]], N=[[0.01]], p=[-0.5], weights=[2.0])]
    cstrs = [dict(kind="trajectory", E=[[1.0, 0.2]], f=[0.05]),
             dict(kind="control", G=[[1.0]], f=[150.0]),
             dict(kind="mixed", E=[[0.0, 1.0]], G=[[0.002]], f=[0.1]),
             dict(kind="trajectory_bound", lower=[-INF, -INF], upper=[0.02, 0.5]),  # (reference quirk Q1: no lower rows)
             dict(kind="control_bound", lower=[-20.0], upper=[200.0])]
    return dict(A=A, B=B, d=d, x0=np.array([0.0, -5.0]), N=N, costs=costs, cstrs=cstrs)
