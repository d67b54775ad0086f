"""Synthetic workloads for the BASELINE.json configs.

All random draws come from splitmix64 (vectorised in numpy, trivially reproducible in C) so that the same inputs
can be regenerated anywhere.  System matrices follow the reference's own fixtures:
  * double integrator "falling mass": tests/systems.h:63-83 (T = 0.005, m = 5, gravity bias)
  * CoM preview system: binding/python/tests/pyTests.py:342-359 (A, B, x_init, x_goal, T = 0.117)
Arrays use natural numpy indexing: A[b] is the (nx, nx) state matrix of instance b.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


class SplitMix64:
    def __init__(self, seed):
        self.state = np.uint64(seed)

    def next_u64(self, count):
        with np.errstate(over="ignore"):
            inc = np.uint64(0x9E3779B97F4A7C15)
            idx = np.arange(1, count + 1, dtype=np.uint64)
            z = self.state + idx * inc
            self.state = self.state + np.uint64(count) * inc
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        return z

    def uniform(self, count, lo=0.0, hi=1.0):
        u = (self.next_u64(count) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return lo + (hi - lo) * u

    def normal(self, count, sigma=1.0):
        u1 = self.uniform(count)
        u2 = self.uniform(count)
        u1 = np.maximum(u1, 1e-300)
        return sigma * np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


# pyTests.py:358-359
COM_X_INIT = np.array([1.5842778860957882, 0.3422260214935311, 2.289067474385933, 0.0, 0.0, 0.0])
COM_X_GOAL = np.array([1.627772868473883, 0.4156386515475985, 2.3984423755527136,
                       0.06745225960685897, 0.3882830795737303, 0.06845759848745198])


def double_integrator(batch, N=10, seed=0):
    """BASELINE config 2: batch of falling-mass double integrators (nx=2, nu=1) with a control bound.
    Per-instance mass m ~ U(3,7) (changes B) and x0 = [U(-1,1), U(-6,-4)]; shared costs / bound (systems.h:63-83)."""
    rng = SplitMix64(seed)
    T = 0.005
    m = rng.uniform(batch, 3.0, 7.0)
    A = np.tile(np.array([[1.0, T], [0.0, 1.0]]), (batch, 1, 1))
    B = np.zeros((batch, 2, 1))
    B[:, 0, 0] = 0.5 * T * T / m
    B[:, 1, 0] = T / m
    d = np.tile(np.array([(-9.81 / 2.0) * T * T, -9.81 * T]), (batch, 1))
    x0 = np.stack([rng.uniform(batch, -1.0, 1.0), rng.uniform(batch, -6.0, -4.0)], axis=1)
    costs = [dict(kind="target", M=np.eye(2), p=[0.0, -1.0], weights=[10.0, 10000.0]),
             dict(kind="control", N=[[1.0]], p=[2.0], weights=[1e-4])]
    cstrs = [dict(kind="control_bound", lower=[-np.inf], upper=[200.0])]
    return dict(name="double-integrator (nx=2,nu=1,N=%d) + control bound" % N, A=A, B=B, d=d, x0=x0, N=N,
                costs=costs, cstrs=cstrs)


def com_preview(batch, N=20, seed=1, v_max=0.6, u_max=3.0):
    """BASELINE configs 3/4 (headline): CoM double integrator in 3-D (nx=6, nu=3), per-instance sampling period
    T ~ U(0.08, 0.15) so A and B differ per instance; x0 = x_init + noise; trajectory cost towards x_goal + small
    control cost; 2 inequality objects: upper velocity bound (TrajectoryBoundConstraint, 63 rows; upper-only because
    of reference quirk Q1) and symmetric control bound.  The defaults v_max=0.6, u_max=3.0 leave ~55 % of the
    instances with at least one active constraint (SURVEY.md 8d asks for 30-60 %); v_max=0.25, u_max=1.2 makes every
    instance hit 3..22 constraints (used by the parity tests to stress the active-set loop)."""
    rng = SplitMix64(seed)
    T = rng.uniform(batch, 0.08, 0.15)
    I3 = np.eye(3)
    A = np.zeros((batch, 6, 6))
    B = np.zeros((batch, 6, 3))
    A[:, :3, :3] = I3
    A[:, 3:, 3:] = I3
    A[:, :3, 3:] = T[:, None, None] * I3
    B[:, :3, :] = (0.5 * T * T)[:, None, None] * I3
    B[:, 3:, :] = T[:, None, None] * I3
    d = np.zeros((batch, 6))
    x0 = np.tile(COM_X_INIT, (batch, 1))
    x0[:, :3] += rng.normal(batch * 3, 0.05).reshape(batch, 3)
    x0[:, 3:] += rng.uniform(batch * 3, -0.2, 0.2).reshape(batch, 3)
    inf = np.inf
    costs = [dict(kind="trajectory", M=np.eye(6), p=COM_X_GOAL, weights=[10.0, 10.0, 10.0, 1.0, 1.0, 1.0]),
             dict(kind="control", N=np.eye(3), p=np.zeros(3), weights=[1e-3] * 3)]
    cstrs = [dict(kind="trajectory_bound", lower=[-inf] * 6, upper=[inf, inf, inf, v_max, v_max, v_max]),
             dict(kind="control_bound", lower=[-u_max] * 3, upper=[u_max] * 3)]
    return dict(name="CoM preview (nx=6,nu=3,N=%d) + trajectory & control bounds" % N, A=A, B=B, d=d, x0=x0, N=N,
                costs=costs, cstrs=cstrs)


def jerk_preview(batch, nu=3, N=20, seed=21, v_max=0.6, j_max=20.0, a_max=None):
    """The jerk-controlled CoM model (the cart-table / preview-control model: position, velocity, acceleration per axis, the jerk as control)
    in `nu` dimensions: nx = 3 nu, state i on axis i % nu.  Per-instance sampling period T ~ U(0.08, 0.15); trajectory cost towards a goal +
    small control cost; an upper velocity bound (TrajectoryBoundConstraint, upper-only: reference quirk Q1), a symmetric bound on the jerk,
    optionally an upper bound on the acceleration as well (two rows per axis and step)."""
    rng = SplitMix64(seed)
    T = rng.uniform(batch, 0.08, 0.15)
    I = np.eye(nu)
    nx = 3 * nu
    A = np.zeros((batch, nx, nx))
    B = np.zeros((batch, nx, nu))
    for a in range(3):
        A[:, a * nu:(a + 1) * nu, a * nu:(a + 1) * nu] = I
    A[:, :nu, nu:2 * nu] = T[:, None, None] * I
    A[:, :nu, 2 * nu:] = (0.5 * T * T)[:, None, None] * I
    A[:, nu:2 * nu, 2 * nu:] = T[:, None, None] * I
    B[:, :nu, :] = (T ** 3 / 6.0)[:, None, None] * I
    B[:, nu:2 * nu, :] = (0.5 * T * T)[:, None, None] * I
    B[:, 2 * nu:, :] = T[:, None, None] * I
    d = np.zeros((batch, nx))
    goal = np.concatenate([COM_X_GOAL[:nu], np.zeros(2 * nu)])
    x0 = np.zeros((batch, nx))
    x0[:, :nu] = COM_X_INIT[:nu] + rng.normal(batch * nu, 0.05).reshape(batch, nu)
    x0[:, nu:2 * nu] = rng.uniform(batch * nu, -0.2, 0.2).reshape(batch, nu)
    x0[:, 2 * nu:] = rng.uniform(batch * nu, -0.5, 0.5).reshape(batch, nu)
    inf = np.inf
    costs = [dict(kind="trajectory", M=np.eye(nx), p=goal, weights=[10.0] * nu + [1.0] * nu + [0.1] * nu),
             dict(kind="control", N=np.eye(nu), p=np.zeros(nu), weights=[1e-4] * nu)]
    upper = [inf] * nu + [v_max] * nu + ([inf] * nu if a_max is None else [a_max] * nu)
    cstrs = [dict(kind="trajectory_bound", lower=[-inf] * nx, upper=upper),
             dict(kind="control_bound", lower=[-j_max] * nu, upper=[j_max] * nu)]
    return dict(name="jerk-controlled CoM preview (nx=%d,nu=%d,N=%d) + trajectory & control bounds" % (nx, nu, N), A=A, B=B, d=d, x0=x0, N=N,
                costs=costs, cstrs=cstrs)


def kinematic_preview(batch, nu=3, N=20, seed=31, p_max=0.55, u_max=1.5):
    """A velocity-controlled point in `nu` dimensions (the kinematic model of mobile-robot MPC): ONE state per control, x+ = a x + T u + d per axis
    with a ~ U(0.95, 1.0) and T ~ U(0.08, 0.15) per instance; trajectory cost towards a goal + small control cost; an upper bound on the
    position (TrajectoryBoundConstraint) and a symmetric bound on the velocity command."""
    rng = SplitMix64(seed)
    T = rng.uniform(batch, 0.08, 0.15)
    a = rng.uniform(batch, 0.95, 1.0)
    I = np.eye(nu)
    A = a[:, None, None] * I
    B = T[:, None, None] * I
    d = np.tile(0.002 * np.arange(1, nu + 1), (batch, 1))
    x0 = rng.normal(batch * nu, 0.15).reshape(batch, nu)  # (inside the position bound: a state row violated by x0 itself is "no solution")
    goal = np.array([0.5, 0.4, 0.45])[:nu]  # (inside the position bound: it is active where the unconstrained response overshoots)
    costs = [dict(kind="trajectory", M=np.eye(nu), p=goal, weights=[10.0, 8.0, 6.0][:nu]),
             dict(kind="control", N=np.eye(nu), p=np.zeros(nu), weights=[1e-2] * nu)]
    cstrs = [dict(kind="trajectory_bound", lower=[-np.inf] * nu, upper=[p_max] * nu),
             dict(kind="control_bound", lower=[-u_max] * nu, upper=[u_max] * nu)]
    return dict(name="kinematic point (nx=%d,nu=%d,N=%d) + position & velocity bounds" % (nu, nu, N), A=np.ascontiguousarray(A), B=np.ascontiguousarray(B),
                d=d, x0=x0, N=N, costs=costs, cstrs=cstrs)


def axis_major(wl):
    """The same controller with its states in AXIS-MAJOR order -- x = (p_x, v_x, p_y, v_y, ..) instead of (p, v) --: a workload of com_preview /
    jerk_preview (identity M, per-step bounds) with systems, costs and bounds permuted.  The engine sees the order from the first system it is
    given."""
    nx, nu = wl["A"].shape[1], wl["B"].shape[2]
    nxa = nx // nu
    perm = np.array([c + nu * a for c in range(nu) for a in range(nxa)])  # new state j is old state perm[j]
    out = dict(wl)
    out["A"] = np.ascontiguousarray(wl["A"][:, perm][:, :, perm])
    out["B"] = np.ascontiguousarray(wl["B"][:, perm, :])
    out["d"] = np.ascontiguousarray(wl["d"][:, perm])
    out["x0"] = np.ascontiguousarray(wl["x0"][:, perm])
    costs = []
    for c in wl["costs"]:
        c = dict(c)
        if c["kind"] in ("trajectory", "target"):
            assert np.array_equal(np.asarray(c["M"]), np.eye(nx))
            c["p"] = np.asarray(c["p"])[perm]
            c["weights"] = list(np.asarray(c["weights"])[perm])
        costs.append(c)
    cstrs = []
    for c in wl["cstrs"]:
        c = dict(c)
        if c["kind"] == "trajectory_bound":
            c["lower"] = list(np.asarray(c["lower"])[perm])
            c["upper"] = list(np.asarray(c["upper"])[perm])
        cstrs.append(c)
    out["costs"], out["cstrs"] = costs, cstrs
    out["name"] = wl["name"] + " (axis-major states)"
    return out


def long_horizon_initial_state(batch, N=50, seed=3, v_max=0.5, u_max=2.0, R_diag=1e-6, T=0.05):
    """BASELINE config 5 as SURVEY.md section 8(d) specifies it: InitialStateLMPC on a 6-DoF double integrator
    (nx=12, nu=6, N=50, T=0.05; 312 decision variables [x0; U]) -- the long-horizon case that does not fit the LDS.
      x = [p(6); v(6)];  TrajectoryCost towards a goal + ControlCost;
      * MixedConstraint (inequality, r=6):  v_k + T u_k <= v_max                            -> 300 rows
      * one FULL-SIZE TrajectoryConstraint (equality): terminal velocity = 0 (6 rows x 612 columns)
      * ControlBoundConstraint |u| <= u_max
      * resetInitialStateCost(R = 1e-6 I, r = 0), x0 bounds = x0_nom +- 0.05 (cf. TestLMPC_InitialState.cpp:356-361)
    Per instance: x0_nom (positions N(0, 0.05), velocities U(-0.3, 0.3)); A, B are shared in value but still handed over
    per instance, as everywhere in this engine."""
    rng = SplitMix64(seed)
    I6 = np.eye(6)
    A1 = np.zeros((12, 12))
    B1 = np.zeros((12, 6))
    A1[:6, :6] = I6
    A1[:6, 6:] = T * I6
    A1[6:, 6:] = I6
    B1[:6, :] = 0.5 * T * T * I6
    B1[6:, :] = T * I6
    A = np.tile(A1, (batch, 1, 1))
    B = np.tile(B1, (batch, 1, 1))
    d = np.zeros((batch, 12))
    x0 = np.zeros((batch, 12))
    x0[:, :6] = rng.normal(batch * 6, 0.05).reshape(batch, 6)
    x0[:, 6:] = rng.uniform(batch * 6, -0.3, 0.3).reshape(batch, 6)
    goal = np.array([0.8, -0.6, 0.5, 0.7, -0.4, 0.3])
    Mpos = np.hstack([I6, np.zeros((6, 6))])
    costs = [dict(kind="trajectory", M=Mpos, p=goal, weights=[10.0] * 6),
             dict(kind="control", N=I6, p=np.zeros(6), weights=[1e-2] * 6)]
    X = 12 * (N + 1)
    Eterm = np.zeros((6, X))
    Eterm[:, X - 6:] = I6  # velocity block of x_N
    cstrs = [dict(kind="mixed", E=np.hstack([np.zeros((6, 6)), I6]), G=T * I6, f=[v_max] * 6),
             dict(kind="trajectory", E=Eterm, f=np.zeros(6), ineq=False),
             dict(kind="control_bound", lower=[-u_max] * 6, upper=[u_max] * 6)]
    ist = dict(R=R_diag * np.eye(12), r=np.zeros(12), x0lb=x0 - 0.05, x0ub=x0 + 0.05)
    return dict(name="InitialStateLMPC (nx=12,nu=6,N=%d): mixed ineq + full-size terminal equality + control bounds" % N,
                A=A, B=B, d=d, x0=x0, N=N, costs=costs, cstrs=cstrs, initial_state=ist)


def wide_state_initial_state(batch, nx=18, nu=2, N=40, seed=11, u_max=1.5):
    """InitialStateLMPC with MORE than 16 states (the condensed kernels' limit; stage-wise, so the Riccati interior-point kernel
    takes it): a stable random system shared in value by the batch, TrajectoryCost on the first six states + ControlCost,
    control bounds, one upper TrajectoryBound on state 0, x0 bounds +- 0.05, R = 1e-2 I."""
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((nx, nx)))
    A1 = 0.97 * Q
    B1 = 0.3 * rng.standard_normal((nx, nu))
    d1 = 0.01 * rng.standard_normal(nx)
    A = np.tile(A1, (batch, 1, 1))
    B = np.tile(B1, (batch, 1, 1))
    d = np.tile(d1, (batch, 1))
    x0 = 0.5 * rng.standard_normal((batch, nx))
    M = np.eye(nx)[:6]
    upper = np.full(nx, np.inf)
    upper[0] = 1.5
    costs = [dict(kind="trajectory", M=M, p=np.zeros(6), weights=[5.0] * 6),
             dict(kind="control", N=np.eye(nu), p=np.zeros(nu), weights=[1e-2] * nu)]
    cstrs = [dict(kind="trajectory_bound", lower=[-np.inf] * nx, upper=list(upper)),
             dict(kind="control_bound", lower=[-u_max] * nu, upper=[u_max] * nu)]
    ist = dict(R=1e-2 * np.eye(nx), r=np.zeros(nx), x0lb=x0 - 0.05, x0ub=x0 + 0.05)
    return dict(name="InitialStateLMPC (nx=%d,nu=%d,N=%d): more than 16 states" % (nx, nu, N), A=A, B=B, d=d, x0=x0, N=N,
                costs=costs, cstrs=cstrs, initial_state=ist)
