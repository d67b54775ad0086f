"""Random LMPC controllers for differential tests (device or emulator against the oracle): random shapes (nx 1..7, nu 1..3, N 2..24),
random per-instance systems, and a random mix of the reference's four cost classes and five constraint classes, each as a per-step or
as a full-size entry (block-diagonal as AutoSpan produces it, or -- costs and rows -- dense across the steps).  TEST INFRASTRUCTURE.

Every constraint is built around the initial states so that step 0 is feasible (reference quirk Q5: the rows of step 0 are part of the
problem); whether the whole horizon is feasible is left to chance -- status 1 is a result like any other and must agree.
Lower trajectory bounds are left at -inf (reference quirk Q1)."""
import numpy as np


def _blockdiag(M, steps, add_cols=0):
    M = np.atleast_2d(np.asarray(M, dtype=float))
    out = np.kron(np.eye(steps), M)
    if add_cols:
        out = np.hstack([out, np.zeros((out.shape[0], add_cols))])
    return out


def make(seed, batch=48, max_vars=64, shape=None):
    """-> dict(nx, nu, N, A, B, d, x0, costs, cstrs, forms, initial_state) -- forms names what was drawn (for failure messages);
    initial_state: None, or dict(R, r, x0lb, x0ub) for an InitialStateLMPC over the same pieces (tests decide whether they use it)"""
    rng = np.random.default_rng(seed)
    nx = int(rng.integers(1, 8))
    nu = int(rng.integers(1, 4))
    N = int(rng.integers(2, 25))
    while nu * N > max_vars:
        N -= 1
    if shape is not None:  # (a given (nx, nu, N): the draws above are made all the same, the structure below does not depend on them)
        nx, nu, N = shape
    X, U = nx * (N + 1), nu * N
    forms = []
    # systems: per-instance perturbations of a contraction-ish A (spectral radius about 1), B full rank
    A0 = np.eye(nx) + 0.15 * rng.standard_normal((nx, nx))
    A0 /= max(1.0, np.abs(np.linalg.eigvals(A0)).max() / 1.02)
    B0 = 0.4 * rng.standard_normal((nx, nu))
    A = A0[None] + 0.02 * rng.standard_normal((batch, nx, nx))
    B = B0[None] + 0.02 * rng.standard_normal((batch, nx, nu))
    d = 0.02 * rng.standard_normal((batch, nx))
    x0 = rng.standard_normal((batch, nx))
    costs = []

    def weights(r):
        return rng.uniform(0.5, 5.0, r)

    # --- costs: always a control cost (keeps the Hessian well conditioned), then 1..2 of the others
    full_u = rng.random() < 0.3
    Nm = np.eye(nu) if rng.random() < 0.6 else np.eye(nu) + 0.3 * rng.standard_normal((nu, nu))
    cu = dict(kind="control", N=Nm, p=0.2 * rng.standard_normal(nu), weights=rng.uniform(0.05, 0.5, nu))
    if full_u:
        pk = np.tile(cu["p"], N) + (0.05 * rng.standard_normal(U) if rng.random() < 0.5 else 0.0)
        cu = dict(kind="control", N=_blockdiag(Nm, N), p=pk, weights=np.tile(cu["weights"], N))
        forms.append("ucost-full")
    costs.append(cu)
    for _ in range(int(rng.integers(1, 3))):
        kind = rng.choice(["trajectory", "target", "mixed"], p=[0.55, 0.2, 0.25])
        r = int(rng.integers(1, nx + 1))
        M = np.eye(nx)[:r] if rng.random() < 0.5 else rng.standard_normal((r, nx))
        p = rng.standard_normal(r)
        w = weights(r)
        if kind == "target":
            costs.append(dict(kind="target", M=M, p=p, weights=w))
            forms.append("target")
        elif kind == "trajectory":
            style = rng.choice(["step", "blockdiag", "dense"], p=[0.5, 0.35, 0.15])
            if style == "step":
                costs.append(dict(kind="trajectory", M=M, p=p, weights=w))
            elif style == "blockdiag":  # what AutoSpan produces, optionally with a reference TRAJECTORY
                pk = np.tile(p, N + 1) + (0.1 * rng.standard_normal(r * (N + 1)) if rng.random() < 0.6 else 0.0)
                costs.append(dict(kind="trajectory", M=_blockdiag(M, N + 1), p=pk, weights=np.tile(w, N + 1)))
            else:  # rows that couple the steps
                R = int(rng.integers(1, 9))
                costs.append(dict(kind="trajectory", M=0.3 * rng.standard_normal((R, X)), p=rng.standard_normal(R), weights=weights(R)))
            forms.append("xcost-" + style)
        else:
            Nn = 0.5 * rng.standard_normal((r, nu))
            if rng.random() < 0.6:
                costs.append(dict(kind="mixed", M=M, N=Nn, p=p, weights=w))
                forms.append("mixed-step")
            else:
                costs.append(dict(kind="mixed", M=_blockdiag(M, N, nx), N=_blockdiag(Nn, N), p=np.tile(p, N), weights=np.tile(w, N)))
                forms.append("mixed-full")
    # --- constraints
    cstrs = []
    x0max = np.abs(x0).max(axis=0)
    if rng.random() < 0.7:
        lo = np.where(rng.random(nu) < 0.8, -rng.uniform(0.1, 1.5, nu), -np.inf)
        hi = np.where(rng.random(nu) < 0.8, rng.uniform(0.1, 1.5, nu), np.inf)
        if rng.random() < 0.25:
            cstrs.append(dict(kind="control_bound", lower=np.tile(lo, N), upper=np.tile(hi, N) + 0.1 * rng.random(U)))
            forms.append("ubound-full")
        else:
            cstrs.append(dict(kind="control_bound", lower=lo, upper=hi))
            forms.append("ubound")
    if rng.random() < 0.5:
        hi = np.where(rng.random(nx) < 0.6, x0max + rng.uniform(0.02, 0.6, nx), np.inf)
        if np.isfinite(hi).any():
            cstrs.append(dict(kind="trajectory_bound", lower=[-np.inf] * nx, upper=hi))
            forms.append("xbound")
    if rng.random() < 0.4:
        r = int(rng.integers(1, 3))
        E = rng.standard_normal((r, nx))
        f = np.abs(E @ x0.T).max(axis=1) + rng.uniform(0.1, 2.0, r)
        if rng.random() < 0.7:
            cstrs.append(dict(kind="trajectory", E=E, f=f, ineq=True))
            forms.append("xrow")
        else:
            cstrs.append(dict(kind="trajectory", E=_blockdiag(E, N + 1), f=np.tile(f, N + 1), ineq=True))
            forms.append("xrow-full")
    if rng.random() < 0.3:
        r = int(rng.integers(1, 3))
        G = rng.standard_normal((r, nu))
        f = rng.uniform(0.5, 3.0, r)
        if rng.random() < 0.7:
            cstrs.append(dict(kind="control", G=G, f=f, ineq=True))
            forms.append("urow")
        else:
            cstrs.append(dict(kind="control", G=_blockdiag(G, N), f=np.tile(f, N), ineq=True))
            forms.append("urow-full")
    if rng.random() < 0.3:
        r = int(rng.integers(1, 3))
        E, G = 0.5 * rng.standard_normal((r, nx)), rng.standard_normal((r, nu))
        f = np.abs(E @ x0.T).max(axis=1) + rng.uniform(0.5, 3.0, r)
        if rng.random() < 0.7:
            cstrs.append(dict(kind="mixed", E=E, G=G, f=f, ineq=True))
            forms.append("xurow")
        else:
            cstrs.append(dict(kind="mixed", E=_blockdiag(E, N, nx), G=_blockdiag(G, N), f=np.tile(f, N), ineq=True))
            forms.append("xurow-full")
    if rng.random() < 0.25 and nu * N >= 4:  # an equality: a terminal component, or a control row at one step
        if rng.random() < 0.5:
            E = np.zeros((1, X))
            E[0, X - nx + int(rng.integers(0, nx))] = 1.0
            cstrs.append(dict(kind="trajectory", E=E, f=[0.2 * rng.standard_normal()], ineq=False))
            forms.append("terminal-eq")
        else:
            G = np.zeros((1, U))
            k = int(rng.integers(0, N))
            G[0, k * nu:(k + 1) * nu] = rng.standard_normal(nu)
            cstrs.append(dict(kind="control", G=G, f=[0.1 * rng.standard_normal()], ineq=False))
            forms.append("u-eq-1step")
    if not cstrs:
        cstrs.append(dict(kind="control_bound", lower=[-1.0] * nu, upper=[1.0] * nu))
        forms.append("ubound")
    # InitialStateLMPC (drawn last: the controllers above do not depend on it): x0 a decision variable in a box around the nominal state
    ist = None
    if rng.random() < 0.25 and (shape is not None or nx + nu * N <= max_vars):
        half = rng.uniform(0.05, 0.3, nx)
        ist = dict(R=np.diag(rng.uniform(0.5, 5.0, nx)), r=0.1 * rng.standard_normal(nx), x0lb=x0 - half, x0ub=x0 + half)
    return dict(nx=nx, nu=nu, N=N, A=A, B=B, d=d, x0=x0, costs=costs, cstrs=cstrs, forms=forms, initial_state=ist)


def make_chain1(seed, batch):
    """Random controllers with ONE state per control (kinematic models: x+ = a x + T u + d per axis) in two and three dimensions -- the shapes of
    the (instance, axis)-per-lane solver's builds for nx = nu: random horizon (two dimensions: up to 31), per-instance a and T, a goal or a
    reference trajectory, optional target cost, position bounds / rows, velocity-command bounds, mixed rows; a dense state row now and then."""
    rng = np.random.default_rng([seed, 4177])
    dim = int(rng.choice([2, 3], p=[0.45, 0.55]))
    nx = nu = dim
    N = int(rng.integers(4, (31 if dim == 2 else 20) + 1))
    forms = []
    T = rng.uniform(0.08, 0.15, batch)
    a = rng.uniform(0.93, 1.0, batch)
    I = np.eye(dim)
    A = np.ascontiguousarray(a[:, None, None] * I)
    B = np.ascontiguousarray(T[:, None, None] * I)
    d = np.zeros((batch, nx)) if rng.random() < 0.5 else np.tile(0.004 * rng.standard_normal(nx), (batch, 1))
    p_max = float(rng.uniform(0.5, 0.9))
    u_max = float(rng.uniform(0.6, 2.0))
    x0 = rng.uniform(-0.8 * p_max, 0.8 * p_max, (batch, dim))
    goal = rng.uniform(-0.45, 0.45, dim)
    wx = rng.uniform(4.0, 15.0, dim)
    costs = []
    if rng.random() < 0.35:
        pk = goal[None, :] * np.linspace(0.3, 1.0, N + 1)[:, None] + 0.01 * rng.standard_normal((N + 1, nx))
        costs.append(dict(kind="trajectory", M=_blockdiag(np.eye(nx), N + 1), p=pk.reshape(-1), weights=np.tile(wx, N + 1)))
        forms.append("xref")
    else:
        costs.append(dict(kind="trajectory", M=np.eye(nx), p=goal, weights=wx))
        forms.append("xcost")
    costs.append(dict(kind="control", N=np.eye(nu), p=np.zeros(nu), weights=[float(rng.uniform(1e-3, 5e-2))] * nu))
    if rng.random() < 0.3:
        costs.append(dict(kind="target", M=np.eye(nx), p=goal, weights=5.0 * wx))
        forms.append("target")
    cstrs = []
    inf = np.inf
    if rng.random() < 0.8:
        cstrs.append(dict(kind="trajectory_bound", lower=[-inf] * nx, upper=[p_max] * dim))
        forms.append("pbound")
    if rng.random() < 0.85:
        cstrs.append(dict(kind="control_bound", lower=[-u_max] * nu, upper=[u_max] * nu))
        forms.append("ubound")
    if rng.random() < 0.3:
        cstrs.append(dict(kind="trajectory", E=-np.eye(dim), f=[p_max] * dim, ineq=True))  # the lower position limit as rows
        forms.append("-p rows")
    elif rng.random() < 0.2:
        cstrs.append(dict(kind="mixed", E=np.eye(dim), G=0.05 * np.eye(dim), f=[p_max * 1.05] * dim, ineq=True))
        forms.append("p+Tu")
    if rng.random() < 0.1:
        E = rng.standard_normal((1, nx))
        cstrs.append(dict(kind="trajectory", E=E, f=[float(np.abs(E @ x0.T).max() + rng.uniform(0.5, 1.5))], ineq=True))
        forms.append("dense-x")
    if not cstrs:
        cstrs.append(dict(kind="control_bound", lower=[-u_max] * nu, upper=[u_max] * nu))
        forms.append("ubound")
    return dict(nx=nx, nu=nu, N=N, A=A, B=B, d=d, x0=x0, costs=costs, cstrs=cstrs, forms=forms)


def make_chain3(seed, batch):
    """Random controllers on chains of THREE states per control (position, velocity, acceleration per axis, the jerk as control) in two and three
    dimensions -- the shapes of the (instance, axis)-per-lane solver's late-round-6 builds: random horizon up to 20, per-instance sampling
    period, a goal or a reference trajectory, optional target cost, random mixes of velocity / acceleration bounds, jerk bounds, rows and
    mixed rows (everything per axis: the solver takes the controller; a dense state row now and then: it does not)."""
    rng = np.random.default_rng([seed, 993])
    dim = int(rng.choice([2, 3], p=[0.4, 0.6]))
    nx, nu = 3 * dim, dim
    N = int(rng.integers(4, 21))
    forms = []
    T = rng.uniform(0.08, 0.15, batch)
    I = np.eye(dim)
    A = np.zeros((batch, nx, nx))
    B = np.zeros((batch, nx, nu))
    for a in range(3):
        A[:, a * dim:(a + 1) * dim, a * dim:(a + 1) * dim] = I
    A[:, :dim, dim:2 * dim] = T[:, None, None] * I
    A[:, :dim, 2 * dim:] = (0.5 * T * T)[:, None, None] * I
    A[:, dim:2 * dim, 2 * dim:] = T[:, None, None] * I
    B[:, :dim, :] = (T ** 3 / 6.0)[:, None, None] * I
    B[:, dim:2 * dim, :] = (0.5 * T * T)[:, None, None] * I
    B[:, 2 * dim:, :] = T[:, None, None] * I
    d = np.zeros((batch, nx)) if rng.random() < 0.6 else np.tile(0.005 * rng.standard_normal(nx), (batch, 1))
    v_max = float(rng.uniform(0.25, 0.6))
    a_max = float(rng.uniform(1.0, 3.0))
    j_max = float(rng.uniform(5.0, 20.0))
    x0 = np.zeros((batch, nx))
    x0[:, :dim] = rng.standard_normal((batch, dim)) * 0.3
    x0[:, dim:2 * dim] = rng.uniform(-0.7 * v_max, 0.7 * v_max, (batch, dim))
    x0[:, 2 * dim:] = rng.uniform(-0.5 * a_max, 0.5 * a_max, (batch, dim))
    goal = np.concatenate([rng.uniform(-1.0, 1.0, dim), np.zeros(2 * dim)])
    wx = np.concatenate([rng.uniform(5.0, 20.0, dim), rng.uniform(0.5, 2.0, dim), rng.uniform(0.05, 0.2, dim)])
    costs = []
    if rng.random() < 0.35:
        pk = goal[None, :] * np.linspace(0.5, 1.0, N + 1)[:, None] + 0.01 * rng.standard_normal((N + 1, nx))
        costs.append(dict(kind="trajectory", M=_blockdiag(np.eye(nx), N + 1), p=pk.reshape(-1), weights=np.tile(wx, N + 1)))
        forms.append("xref")
    else:
        costs.append(dict(kind="trajectory", M=np.eye(nx), p=goal, weights=wx))
        forms.append("xcost")
    costs.append(dict(kind="control", N=np.eye(nu), p=np.zeros(nu), weights=[float(rng.uniform(1e-4, 1e-2))] * nu))
    if rng.random() < 0.3:
        costs.append(dict(kind="target", M=np.eye(nx), p=goal, weights=5.0 * wx))
        forms.append("target")
    cstrs = []
    inf = np.inf
    two_rows = rng.random() < 0.35
    if rng.random() < 0.85:
        cstrs.append(dict(kind="trajectory_bound", lower=[-inf] * nx, upper=[inf] * dim + [v_max] * dim + ([a_max] * dim if two_rows else [inf] * dim)))
        forms.append("v,a bound" if two_rows else "vbound")
    if rng.random() < 0.85:
        cstrs.append(dict(kind="control_bound", lower=[-j_max] * nu, upper=[j_max] * nu))
        forms.append("jbound")
    vsel = np.hstack([np.zeros((dim, dim)), np.eye(dim), np.zeros((dim, dim))])
    if not two_rows and rng.random() < 0.3:
        cstrs.append(dict(kind="trajectory", E=-vsel, f=[v_max] * dim, ineq=True))  # the lower velocity limit as rows
        forms.append("-v rows")
    if not two_rows and rng.random() < 0.2:
        cstrs.append(dict(kind="mixed", E=vsel, G=0.01 * np.eye(dim), f=[v_max * 1.1] * dim, ineq=True))
        forms.append("v+Tj")
    if rng.random() < 0.1:
        E = rng.standard_normal((1, nx))
        cstrs.append(dict(kind="trajectory", E=E, f=[float(np.abs(E @ x0.T).max() + rng.uniform(0.5, 1.5))], ineq=True))
        forms.append("dense-x")
    if not cstrs:
        cstrs.append(dict(kind="control_bound", lower=[-j_max] * nu, upper=[j_max] * nu))
        forms.append("jbound")
    return dict(nx=nx, nu=nu, N=N, A=A, B=B, d=d, x0=x0, costs=costs, cstrs=cstrs, forms=forms)


def make_integrator(seed, batch):
    """Random controllers on the shapes whose Riccati-factor tier and one-instance-per-lane pass the library ships for every horizon
    (double integrators in one, two and three dimensions, per-instance sampling period): random horizon, random per-step costs (incl.
    reference trajectories as block-diagonal full-size entries), random mixes of bound, row and mixed constraints.  At a batch of a few
    thousand and more these run the headline's pair of kernels (pass + tier, hand-over or filter), with or without general rows."""
    rng = np.random.default_rng([seed, 77])
    dim = int(rng.choice([1, 2, 3], p=[0.25, 0.3, 0.45]))
    nx, nu = 2 * dim, dim
    N = int(rng.integers(4, min(24, 64 // nu) + 1))
    X, U = nx * (N + 1), nu * N
    forms = []
    T = rng.uniform(0.08, 0.15, batch)
    I = np.eye(dim)
    A = np.zeros((batch, nx, nx))
    B = np.zeros((batch, nx, nu))
    A[:, :dim, :dim] = I
    A[:, dim:, dim:] = I
    A[:, :dim, dim:] = T[:, None, None] * I
    B[:, :dim, :] = (0.5 * T * T)[:, None, None] * I
    B[:, dim:, :] = T[:, None, None] * I
    d = np.zeros((batch, nx)) if rng.random() < 0.5 else np.tile(0.01 * rng.standard_normal(nx), (batch, 1))
    v_max = float(rng.uniform(0.25, 0.7))
    u_max = float(rng.uniform(1.0, 3.0))
    x0 = np.zeros((batch, nx))
    x0[:, :dim] = rng.standard_normal((batch, dim)) * 0.3
    x0[:, dim:] = rng.uniform(-0.8 * v_max, 0.8 * v_max, (batch, dim))
    goal = np.concatenate([rng.uniform(-1.0, 1.0, dim), rng.uniform(-0.1, 0.1, dim)])
    wx = np.concatenate([rng.uniform(5.0, 20.0, dim), rng.uniform(0.5, 2.0, dim)])
    costs = []
    if rng.random() < 0.35:  # a reference TRAJECTORY (block-diagonal full-size entry)
        pk = goal[None, :] * np.linspace(0.5, 1.0, N + 1)[:, None] + 0.01 * rng.standard_normal((N + 1, nx))
        costs.append(dict(kind="trajectory", M=_blockdiag(np.eye(nx), N + 1), p=pk.reshape(-1), weights=np.tile(wx, N + 1)))
        forms.append("xref")
    else:
        plain = rng.random() < 0.7
        M = np.eye(nx) if plain else np.eye(nx) + 0.2 * rng.standard_normal((nx, nx))
        costs.append(dict(kind="trajectory", M=M, p=goal, weights=wx))
        forms.append("xcost" if plain else "xcost-M")
    costs.append(dict(kind="control", N=np.eye(nu), p=np.zeros(nu), weights=[float(rng.uniform(1e-3, 1e-1))] * nu))
    if rng.random() < 0.3:
        costs.append(dict(kind="target", M=np.eye(nx), p=goal, weights=5.0 * wx))
        forms.append("target")
    if rng.random() < 0.2:
        costs.append(dict(kind="mixed", M=0.3 * rng.standard_normal((1, nx)), N=0.3 * rng.standard_normal((1, nu)), p=[0.0], weights=[1.0]))
        forms.append("mixed")
    cstrs = []
    inf = np.inf
    if rng.random() < 0.8:
        cstrs.append(dict(kind="trajectory_bound", lower=[-inf] * nx, upper=[inf] * dim + [v_max] * dim))
        forms.append("vbound")
    if rng.random() < 0.8:
        cstrs.append(dict(kind="control_bound", lower=[-u_max] * nu, upper=[u_max] * nu))
        forms.append("ubound")
    vsel = np.hstack([np.zeros((dim, dim)), np.eye(dim)])
    if rng.random() < 0.3:
        cstrs.append(dict(kind="trajectory", E=np.vstack([-vsel]), f=[v_max] * dim, ineq=True))  # the lower velocity limit as rows
        forms.append("-v rows")
    if rng.random() < 0.25:
        E = rng.standard_normal((1, nx))
        cstrs.append(dict(kind="trajectory", E=E, f=[float(np.abs(E @ x0.T).max() + rng.uniform(0.2, 1.0))], ineq=True))
        forms.append("dense-x")
    if rng.random() < 0.25:
        cstrs.append(dict(kind="mixed", E=np.hstack([np.zeros((dim, dim)), np.eye(dim)]), G=0.1 * np.eye(dim), f=[v_max * 1.1] * dim, ineq=True))
        forms.append("v+Tu")
    if rng.random() < 0.25:  # terminal velocity box as full-size rows
        E = np.zeros((2 * dim, X))
        E[:dim, X - dim:] = np.eye(dim)
        E[dim:, X - dim:] = -np.eye(dim)
        cstrs.append(dict(kind="trajectory", E=E, f=[0.5 * v_max] * (2 * dim), ineq=True))
        forms.append("terminal-full")
    if not cstrs:
        cstrs.append(dict(kind="control_bound", lower=[-u_max] * nu, upper=[u_max] * nu))
        forms.append("ubound")
    return dict(nx=nx, nu=nu, N=N, A=A, B=B, d=d, x0=x0, costs=costs, cstrs=cstrs, forms=forms)


def com_preview_with_general_rows(seed, batch):
    """The CoM preview workload (6, 3, 20) -- the shape whose Riccati-factor tier has a shared-model mode -- with its bounds and one more
    constraint of general rows: seed % 3 == 0 two DENSE state rows at every step (they go through the free response of the preview),
    1 a mixed state/control row, 2 a control row.  -> (workload, constraints)"""
    from copra_amd import workloads
    rng = np.random.default_rng(seed)
    wl = workloads.com_preview(batch, seed=seed, v_max=float(rng.uniform(0.25, 0.6)), u_max=float(rng.uniform(1.2, 3.0)))
    cstrs = list(wl["cstrs"])
    kind = seed % 3
    if kind == 0:
        cstrs.append(dict(kind="trajectory", E=rng.standard_normal((2, 6)) * 0.5, f=np.abs(rng.standard_normal(2)) * 0.3 + 0.2))
    elif kind == 1:
        cstrs.append(dict(kind="mixed", E=rng.standard_normal((1, 6)) * 0.5, G=rng.standard_normal((1, 3)) * 0.3, f=[0.8]))
    else:
        cstrs.append(dict(kind="control", G=rng.standard_normal((1, 3)), f=[1.0]))
    return wl, cstrs
