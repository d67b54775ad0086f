"""The reference's Python surface (binding/python/CopraBindings.cpp, `import pyCopra as copra`) over the HIP engine:
binding/python/tests/pyTests.py restated against copra_amd.pycopra with the reference's own setUp values
(pyTests.py:12-56: falling mass, nbStep = 300 -> the workgroup-per-instance kernel) and acceptance checks."""
import numpy as np
import pytest

import fixtures as F


@pytest.fixture()
def S():
    class Setup:
        timestep, mass, nbStep = 0.005, 5, 300
        A = np.array([[1.0, timestep], [0.0, 1.0]])
        B = np.array([[0.5 * timestep * timestep / mass], [timestep / mass]])
        c = np.array([(-9.81 / 2.0) * timestep ** 2, -9.81 * timestep])
        x0 = np.array([0.0, -5.0])
        wu, wx = np.array([1e-4]), np.array([10.0, 10000.0])
        xd, ud = np.zeros(2), np.zeros(1)
        M, N = np.identity(2), np.ones((1, 1))
        Gineq, hineq = np.ones((1, 1)), np.array([200.0])
        Eineq, fineq = np.array([[0.0, 1.0]]), np.zeros(1)
        uLower, uUpper = np.array([-np.inf]), np.array([200.0])
        xLower, xUpper = np.array([-np.inf, -np.inf]), np.array([np.inf, 0.0])
        x0Eq, xdEq = np.zeros(2), np.zeros(2)
        Geq, heq = np.ones((1, 1)), np.array([200.0])
        Eeq, feq = np.array([[1.0, 0.0], [0.0, 0.0]]), np.zeros(2)
    return Setup


def _controller(copra, S, x0=None):
    ps = copra.PreviewSystem()
    ps.system(S.A, S.B, S.c, S.x0 if x0 is None else x0, S.nbStep)
    controller = copra.LMPC(ps)
    xCost = copra.TargetCost(S.M, -S.xd)
    uCost = copra.ControlCost(S.N, -S.ud)
    xCost.weights(S.wx)
    uCost.weights(S.wu)
    controller.add_cost(xCost)
    controller.add_cost(uCost)
    return ps, controller, (xCost, uCost)


def _split(traj):
    return traj[0::2], traj[1::2]


@pytest.mark.gpu
def test_lmpc_ineq(S):  # pyTests.py:58-92
    import copra_amd.pycopra as copra
    ps, controller, keep = _controller(copra, S)
    trajConstr = copra.TrajectoryConstraint(S.Eineq, S.fineq)
    contConstr = copra.ControlConstraint(S.Gineq, S.hineq)
    controller.add_constraint(trajConstr)
    controller.add_constraint(contConstr)
    assert controller.solve()
    pos, vel = _split(controller.trajectory())
    assert abs(S.xd[1] - vel[-1]) < 5e-4  # assertAlmostEqual(places=3)
    assert pos.max() <= S.x0[0] + 1e-9
    assert controller.control().max() <= S.hineq[0] + 1e-6
    assert controller.solve_time() > 0 and controller.solve_and_build_time() >= controller.solve_time()


@pytest.mark.gpu
def test_lmpc_mixed(S):  # pyTests.py:94-132
    import copra_amd.pycopra as copra
    ps, controller, keep = _controller(copra, S)
    mixedConstr = copra.MixedConstraint(S.Eineq, S.Gineq, S.hineq)
    controller.add_constraint(mixedConstr)
    assert controller.solve()
    control, traj = controller.control(), controller.trajectory()
    pos, vel = _split(traj)
    assert abs(S.xd[1] - vel[-1]) < 5e-4 and pos.max() <= S.x0[0] + 1e-9
    assert (vel[:-1] + control <= S.hineq[0] + 1e-6).all()


@pytest.mark.gpu
def test_lmpc_bound(S):  # pyTests.py:134-169
    import copra_amd.pycopra as copra
    ps, controller, keep = _controller(copra, S)
    trajConstr = copra.TrajectoryBoundConstraint(S.xLower, S.xUpper)
    contConstr = copra.ControlBoundConstraint(S.uLower, S.uUpper)
    controller.add_constraint(trajConstr)
    controller.add_constraint(contConstr)
    assert controller.solve()
    pos, vel = _split(controller.trajectory())
    assert abs(S.xd[1] - vel[-1]) < 5e-4 and pos.max() <= S.x0[0] + 1e-9
    assert vel.max() <= S.xUpper[1] + 1e-6 and controller.control().max() <= S.uUpper[0] + 1e-6


@pytest.mark.gpu
def test_default_solver_flag_is_quadprog_dense(S):
    """src/solverUtils.cpp:9-34: SolverFlag::DEFAULT -> QuadProgDense, at every size.  (Round-4 verdict: above 64 variables the mirrors'
    DEFAULT silently ran the interior-point kernel, SI_iter counted Newton steps.)  That kernel is an explicit opt-in: SolverFlag.HipRiccati."""
    import copra_amd.pycopra as copra
    ps, controller, keep = _controller(copra, S)  # 300 decision variables, stage-wise: the engine's own choice would be the interior-point kernel
    cb = copra.ControlBoundConstraint(S.uLower, S.uUpper)
    controller.add_constraint(cb)
    assert controller.solver_kind() == "quadprog_dense"
    assert controller.solve()
    u_gi = controller.control().copy()
    fast = copra.LMPC(ps, copra.SolverFlag.HipRiccati)
    xCost, uCost = copra.TargetCost(S.M, -S.xd), copra.ControlCost(S.N, -S.ud)
    xCost.weights(S.wx)
    uCost.weights(S.wu)
    fast.add_cost(xCost)
    fast.add_cost(uCost)
    fast.add_constraint(cb)
    assert fast.solver_kind() == "riccati_ipm"
    assert fast.solve()
    assert np.abs(fast.control() - u_gi).max() <= 1e-6 * max(1.0, np.abs(u_gi).max())
    fast.select_qp_solver(copra.SolverFlag.DEFAULT)
    assert fast.solver_kind() == "quadprog_dense"
    explicit = copra.LMPC(ps, copra.SolverFlag.QuadProgDense)
    assert copra.SolverFlag.engine_solver(copra.SolverFlag.DEFAULT) == copra.SolverFlag.engine_solver(copra.SolverFlag.QuadProgDense) == "quadprog_dense"
    del explicit


@pytest.mark.gpu
def test_lmpc_eq(S):  # pyTests.py:171-203
    import copra_amd.pycopra as copra
    ps, controller, keep = _controller(copra, S, x0=S.x0Eq)
    trajConstr = copra.TrajectoryConstraint(S.Eeq, S.feq, False)
    controller.add_constraint(trajConstr)
    assert controller.solve()
    pos, vel = _split(controller.trajectory())
    assert abs(S.xdEq[1] - vel[-1]) < 5e-4
    assert np.abs(pos).max() <= 1e-6 and np.abs(vel).max() <= 1e-6


@pytest.mark.gpu
def test_constraint_and_cost_deletion(S):  # pyTests.py:233-277: dropped pieces leave the controller after a solve
    import copra_amd.pycopra as copra
    ps = copra.PreviewSystem()
    ps.system(S.A, S.B, S.c, S.x0, S.nbStep)
    controller = copra.LMPC(ps)
    trajConstr = copra.TrajectoryConstraint(S.Eineq, S.fineq)
    contConstr = copra.ControlConstraint(S.Gineq, S.hineq)
    trajEqConstr = copra.TrajectoryConstraint(S.Eeq, S.feq, False)
    contEqConstr = copra.ControlConstraint(S.Geq, S.heq, False)
    trajBdConstr = copra.TrajectoryBoundConstraint(S.xLower, S.xUpper)
    contBdConstr = copra.ControlBoundConstraint(S.uLower, S.uUpper)
    targetCost = copra.TargetCost(S.M, -S.xd)
    trajectoryCost = copra.TrajectoryCost(S.M, -S.xd)
    controlCost = copra.ControlCost(S.N, -S.ud)
    mixedCost = copra.MixedCost(np.ones((1, 2)), S.N, -S.ud)
    for c in (trajConstr, contConstr, trajEqConstr, contEqConstr, trajBdConstr, contBdConstr):
        controller.add_constraint(c)
    for c in (targetCost, trajectoryCost, controlCost, mixedCost):
        controller.add_cost(c)
    del c
    del trajConstr
    targetCost.weights(S.wx)
    controlCost.weights(S.wu)
    del trajEqConstr, contEqConstr, trajBdConstr, contBdConstr, trajectoryCost, mixedCost
    assert not controller.solve()  # contradictory equalities are still in
    assert controller.solve()  # "Has kept the contConstr only"


@pytest.mark.gpu
def test_preview_system_still_exists_and_receding_horizon(S):  # pyTests.py:279-309 + PreviewSystem::xInit
    import copra_amd.pycopra as copra
    ps, controller, keep = _controller(copra, S)
    trajConstr = copra.TrajectoryConstraint(S.Eineq, S.fineq)
    contConstr = copra.ControlConstraint(S.Gineq, S.hineq)
    controller.add_constraint(trajConstr)
    controller.add_constraint(contConstr)
    del ps
    assert controller.solve()
    traj = controller.trajectory()
    pos, vel = _split(traj)
    assert abs(S.xd[1] - vel[-1]) < 5e-4 and pos.max() <= S.x0[0] + 1e-9
    controller._ps.x_init(traj[2:4])
    assert controller.solve() and np.abs(controller.trajectory()[:2] - traj[2:4]).max() < 1e-12


def test_constructors_and_throw_handler(S):  # pyTests.py:205-216, 311-339 (no GPU needed)
    import copra_amd.pycopra as copra
    ps = copra.PreviewSystem()
    ps.system(S.A, S.B, S.c, S.x0, S.nbStep)
    controller = copra.LMPC(ps)
    copra.LMPC()
    copra.LMPC(copra.SolverFlag.QuadProgDense)
    copra.LMPC(ps, copra.SolverFlag.QuadProgDense)
    controller.initialize_controller(ps)
    for bad in (lambda: copra.TrajectoryConstraint(np.identity(5), np.ones(2)),
                lambda: copra.ControlConstraint(np.identity(5), np.ones(2)),
                lambda: copra.MixedConstraint(np.identity(5), np.identity(5), np.ones(2)),
                lambda: copra.TrajectoryBoundConstraint(np.ones(3), np.ones(2)),
                lambda: copra.ControlBoundConstraint(np.ones(3), np.ones(2))):
        with pytest.raises(RuntimeError):
            controller.add_constraint(bad())
    with pytest.raises(RuntimeError):
        copra.PreviewSystem().system(np.ones((5, 2)), S.B, S.c, S.x0, S.nbStep)
    with pytest.raises(TypeError):
        copra.TrajectoryConstraint()  # pyTests.py:218-231: no default constructors
    assert copra.AutoSpan.span_matrix(np.ones((1, 2)), 3).shape == (3, 6)
    assert copra.AutoSpan.span_vector(np.ones(2), 6).shape == (6,)
    with pytest.raises(RuntimeError):
        copra.AutoSpan.span_vector(np.ones(4), 6)


@pytest.mark.gpu
def test_dynamic_walk():  # pyTests.py:341-443: CoM system, full-size 66 x 30 ControlConstraint, TargetCost
    import copra_amd.pycopra as copra
    pb = F.com_walk_problem()
    ps = copra.PreviewSystem()
    ps.system(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"])
    controller = copra.LMPC(ps)
    c0 = pb["cstrs"][0]
    contConstr = copra.ControlConstraint(c0["G"], c0["f"])
    k0 = pb["costs"][0]
    targetCost = copra.TargetCost(k0["M"], k0["p"])
    controller.add_constraint(contConstr)
    controller.add_cost(targetCost)
    assert controller.solve()
    assert (np.asarray(c0["G"]) @ controller.control() <= np.asarray(c0["f"]) + 1e-6).all()
    assert controller.solve_time() > 0


@pytest.mark.gpu
def test_initial_state_lmpc(S):  # include/InitialStateLMPC.h through the Python surface
    import copra_amd.pycopra as copra
    ps = copra.PreviewSystem()
    ps.system(S.A, S.B, S.c, S.x0, 40)
    controller = copra.InitialStateLMPC(ps)
    controller.reset_initial_state_cost(10.0 * np.eye(2), np.zeros(2))
    controller.reset_initial_state_bounds(S.x0 - 0.05, S.x0 + 0.05)
    xCost = copra.TrajectoryCost(S.M, -S.xd)
    uCost = copra.ControlCost(S.N, -S.ud)
    xCost.weights(S.wx)
    uCost.weights(S.wu)
    bound = copra.ControlBoundConstraint(S.uLower, S.uUpper)
    controller.add_cost(xCost)
    controller.add_cost(uCost)
    controller.add_constraint(bound)
    assert controller.solve()
    x0s = controller.initial_state()
    assert (x0s <= S.x0 + 0.05 + 1e-6).all() and (x0s >= S.x0 - 0.05 - 1e-6).all()
    assert np.abs(controller.trajectory()[:2] - x0s).max() < 1e-12


@pytest.mark.gpu
def test_cost_changes_between_solves_like_the_reference():
    """the reference evaluates every cost anew in every solve (LMPC.cpp:233-247): weights() on a cost that is inside the controller takes
    effect at the next solve (a new plan), and a cost REPLACED by one that differs in p alone -- the reference's only way to move a
    reference trajectory -- is sent to the engine that exists (copra_batch_set_cost_reference) instead of building a new one; every state
    against a controller built from scratch"""
    import copra_amd.pycopra as copra
    from copra_amd import workloads
    wl = workloads.com_preview(1)
    N, A, B, d = wl["N"], wl["A"][0], wl["B"][0], wl["d"][0]
    x0, goal = workloads.COM_X_INIT, workloads.COM_X_GOAL
    wx, inf = np.array([10.0, 10, 10, 1, 1, 1]), np.inf

    def reference(tick):
        s = np.minimum(1.0, (np.arange(N + 1) + 0.3 * tick) / N)
        return (x0[None, :] + s[:, None] * (goal - x0)[None, :]).reshape(-1)

    def build(p, w, x):
        ps = copra.PreviewSystem(A, B, d, x, N)
        c = copra.LMPC(ps)
        xc = copra.TrajectoryCost(np.eye(6 * (N + 1)), p)
        xc.weights(w)
        uc = copra.ControlCost(np.eye(3), np.zeros(3))
        uc.weights(np.full(3, 1e-3))
        keep = (xc, uc, copra.TrajectoryBoundConstraint(np.full(6, -inf), np.array([inf, inf, inf, 0.6, 0.6, 0.6])),
                copra.ControlBoundConstraint(np.full(3, -3.0), np.full(3, 3.0)))
        c.add_cost(xc), c.add_cost(uc), c.add_constraint(keep[2]), c.add_constraint(keep[3])
        return ps, c, keep

    ps, ctl, keep = build(reference(0), wx, x0)
    xc = keep[0]
    assert ctl.solve()
    builds = None
    for tick in range(1, 8):
        x = x0 + 0.002 * tick
        ps.x_init(x)
        nxt = copra.TrajectoryCost(np.eye(6 * (N + 1)), reference(tick))
        nxt.weights(wx)
        ctl.remove_cost(xc)
        ctl.add_cost(nxt)
        xc = nxt
        assert ctl.solve()
        if tick == 1:
            builds = ctl.handle_builds  # (remove + add moved the cost to the end of the list: a new order once, the same from then on)
        _, fresh, keep2 = build(reference(tick), wx, x)
        assert fresh.solve() and np.abs(fresh.control() - ctl.control()).max() <= 1e-10
    assert ctl.handle_builds == builds
    w2 = wx.copy()
    w2[0], w2[4] = 3.0, 2.5
    xc.weights(w2)
    assert ctl.solve() and ctl.handle_builds == builds + 1
    _, fresh, keep2 = build(reference(7), w2, x)
    assert fresh.solve() and np.abs(fresh.control() - ctl.control()).max() <= 1e-10
    _, stale, keep3 = build(reference(7), wx, x)
    assert stale.solve() and np.abs(stale.control() - ctl.control()).max() > 1e-6  # (the weights did something)


@pytest.mark.gpu
def test_reference_accumulation_switch():
    """reference quirk Q2 (src/costFunctions.cpp:73-80, 205-213) as an opt-in of the Python mirror (LMPC.reference_accumulation): the k-th
    solve of one controller with a per-step TrajectoryCost equals a fresh controller with k x its weights; a per-step MixedCost also
    accumulates c -- checked against the dense QP  (1e-6 I + k Q1 + Qu) U + (sum_j j c1(x0_j) + cu)  assembled from the oracle's evaluation
    of ONE update and solved by the oracle's QuadProgDense restatement.  Switched off (the default), every solve is a fresh controller's."""
    import copra_amd.pycopra as copra
    import pyoracle
    from copra_amd import workloads
    wl = workloads.com_preview(1, v_max=0.4, u_max=2.0)
    N, A, B, d = wl["N"], wl["A"][0], wl["B"][0], wl["d"][0]
    x0, goal = workloads.COM_X_INIT, workloads.COM_X_GOAL
    wx = np.array([10.0, 10, 10, 1, 1, 1])
    inf = np.inf

    def build(scale, x, accumulate=False):
        ps = copra.PreviewSystem(A, B, d, x, N)
        c = copra.LMPC(ps)
        c.reference_accumulation(accumulate)
        xc = copra.TrajectoryCost(np.eye(6), goal)
        xc.weights(wx * scale)
        uc = copra.ControlCost(np.eye(3), np.zeros(3))
        uc.weights(np.full(3, 1e-3))
        keep = (xc, uc, copra.ControlBoundConstraint(np.full(3, -2.0), np.full(3, 2.0)),
                copra.TrajectoryBoundConstraint(np.full(6, -inf), np.array([inf, inf, inf, 0.4, 0.4, 0.4])))
        c.add_cost(xc), c.add_cost(uc), c.add_constraint(keep[2]), c.add_constraint(keep[3])
        return ps, c, keep

    for accumulate in (False, True):
        ps, ctl, keep = build(1.0, x0, accumulate)
        for k in (1, 2, 3):
            x = x0 + 0.01 * (k - 1)
            ps.x_init(x)
            assert ctl.solve()
            _, fresh, keep2 = build(float(k) if accumulate else 1.0, x)
            assert fresh.solve() and np.abs(fresh.control() - ctl.control()).max() <= 1e-8
        if accumulate:
            _, plain, keep3 = build(1.0, x)
            assert plain.solve() and np.abs(plain.control() - ctl.control()).max() > 1e-4  # (the accumulation did something)

    # MixedCost: Q_k = k Q1, c_k = c_{k-1} + k (E1' x0_k + f1)
    Mm, Nm, pm, wm = np.hstack([np.zeros((3, 3)), np.eye(3)]), 0.05 * np.eye(3), np.full(3, 0.2), np.full(3, 40.0)
    mixed = dict(kind="mixed", M=Mm, N=Nm, p=pm, weights=wm)
    ucost = dict(kind="control", N=np.eye(3), p=np.zeros(3), weights=np.full(3, 1e-3))
    cstrs = [dict(kind="control_bound", lower=[-2.0] * 3, upper=[2.0] * 3)]
    ps = copra.PreviewSystem(A, B, d, x0, N)
    ctl = copra.LMPC(ps)
    ctl.reference_accumulation(True)
    mc = copra.MixedCost(Mm, Nm, pm)
    mc.weights(wm)
    uc = copra.ControlCost(np.eye(3), np.zeros(3))
    uc.weights(np.full(3, 1e-3))
    ub = copra.ControlBoundConstraint(np.full(3, -2.0), np.full(3, 2.0))
    ctl.add_cost(mc), ctl.add_cost(uc), ctl.add_constraint(ub)
    n = 3 * N
    cacc = np.zeros(n)
    for k in (1, 2, 3):
        x = x0 + 0.01 * (k - 1)
        ps.x_init(x)
        assert ctl.solve()
        q1 = pyoracle.lmpc_build(A, B, d, x, N, [mixed], [])  # Q = 1e-6 I + Q1, c = c1(x)
        qu = pyoracle.lmpc_build(A, B, d, x, N, [ucost], cstrs)
        cacc += k * q1["c"]
        Q = k * (q1["Q"] - 1e-6 * np.eye(n)) + qu["Q"]
        want, fail, _ = pyoracle.quadprog_dense(Q, cacc + qu["c"], None, None, None, None, qu["lb"], qu["ub"])
        assert fail == 0 and np.abs(want - ctl.control()).max() <= 1e-7


@pytest.mark.gpu
def test_random_controllers_through_the_python_surface():
    """tests/random_controllers.py through copra_amd.pycopra (the reference's Python classes over one-problem controllers): 80 random
    mixes of the four cost and five constraint classes, per-step and full-size -- solve() returns what the oracle's status says, control()
    and trajectory() within 1e-6 entry-wise (floor 1e-3); the same LMPC object then solves a second initial state (xInit, the receding-
    horizon idiom of the reference's tests)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyoracle as oracle
    import random_controllers as RC
    from copra_amd import pycopra as P

    def rel(a, b):
        return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3)))

    nsolved = 0
    for seed in range(80):
        c = RC.make(seed, batch=2)
        N = c["N"]
        ps = P.PreviewSystem(c["A"][0], c["B"][0], c["d"][0], c["x0"][0], N)
        lmpc = P.LMPC(ps)
        keep = []
        for d in c["costs"]:
            kind = d["kind"]
            obj = (P.TrajectoryCost(d["M"], d["p"]) if kind == "trajectory" else P.TargetCost(d["M"], d["p"]) if kind == "target"
                   else P.ControlCost(d["N"], d["p"]) if kind == "control" else P.MixedCost(d["M"], d["N"], d["p"]))
            obj.weights(np.asarray(d["weights"], dtype=float))
            lmpc.add_cost(obj)
            keep.append(obj)
        for d in c["cstrs"]:
            kind = d["kind"]
            if kind == "trajectory":
                obj = P.TrajectoryConstraint(d["E"], d["f"], d.get("ineq", True))
            elif kind == "control":
                obj = P.ControlConstraint(d["G"], d["f"], d.get("ineq", True))
            elif kind == "mixed":
                obj = P.MixedConstraint(d["E"], d["G"], d["f"], d.get("ineq", True))
            elif kind == "trajectory_bound":
                obj = P.TrajectoryBoundConstraint(d["lower"], d["upper"])
            else:
                obj = P.ControlBoundConstraint(d["lower"], d["upper"])
            lmpc.add_constraint(obj)
            keep.append(obj)
        for k in range(2):
            if k == 1:
                ps.x_init(c["x0"][1])
            x0 = c["x0"][k]
            ref = oracle.lmpc_solve(c["A"][0], c["B"][0], c["d"][0], x0, N, c["costs"], c["cstrs"])
            ok = lmpc.solve()
            what = "seed %d (%d, %d, %d) %s solve %d" % (seed, c["nx"], c["nu"], N, c["forms"], k)
            assert bool(ok) == (ref["status"] == 0), what
            if ok:
                assert rel(np.asarray(lmpc.control()), ref["control"]) <= 1e-6 and rel(np.asarray(lmpc.trajectory()), ref["trajectory"]) <= 1e-6, what
                nsolved += 1
    assert nsolved >= 120
