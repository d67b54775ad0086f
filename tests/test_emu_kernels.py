"""The HIP kernel BODIES (copra_amd/csrc/*.hpp) executed lane-by-lane on the CPU by the fiber emulator under
tests/emu/ and compared with the oracle.  This is how the kernel logic is covered without a GPU; the same bodies,
compiled by hipcc, are what `-m gpu` tests run on the MI355X.  (The emulator is test infrastructure: see
tests/emu/wave_prims.hpp.)"""
import os

import numpy as np
import pytest

from copra_amd._capi import OPTIONS  # engine options (copra_options_t): tests pin a tier by switching the others off

import fixtures as F

RTOL = 1e-6  # BASELINE.json north_star tolerance


@pytest.fixture(scope="module")
def emu():
    import pyemu
    pyemu.lib()
    return pyemu


def _rel(a, b):
    return np.nanmax(np.abs(a - b) / (1.0 + np.abs(b)))


def _compare(emu, oracle, A, B, d, x0, N, costs, cstrs, specialised=True, same_iters=True):
    re = emu.lmpc_solve(A, B, d, x0, N, costs, cstrs, specialised=specialised)
    ro = oracle.lmpc_solve_batch(np.atleast_3d(A) if np.ndim(A) == 3 else A[None], B if np.ndim(B) == 3 else B[None],
                                 d if np.ndim(d) == 2 else d[None], x0 if np.ndim(x0) == 2 else x0[None], N, costs,
                                 cstrs)
    assert (re["status"] == ro["status"]).all()
    ok = ro["status"] == 0
    if ok.any():
        assert _rel(re["control"][ok], ro["control"][ok]) <= RTOL
        assert _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= RTOL
    if same_iters:
        assert (re["iter"] == ro["iter"]).all()
    return re, ro



@pytest.fixture
def no_axis(monkeypatch):
    """the tests of the one-instance-per-lane pass (lmpc_lane.hpp) and of what it hands to the tier: without the one-(instance, axis)-per-lane
    solver (lmpc_axis.hpp, round 6), which takes the same controllers first"""
    monkeypatch.setitem(OPTIONS, "no_axis_solver", 1)


@pytest.mark.parametrize("specialised", [True, False])
def test_config2_double_integrator(emu, oracle, specialised):
    from copra_amd import workloads
    wl = workloads.double_integrator(12)
    _compare(emu, oracle, wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"], specialised)


@pytest.mark.parametrize("specialised", [True, False])
@pytest.mark.parametrize("vmax,umax", [(0.6, 3.0), (0.25, 1.2)])
def test_config3_com_preview(emu, oracle, specialised, vmax, umax):
    from copra_amd import workloads
    wl = workloads.com_preview(10, v_max=vmax, u_max=umax, seed=7)
    _compare(emu, oracle, wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"], specialised)


@pytest.mark.parametrize("case", ["bounded", "ineq", "mixed", "eq", "nine", "com15", "di48"])
def test_factor_only_layout(emu, oracle, case):
    """More than 32 decision variables: the first tier keeps only the packed Cholesky factor in LDS and runs the active
    set on R^-T N = Q1 Rq (gi_core.hpp, TRI); same statuses, iterates and iteration counts as the oracle, with
    equality rows, every cost / constraint class, and instances that overflow into the square-layout second tier"""
    import fixtures as F
    from copra_amd import workloads
    if case == "nine":
        pb = F.nine_class_problem(44)
    elif case == "com15":
        wl = workloads.com_preview(12, N=15, v_max=0.2, u_max=1.0, seed=3)
        pb = dict(wl, x0=wl["x0"])
    elif case == "di48":
        wl = workloads.double_integrator(8, N=48)
        pb = dict(wl)
    else:
        pb = getattr(F, case + "_system")("trajectory" if case != "eq" else "target", N=50)
    re, ro = _compare(emu, oracle, pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"],
                      specialised=False)
    assert re["factor_only"] and re["rcap"] < 40
    if case == "com15":
        assert re["overflowed"] > 0 and ro["iter"][:, 0].max() > re["rcap"] + 1  # the second tier finished some


@pytest.mark.parametrize("specialised", [True, False])
def test_headline_shape_with_a_general_output_map(emu, oracle, specialised):
    """the cost phase reads G instead of forming M G_k when M is the identity (CostTerm::ident); a general 6 x 6 M and a
    5-row selection (padded to the six rows of the compile-time shape) take the ordinary route"""
    from copra_amd import workloads
    wl = workloads.com_preview(6, v_max=0.3, u_max=1.5, seed=5)
    rng = np.random.default_rng(2)
    c0 = wl["costs"][0]
    Mg = np.eye(6) + 0.2 * rng.standard_normal((6, 6))
    for M, p, w in ((Mg, Mg @ c0["p"], c0["weights"]), (np.eye(6)[:5], c0["p"][:5], c0["weights"][:5])):
        costs = [dict(kind="trajectory", M=M, p=p, weights=w), wl["costs"][1]]
        _compare(emu, oracle, wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], costs, wl["cstrs"], specialised)


@pytest.mark.parametrize("order", ["target-first", "control-first", "mixed-first", "trajectory-mixed-target"])
def test_headline_shape_every_per_step_cost_kind(emu, oracle, order):
    """compile-time (6, 3, 20) instantiation: the grouped walk along the block diagonals with a TargetCost (no running
    sum), a MixedCost (cross terms, one step shorter) and a ControlCost in first position (then the Hessian is cleared
    and added to instead of stored outright); Hessian / gradient dump and solution against the oracle"""
    from copra_amd import workloads
    wl = workloads.com_preview(4, v_max=0.4, u_max=2.0, seed=11)
    rng = np.random.default_rng(5)
    traj, ctrl = wl["costs"]
    target = dict(kind="target", M=np.eye(6)[[0, 2, 4]], p=traj["p"][[0, 2, 4]], weights=[40.0, 40.0, 10.0])
    mixed = dict(kind="mixed", M=0.3 * rng.standard_normal((2, 6)), N=0.1 * rng.standard_normal((2, 3)), p=[0.1, -0.2],
                 weights=[3.0, 5.0])
    costs = {"target-first": [target, ctrl], "control-first": [ctrl, traj, target], "mixed-first": [mixed, ctrl, traj],
             "trajectory-mixed-target": [traj, mixed, target, ctrl]}[order]
    for spec in (True, False):
        re = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], costs, wl["cstrs"], dump_instance=1, specialised=spec)
        qp = oracle.lmpc_build(wl["A"][1], wl["B"][1], wl["d"][1], wl["x0"][1], wl["N"], costs, wl["cstrs"])
        assert np.abs(re["Q"] - qp["Q"]).max() <= 1e-12 * np.abs(qp["Q"]).max()
        assert np.abs(re["c"] - qp["c"]).max() <= 1e-12 * max(1.0, np.abs(qp["c"]).max())
        _compare(emu, oracle, wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], costs, wl["cstrs"], spec)


def test_riccati_factor_tier_selection_and_parity(emu, oracle):
    """lmpc_fused_ric.hpp (the factor of the condensed Hessian in Riccati form): what the plan builder picks for the headline
    shape when every cost is a per-step entry (at most kRicMaxCosts of them); statuses, iteration counts and controls
    against the oracle on the tight workload (instances that overflow the five register columns finish in the second
    tier), with per-instance references, and with an instance whose QP is infeasible (reference quirk Q5)"""
    from copra_amd import workloads
    wl = workloads.com_preview(12, v_max=0.25, u_max=1.2, seed=4)
    x0 = wl["x0"].copy()
    x0[3, 3] = 0.9  # velocity beyond the bound at step 0: infeasible
    goals = wl["costs"][0]["p"][None, :] + 0.2 * np.random.default_rng(3).standard_normal((12, 6))
    re = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], x0, wl["N"], wl["costs"], wl["cstrs"], cost_refs={0: goals})
    assert re["riccati_factor"] and re["factor_only"] and re["overflowed"] > 0
    for k in range(12):
        costs = [dict(wl["costs"][0], p=goals[k]), wl["costs"][1]]
        ro = oracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], x0[k], wl["N"], costs, wl["cstrs"])
        assert re["status"][k] == ro["status"], k
        if ro["status"] == 0:
            assert tuple(re["iter"][k]) == tuple(ro["iter"])
            assert np.abs(re["control"][k] - ro["control"]).max() <= 1e-9 * (1 + np.abs(ro["control"]).max())
            assert np.abs(re["trajectory"][k] - ro["trajectory"]).max() <= 1e-9 * (1 + np.abs(ro["trajectory"]).max())
    assert re["status"][3] == 1
    traj, ctrl = wl["costs"]
    four = [traj, ctrl, dict(kind="target", M=np.eye(6)[:2], p=traj["p"][:2], weights=[1.0, 1.0]),
            dict(kind="control", N=np.eye(3), p=np.zeros(3), weights=[1e-4] * 3)]
    assert not emu.lmpc_solve(wl["A"], wl["B"], wl["d"], x0, wl["N"], four, wl["cstrs"])["riccati_factor"]
    # an indefinite Hessian (negative weight): the condensed Q is positive definite <=> every stage's control block is, so the
    # sweep reports "Problems with the decomposition of Q" (status 2) exactly where the reference's Cholesky fails
    bad = [dict(traj, weights=[10.0, 10.0, -50.0, 1.0, 1.0, 1.0]), ctrl]
    rb = emu.lmpc_solve(wl["A"][:3], wl["B"][:3], wl["d"][:3], wl["x0"][:3], wl["N"], bad, wl["cstrs"])
    assert rb["riccati_factor"] and (rb["status"] == 2).all() and np.isnan(rb["control"]).all()
    assert oracle.lmpc_solve(wl["A"][0], wl["B"][0], wl["d"][0], wl["x0"][0], wl["N"], bad, wl["cstrs"])["status"] == 2


def test_riccati_factor_tier_without_any_cost(emu, oracle):
    """round-2 advisor finding: an LMPC with an EMPTY cost list (valid in the reference: Q = 1e-6 I, src/LMPC.cpp:228-230) is
    eligible for the Riccati-factor tier, whose branch-free prologue then evaluates the absent cost 0 -- the reference index
    must stay inside the parameter blob.  U = argmin 1e-6/2 |U|^2 under the bounds: zero where the velocity rows allow it"""
    from copra_amd import workloads
    wl = workloads.com_preview(6, v_max=0.25, u_max=1.2, seed=9)
    x0 = wl["x0"].copy()
    x0[:, 3:] = 0.3  # above the velocity bound... 
    x0[:3, 3:] = 0.2  # ... for the last three instances only (quirk Q5: those are infeasible at step 0)
    re = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], x0, wl["N"], [], wl["cstrs"])
    assert re["riccati_factor"]
    ro = oracle.lmpc_solve_batch(wl["A"], wl["B"], wl["d"], x0, wl["N"], [], wl["cstrs"])
    assert (re["status"] == ro["status"]).all() and (ro["status"][:3] == 0).all()
    ok = ro["status"] == 0
    assert _rel(re["control"][ok], ro["control"][ok]) <= RTOL and (re["iter"][ok] == ro["iter"][ok]).all()


def test_riccati_factor_tier_compact_variant_with_control_rows(emu, oracle):
    """compact variant of the Riccati-factor tier (every state term of a row is one component of one state: the blocks G are
    dead after the row norms, the normal of a state row enters w = R^-T n as a unit injection into the recursion, the
    trajectory takes G's place) together with ControlConstraint rows, whose normals DO go through the vector: statuses,
    iteration counts, U and X against the oracle"""
    from copra_amd import workloads
    wl = workloads.com_preview(8, v_max=0.3, u_max=1.5, seed=8)
    cstrs = wl["cstrs"] + [dict(kind="control", G=[[0.0, 1.0, 1.0], [1.0, -1.0, 0.0]], f=[1.2, 0.9])]
    re = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], cstrs)
    assert re["riccati_factor"] and re["lds_bytes"] < 17000  # (the general variant needs 17.7 KB + the extra rows)
    for k in range(8):
        ro = oracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], wl["costs"], cstrs)
        assert re["status"][k] == ro["status"] == 0 and tuple(re["iter"][k]) == tuple(ro["iter"]) and ro["iter"][0] >= 4
        assert _rel(re["control"][k], ro["control"]) <= RTOL and _rel(re["trajectory"][k], ro["trajectory"]) <= RTOL


def test_riccati_factor_tier_compact_variant_shared_model_more_than_64_rows(emu, oracle):
    """the compact variant in shared-model mode with 103 rows: stage records and row norms come from the model, the norms of
    rows 64.. are written over the system's slots (x0 has moved into the trajectory buffer by then); against the oracle"""
    from copra_amd import workloads
    wl = workloads.com_preview(6, v_max=0.3, u_max=1.5, seed=9)
    cstrs = wl["cstrs"] + [dict(kind="control", G=[[0.0, 1.0, 1.0], [1.0, -1.0, 0.0]], f=[1.2, 0.9])]
    A, B, d = wl["A"][0], wl["B"][0], wl["d"][0]
    re = emu.lmpc_solve_shared(A, B, d, wl["x0"], wl["N"], wl["costs"], cstrs)
    assert re["riccati_factor"]
    for k in range(6):
        ro = oracle.lmpc_solve(A, B, d, wl["x0"][k], wl["N"], wl["costs"], cstrs)
        assert re["status"][k] == ro["status"] == 0 and tuple(re["iter"][k]) == tuple(ro["iter"])
        assert _rel(re["control"][k], ro["control"]) <= RTOL and _rel(re["trajectory"][k], ro["trajectory"]) <= RTOL


def test_riccati_factor_tier_with_equality_rows(emu, oracle):
    """the Riccati-factor tier with a full-size equality entry (two rows: terminal velocities prescribed) next to the bounds:
    the equality rows go through the orientation logic of the active-set loop (eqsgn), the full-size rows keep the trajectory
    buffer of their own (seven instances per CU); statuses, iteration counts, U and X against the oracle"""
    from copra_amd import workloads
    wl = workloads.com_preview(6, v_max=0.5, u_max=2.5, seed=3)
    E = np.zeros((2, 6 * 21))
    E[0, 6 * 20 + 3] = 1.0
    E[1, 6 * 20 + 4] = 1.0
    cstrs = wl["cstrs"] + [dict(kind="trajectory", E=E, f=[0.05, -0.05], ineq=False)]
    re = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], cstrs)
    assert re["riccati_factor"]
    for k in range(6):
        ro = oracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], wl["costs"], cstrs)
        assert re["status"][k] == ro["status"] == 0 and tuple(re["iter"][k]) == tuple(ro["iter"]) and ro["iter"][0] >= 3
        assert _rel(re["control"][k], ro["control"]) <= RTOL and _rel(re["trajectory"][k], ro["trajectory"]) <= RTOL
        assert abs(re["trajectory"][k][6 * 20 + 3] - 0.05) <= 1e-9 and abs(re["trajectory"][k][6 * 20 + 4] + 0.05) <= 1e-9


@pytest.mark.parametrize("N", [10, 15])
def test_riccati_factor_tier_shorter_horizons(emu, oracle, N):
    """the other two instantiated horizons of the CoM shape (30 and 45 variables; the body is generic in the horizon): selected
    by the plan builder, statuses / iteration counts / U / X against the oracle on the default and the tight workload"""
    from copra_amd import workloads
    for kw in (dict(), dict(v_max=0.25, u_max=1.2)):
        wl = workloads.com_preview(8, N=N, seed=6, **kw)
        re = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], N, wl["costs"], wl["cstrs"])
        assert re["riccati_factor"]
        for k in range(8):
            ro = oracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], N, wl["costs"], wl["cstrs"])
            assert re["status"][k] == ro["status"]
            if ro["status"] == 0:
                assert tuple(re["iter"][k]) == tuple(ro["iter"])
                assert _rel(re["control"][k], ro["control"]) <= RTOL and _rel(re["trajectory"][k], ro["trajectory"]) <= RTOL


def test_riccati_factor_tier_with_general_rows(emu, oracle):
    """the Riccati-factor tier with rows that are NOT one component of one state: a dense TrajectoryConstraint (velocity
    sum), a MixedConstraint (state + control at the same step) and a ControlConstraint -- the row policy then has no
    maintained trajectory to read (StageRows::xu stays null: every scan refreshes the trajectory from G); statuses,
    iteration counts, U and X against the oracle"""
    from copra_amd import workloads
    wl = workloads.com_preview(6, v_max=0.4, u_max=2.0, seed=21)
    E1 = np.zeros((1, 6)); E1[0, 3:] = 1.0           # vx + vy + vz <= 0.8
    Em = np.zeros((1, 6)); Em[0, 3] = 1.0            # vx_k + 0.05 ux_k <= 0.45
    Gm = np.array([[0.05, 0.0, 0.0]])
    cstrs = [dict(kind="trajectory", E=E1, f=[0.8]),
             dict(kind="mixed", E=Em, G=Gm, f=[0.45]),
             dict(kind="control", G=[[0.0, 1.0, 1.0]], f=[2.5]),
             dict(kind="control_bound", lower=[-2.0] * 3, upper=[2.0] * 3)]
    re = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], cstrs)
    assert re["riccati_factor"]
    seen = 0
    for k in range(6):
        ro = oracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], wl["costs"], cstrs)
        assert re["status"][k] == ro["status"], k
        if ro["status"] == 0:
            assert tuple(re["iter"][k]) == tuple(ro["iter"])
            assert np.abs(re["control"][k] - ro["control"]).max() <= 1e-9 * (1 + np.abs(ro["control"]).max())
            assert np.abs(re["trajectory"][k] - ro["trajectory"]).max() <= 1e-9 * (1 + np.abs(ro["trajectory"]).max())
            seen += int(ro["iter"][0] > 1)
    assert seen >= 3  # (constraints are active in most instances)


def test_riccati_factor_tier_with_q1_in_lds(emu, oracle, monkeypatch):
    """the steps of the layout ladder below the register-Q1 one (what copra_batch_solve moves to when more than an eighth of
    the batch overflows five columns): the same body with Q1 in LDS and as many columns as five instances per CU leave --
    the tight workload then finishes in the first tier; option ric_k starts the plan there"""
    from copra_amd import workloads
    monkeypatch.setitem(OPTIONS, "ric_k", 5)
    wl = workloads.com_preview(10, v_max=0.25, u_max=1.2, seed=9)
    re = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    assert re["riccati_factor"] and re["rcap"] >= 16 and re["overflowed"] == 0
    for k in range(10):
        ro = oracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], wl["costs"], wl["cstrs"])
        assert re["status"][k] == ro["status"] and tuple(re["iter"][k]) == tuple(ro["iter"])
        assert np.abs(re["control"][k] - ro["control"]).max() <= 1e-9 * (1 + np.abs(ro["control"]).max())
    assert re["iter"][:, 0].max() > 6


def test_condensed_qp_dump_matches_oracle_build(emu, oracle):
    """Q, c, Aineq, bineq written by the device condense code == LMPC::Q() c() Aineq() bineq() of the oracle"""
    from copra_amd import workloads
    wl = workloads.com_preview(3)
    for spec in (True, False):
        re = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"], dump_instance=2,
                            specialised=spec)
        qp = oracle.lmpc_build(wl["A"][2], wl["B"][2], wl["d"][2], wl["x0"][2], wl["N"], wl["costs"], wl["cstrs"])
        assert np.abs(re["Q"] - qp["Q"]).max() <= 1e-12 * np.abs(qp["Q"]).max()
        assert np.abs(re["c"] - qp["c"]).max() <= 1e-12 * max(1.0, np.abs(qp["c"]).max())
        assert np.abs(re["Aineq"] - qp["Aineq"]).max() <= 1e-13
        assert np.abs(re["bineq"] - qp["bineq"]).max() <= 1e-12


@pytest.mark.parametrize("system", ["bounded", "ineq", "mixed", "eq"])
@pytest.mark.parametrize("xcost", ["target", "trajectory", "mixed"])
def test_reference_fixtures_short_horizon(emu, oracle, system, xcost):
    """The twelve {cost} x {constraint} combinations of tests/TestLMPC.cpp on the systems.h matrices with a horizon
    that fits the one-wave kernel (N = 12): every cost class and every constraint class goes through the generic
    instantiation of the fused kernel; the QP dump and the solution must match the oracle."""
    pb = getattr(F, system + "_system")(xcost, N=12)
    re = emu.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"], dump_instance=0)
    qp = oracle.lmpc_build(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"])
    scale = np.abs(qp["Q"]).max()
    assert np.abs(re["Q"] - qp["Q"]).max() <= 1e-12 * scale
    assert np.abs(re["c"] - qp["c"]).max() <= 1e-11 * max(1.0, np.abs(qp["c"]).max())
    for k in ("Aeq", "Aineq"):
        if qp[k].size:
            assert np.abs(re[k] - qp[k]).max() <= 1e-12 * max(1.0, np.abs(qp[k]).max())
    for k in ("beq", "bineq"):
        if qp[k].size:
            assert np.abs(re[k] - qp[k]).max() <= 1e-10 * max(1.0, np.abs(qp[k]).max())
    ro = oracle.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"])
    assert re["status"][0] == ro["status"]
    if ro["status"] == 0:
        assert _rel(re["control"][0], ro["control"]) <= RTOL
        assert _rel(re["trajectory"][0], ro["trajectory"]) <= RTOL


@pytest.mark.parametrize("full_size", [False, True])
def test_all_nine_classes_at_once(emu, oracle, full_size):
    """tests/TestLMPC_InitialState.cpp:29-130 problem: four cost classes + five constraint classes in one controller,
    with per-step entries and with the full-size entries autoSpan() produces (the latter go through the dense MFMA
    Psi' W Psi contraction)"""
    pb = F.initial_state_problem(full_size)
    _compare(emu, oracle, pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"])


def test_full_size_constraint_entries(emu, oracle, full_size_paths):
    """full-size (autoSpan'ed) constraint entries run on the fused path: E (R x fullXDim), G (R x fullUDim)"""
    from copra_amd.autospan import autospan_cstr
    pb = F.ineq_system("target", N=12)
    pb["cstrs"] = [autospan_cstr(dict(c, f=np.tile(np.atleast_1d(c["f"]), 13 if c["kind"] == "trajectory" else 12)))
                   for c in pb["cstrs"]]
    _compare(emu, oracle, pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"])
    pb = F.com_walk_problem()  # 66 x 30 full-size ControlConstraint polytope
    _compare(emu, oracle, pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"], same_iters=False)


def test_status_codes(emu, oracle):
    pb = F.bounded_system("target", N=10)
    pb["x0"] = np.array([0.0, 1.0])
    re = emu.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"])
    assert re["status"][0] == 1 and np.isnan(re["control"]).all()
    pb = F.bounded_system("target", N=10)
    pb["costs"][0]["weights"] = [-1e9, -1e9]
    re = emu.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"])
    assert re["status"][0] == 2


def test_dense_qp_kernel_body(emu, oracle):
    """qp_dense.hpp == QuadProgDenseSolver::SI_solve on the Scilab problem and on random strictly convex QPs with
    equalities, inequalities and partly infinite bounds"""
    P = F.scilab_problem()
    x, fail, it = emu.qp_dense(P["Q"], P["c"], P["Aeq"], P["beq"], P["Aineq"], P["bineq"], P["XL"], P["XU"])
    assert fail[0] == 0 and np.abs(x[0] - P["x_star"]).max() < 1e-9
    rng = np.random.default_rng(3)
    for trial in range(6):
        n, meq, mi = int(rng.integers(3, 20)), int(rng.integers(0, 3)), int(rng.integers(0, 12))
        Mx = rng.standard_normal((n, n))
        Q = Mx @ Mx.T + 0.1 * np.eye(n)
        c = rng.standard_normal(n)
        Aeq, beq = rng.standard_normal((meq, n)), rng.standard_normal(meq)
        Ai, bi = rng.standard_normal((mi, n)), rng.standard_normal(mi) + 1.0
        XL, XU = -np.abs(rng.standard_normal(n)) - 0.2, np.abs(rng.standard_normal(n)) + 0.2
        XL[::3] = -np.inf
        XU[1::4] = np.finfo(float).max
        xo, fo, ito = oracle.quadprog_dense(Q, c, Aeq, beq, Ai, bi, XL, XU)
        x, fail, it = emu.qp_dense(Q, c, Aeq if meq else None, beq if meq else None, Ai if mi else None,
                                   bi if mi else None, XL, XU)
        assert fail[0] == fo
        if fo == 0:
            assert np.abs(x[0] - xo).max() <= 1e-8 * (1 + np.abs(xo).max())


def test_dense_qp_large_kernel_body(emu, oracle):
    """qp_dense_large.hpp / gi_large.hpp (n > 64: one problem per workgroup, J and R in the HBM workspace) takes the same
    decisions as the oracle's qpgen2 restatement: same iteration counts, same solution; plus the status codes"""
    rng = np.random.default_rng(11)
    for n, meq, mi in [(65, 3, 20), (100, 10, 120)]:
        P = F.random_dense_qp(rng, n, meq, mi)
        xo, fo, ito = oracle.quadprog_dense(P["Q"], P["c"], P["Aeq"], P["beq"], P["Aineq"], P["bineq"], P["XL"], P["XU"])
        Qu = np.triu(P["Q"]) + np.tril(np.full_like(P["Q"], np.nan), -1)  # only the upper triangle may be read
        x, fail, it = emu.qp_dense(Qu, P["c"], P["Aeq"], P["beq"], P["Aineq"], P["bineq"], P["XL"], P["XU"])
        assert fo == 0 and fail[0] == 0 and tuple(it[0]) == tuple(ito)
        assert np.abs(x[0] - xo).max() <= 1e-10 * (1 + np.abs(xo).max())
    # not positive definite -> 2 ; infeasible (contradictory bounds through an equality) -> 1
    P = F.random_dense_qp(rng, 70, 0, 4)
    Qbad = P["Q"].copy()
    Qbad[40, 40] = -1.0
    _, fail, _ = emu.qp_dense(Qbad, P["c"], None, None, P["Aineq"], P["bineq"], P["XL"], P["XU"])
    assert fail[0] == 2 and oracle.quadprog_dense(Qbad, P["c"], None, None, P["Aineq"], P["bineq"], P["XL"], P["XU"])[1] == 2
    Aeq = np.zeros((1, 70))
    Aeq[0, 0] = 1.0
    beq = np.array([P["XU"][0] + 1.0])
    _, fail, _ = emu.qp_dense(P["Q"], P["c"], Aeq, beq, None, None, P["XL"], P["XU"])
    assert fail[0] == 1 and oracle.quadprog_dense(P["Q"], P["c"], Aeq, beq, None, None, P["XL"], P["XU"])[1] == 1


@pytest.mark.parametrize("system,xcost", [("bounded", "trajectory"), ("ineq", "mixed"), ("mixed", "trajectory"),
                                          ("eq", "target")])
def test_large_lmpc_reference_fixtures(emu, oracle, system, xcost):
    """More than 64 decision variables (N = 70 on the systems.h fixtures): the workgroup-per-instance kernel body
    (lmpc_large.hpp + gi_large.hpp) builds the same QP and takes the same active-set path as the oracle"""
    pb = getattr(F, system + "_system")(xcost, N=70)
    re = emu.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"], dump_instance=0)
    qp = oracle.lmpc_build(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"])
    assert np.abs(re["Q"] - qp["Q"]).max() <= 1e-12 * np.abs(qp["Q"]).max()
    assert np.abs(re["c"] - qp["c"]).max() <= 1e-11 * max(1.0, np.abs(qp["c"]).max())
    for k in ("Aeq", "Aineq", "beq", "bineq"):
        if qp[k].size:
            assert np.abs(re[k] - qp[k]).max() <= 1e-10 * max(1.0, np.abs(qp[k]).max())
    ro = oracle.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"])
    assert re["status"][0] == ro["status"] == 0 and tuple(re["iter"][0]) == tuple(ro["iter"])
    assert _rel(re["control"][0], ro["control"]) <= RTOL
    assert _rel(re["trajectory"][0], ro["trajectory"]) <= RTOL


@pytest.mark.parametrize("initial_state", [False, True])
def test_large_nine_classes(emu, oracle, initial_state):
    """All nine cost / constraint classes at N = 70 through the workgroup-per-instance body, as LMPC and as
    InitialStateLMPC (72 variables [x0; U]: Q is factorised for E Q^-1 E', then the full Hessian)"""
    pb = F.nine_class_problem(70)
    ist = dict(R=10.0 * np.eye(2), r=np.array([0.1, -0.2]), x0lb=pb["x0"] - 0.05, x0ub=pb["x0"] + 0.05) \
        if initial_state else None
    re = emu.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"], initial_state=ist)
    ro = oracle.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"], initial_state=ist)
    assert re["status"][0] == ro["status"] == 0 and tuple(re["iter"][0]) == tuple(ro["iter"])
    assert _rel(re["control"][0], ro["control"]) <= RTOL
    assert _rel(re["trajectory"][0], ro["trajectory"]) <= RTOL
    if initial_state:
        assert _rel(re["x0_opt"][0], ro["x0_opt"]) <= RTOL


@pytest.mark.parametrize("initial_state", [False, True])
def test_large_full_size_constraint_rows(emu, oracle, initial_state):
    """Full-size constraint entries with more than 64 variables: a terminal-velocity equality (1 x fullXDim) and a
    budget on a weighted sum of the controls (1 x fullUDim), both active at the optimum.  The workgroup evaluates such rows
    cooperatively (lmpc_large.hpp: lhs_cooperative)."""
    N = 70
    pb = F.bounded_system("trajectory", N=N)
    X = 2 * (N + 1)
    E = np.zeros((1, X))
    E[0, X - 1] = 1.0
    cstrs = pb["cstrs"] + [dict(kind="trajectory", E=E, f=[-1.0], ineq=False),
                           dict(kind="control", G=(0.02 * np.arange(N) / N)[None], f=[47.0])]
    ist = dict(R=10.0 * np.eye(2), r=np.array([0.1, -0.2]), x0lb=pb["x0"] - 0.05, x0ub=pb["x0"] + 0.05) \
        if initial_state else None
    re = emu.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], N, pb["costs"], cstrs, initial_state=ist, dump_instance=0)
    ro = oracle.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], N, pb["costs"], cstrs, initial_state=ist)
    qp = oracle.lmpc_build(pb["A"], pb["B"], pb["d"], pb["x0"], N, pb["costs"], cstrs, initial_state=ist)
    for k in ("Aeq", "Aineq", "beq", "bineq"):
        assert np.abs(re[k] - qp[k]).max() <= 1e-10 * max(1.0, np.abs(qp[k]).max())
    assert re["status"][0] == ro["status"] == 0 and tuple(re["iter"][0]) == tuple(ro["iter"])
    assert _rel(re["control"][0], ro["control"]) <= RTOL
    assert abs((0.02 * np.arange(N) / N) @ re["control"][0] - 47.0) <= 1e-6  # the budget row is active (48.4 without it)
    assert abs(re["trajectory"][0][-1] + 1.0) <= 1e-9  # the terminal equality holds


@pytest.mark.parametrize("initial_state", [False, True])
def test_large_full_size_cost_entries(emu, oracle, initial_state, full_size_paths):
    """Full-size COST entries (time-varying references: p and weights of size r (N+1), M spanned by autoSpan) with
    more than 64 variables: rank-4 updates of the Hessian in the HBM workspace (lmpc_large.hpp)"""
    from copra_amd.autospan import autospan_cost
    N = 70
    pb = F.nine_class_problem(N)

    def span(c):
        c = dict(c)
        if c["kind"] == "target":
            return c
        reps = N + 1 if c["kind"] == "trajectory" else N
        ramp = np.linspace(1.0, 0.5, reps)  # genuinely time-varying reference and weights
        c["p"] = (np.atleast_1d(c["p"])[None, :] * ramp[:, None]).ravel()
        c["weights"] = (np.atleast_1d(c["weights"])[None, :] * (2.0 - ramp[:, None])).ravel()
        return autospan_cost(c)

    costs = [span(c) for c in pb["costs"]]
    ist = dict(R=10.0 * np.eye(2), r=np.array([0.1, -0.2]), x0lb=pb["x0"] - 0.05, x0ub=pb["x0"] + 0.05) \
        if initial_state else None
    re = emu.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], N, costs, pb["cstrs"], initial_state=ist, dump_instance=0)
    qp = oracle.lmpc_build(pb["A"], pb["B"], pb["d"], pb["x0"], N, costs, pb["cstrs"], initial_state=ist)
    ro = oracle.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], N, costs, pb["cstrs"], initial_state=ist)
    assert np.abs(re["Q"] - qp["Q"]).max() <= 1e-12 * np.abs(qp["Q"]).max()
    assert np.abs(re["c"] - qp["c"]).max() <= 1e-11 * max(1.0, np.abs(qp["c"]).max())
    assert re["status"][0] == ro["status"] == 0 and tuple(re["iter"][0]) == tuple(ro["iter"])
    assert _rel(re["control"][0], ro["control"]) <= RTOL


def test_shared_model_fast_path(emu, oracle):
    """lmpc_shared.hpp: one (A, B, d) for the whole batch, factorised once by a prepare run of the fused body;
    per instance only xbar = Phi x0 + xi, c = c0 + C1 x0, x = -J J'c and the active-set loop.  Same decisions and the
    same solution as a fresh controller per instance (the oracle), for the compile-time and the generic shapes."""
    from copra_amd import workloads
    wl = workloads.com_preview(10, v_max=0.25, u_max=1.2)
    A, B, d = wl["A"][3], wl["B"][3], wl["d"][3]
    # the headline shape runs it twice: on the Riccati-factor tier in shared-model mode (stage records swept once by a prepare
    # run, copied by every instance: what copra_batch_solve picks for cold starts) and on lmpc_shared.hpp (option no_ric_shared)
    for ric in (True, False):
        if not ric:
            OPTIONS["no_ric_shared"] = 1
        try:
            re = emu.lmpc_solve_shared(A, B, d, wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
        finally:
            OPTIONS.pop("no_ric_shared", None)
        assert re["riccati_factor"] == ric and re["overflowed"] > 0
        for k in range(10):
            ro = oracle.lmpc_solve(A, B, d, wl["x0"][k], wl["N"], wl["costs"], wl["cstrs"])
            assert re["status"][k] == ro["status"] == 0 and tuple(re["iter"][k]) == tuple(ro["iter"])
            assert _rel(re["control"][k], ro["control"]) <= RTOL and _rel(re["trajectory"][k], ro["trajectory"]) <= RTOL
    wl = workloads.com_preview(8, N=15, v_max=0.2, u_max=1.0, seed=3)  # 45 variables: factor-only tier, run-time shape
    A, B, d = wl["A"][1], wl["B"][1], wl["d"][1]
    re = emu.lmpc_solve_shared(A, B, d, wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    for k in range(8):
        ro = oracle.lmpc_solve(A, B, d, wl["x0"][k], wl["N"], wl["costs"], wl["cstrs"])
        assert re["status"][k] == ro["status"] and (ro["status"] != 0 or tuple(re["iter"][k]) == tuple(ro["iter"]))
        assert ro["status"] != 0 or _rel(re["control"][k], ro["control"]) <= RTOL
    pb = F.mixed_system("mixed", N=12)
    x0 = np.tile(pb["x0"], (4, 1))
    x0[:, 1] += np.linspace(-0.5, 0.5, 4)
    re = emu.lmpc_solve_shared(pb["A"], pb["B"], pb["d"], x0, 12, pb["costs"], pb["cstrs"])
    for k in range(4):
        ro = oracle.lmpc_solve(pb["A"], pb["B"], pb["d"], x0[k], 12, pb["costs"], pb["cstrs"])
        assert re["status"][k] == ro["status"] == 0 and tuple(re["iter"][k]) == tuple(ro["iter"])
        assert _rel(re["control"][k], ro["control"]) <= RTOL


def test_shared_model_fast_path_more_than_16_states(emu, oracle):
    """xDim = 18 on the shared-model path (the x0 register cache of lmpc_shared.hpp holds 16 components; the tail is
    read from memory): same answer as a fresh controller per instance"""
    rng = np.random.default_rng(18)
    nx, nu, N, b = 18, 1, 8, 4
    A = np.eye(nx) + 0.05 * rng.standard_normal((nx, nx))
    B = 0.3 * rng.standard_normal((nx, nu))
    d = 0.01 * rng.standard_normal(nx)
    x0 = rng.standard_normal((b, nx))
    costs = [dict(kind="trajectory", M=np.eye(nx), p=np.zeros(nx), weights=np.ones(nx)),
             dict(kind="control", N=np.eye(nu), p=np.zeros(nu), weights=[1e-2])]
    cstrs = [dict(kind="control_bound", lower=[-0.5], upper=[0.5])]
    re = emu.lmpc_solve_shared(A, B, d, x0, N, costs, cstrs)
    for k in range(b):
        ro = oracle.lmpc_solve(A, B, d, x0[k], N, costs, cstrs)
        assert re["status"][k] == ro["status"] == 0
        assert _rel(re["control"][k], ro["control"]) <= RTOL and _rel(re["trajectory"][k], ro["trajectory"]) <= RTOL


def test_per_instance_cost_references(emu, oracle):
    """copra_batch_set_cost_reference: every instance tracks its own goal (one TrajectoryCost(M, p_b) per LMPC in the
    reference).  One-wave kernel (compile-time and generic shape), InitialStateLMPC, and the workgroup kernel."""
    from copra_amd import workloads
    rng = np.random.default_rng(9)
    b = 6
    wl = workloads.com_preview(b)
    goals = wl["costs"][0]["p"][None, :] + 0.2 * rng.standard_normal((b, 6))
    for spec in (True, False):
        re = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"], specialised=spec,
                            cost_refs={0: goals})
        for k in range(b):
            costs = [dict(wl["costs"][0], p=goals[k]), wl["costs"][1]]
            ro = oracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], costs, wl["cstrs"])
            assert re["status"][k] == ro["status"] == 0 and _rel(re["control"][k], ro["control"]) <= RTOL
    for N, ist in ((12, dict(R=10.0 * np.eye(2), r=np.array([0.1, -0.2]))), (70, None), (70, dict(R=10.0 * np.eye(2), r=np.zeros(2)))):
        pb = F.nine_class_problem(N)
        x0 = np.tile(pb["x0"], (3, 1))
        refs = np.array([[0.0, -1.0], [0.01, -0.8], [-0.01, -1.3]])
        io = None if ist is None else dict(ist, x0lb=x0 - 0.05, x0ub=x0 + 0.05)
        re = emu.lmpc_solve(np.tile(pb["A"], (3, 1, 1)), np.tile(pb["B"], (3, 1, 1)), np.tile(pb["d"], (3, 1)), x0, N,
                            pb["costs"], pb["cstrs"], initial_state=io, cost_refs={0: refs})
        for k in range(3):
            costs = [dict(pb["costs"][0], p=refs[k])] + pb["costs"][1:]
            iok = None if ist is None else dict(ist, x0lb=x0[k] - 0.05, x0ub=x0[k] + 0.05)
            ro = oracle.lmpc_solve(pb["A"], pb["B"], pb["d"], x0[k], N, costs, pb["cstrs"], initial_state=iok)
            assert re["status"][k] == ro["status"] == 0 and _rel(re["control"][k], ro["control"]) <= RTOL


def test_per_instance_constraint_rhs_and_bounds(emu, oracle):
    """Per-instance right-hand sides (stacked row order) and control bounds: one-wave kernel and workgroup kernel"""
    rng = np.random.default_rng(13)
    for N in (12, 70):
        pb = F.ineq_system("trajectory", N=N)  # rows: E x_k <= f (N+1 rows), then G u_k <= h (N rows)
        b = 4
        x0 = np.tile(pb["x0"], (b, 1))
        fv = np.array([0.0, -0.2, 0.3, -0.05])  # velocity limit per instance
        hv = np.array([200.0, 150.0, 120.0, 180.0])  # force limit per instance
        rhs = np.hstack([np.repeat(fv[:, None], N + 1, axis=1), np.repeat(hv[:, None], N, axis=1)])
        re = emu.lmpc_solve(np.tile(pb["A"], (b, 1, 1)), np.tile(pb["B"], (b, 1, 1)), np.tile(pb["d"], (b, 1)), x0, N,
                            pb["costs"], pb["cstrs"], row_rhs=rhs)
        for k in range(b):
            cs = [dict(pb["cstrs"][0], f=[fv[k]]), dict(pb["cstrs"][1], f=[hv[k]])]
            ro = oracle.lmpc_solve(pb["A"], pb["B"], pb["d"], x0[k], N, pb["costs"], cs)
            assert re["status"][k] == ro["status"]
            if ro["status"] == 0:
                assert tuple(re["iter"][k]) == tuple(ro["iter"]) and _rel(re["control"][k], ro["control"]) <= RTOL
        pbb = F.bounded_system("trajectory", N=N)
        up = np.repeat(np.array([200.0, 90.0, 140.0, 60.0])[:, None], N, axis=1)
        lo = np.full((b, N), -np.inf)
        re = emu.lmpc_solve(np.tile(pbb["A"], (b, 1, 1)), np.tile(pbb["B"], (b, 1, 1)), np.tile(pbb["d"], (b, 1)), x0, N,
                            pbb["costs"], pbb["cstrs"], bounds=(lo, up))
        for k in range(b):
            cs = [pbb["cstrs"][0], dict(pbb["cstrs"][1], upper=[up[k, 0]])]
            ro = oracle.lmpc_solve(pbb["A"], pbb["B"], pbb["d"], x0[k], N, pbb["costs"], cs)
            assert re["status"][k] == ro["status"] == 0 and _rel(re["control"][k], ro["control"]) <= RTOL
            assert re["control"][k].max() <= up[k, 0] + 1e-6


def test_host_plan_errors(emu):
    """copra_batch_create's dimension checks (plan_builder.hpp) == std::domain_error of TestLMPC.cpp:949-1087"""
    from copra_amd import _capi
    pb = F.ineq_system("target", N=10)
    I5 = np.eye(5)
    bad = [([dict(kind="trajectory", M=I5, p=np.ones(5))], []), ([dict(kind="target", M=I5, p=np.ones(5))], []),
           ([dict(kind="control", N=I5, p=np.ones(5))], []), ([dict(kind="mixed", M=I5, N=I5, p=np.ones(5))], []),
           ([], [dict(kind="trajectory", E=I5, f=np.ones(5))]), ([], [dict(kind="control", G=I5, f=np.ones(5))]),
           ([], [dict(kind="mixed", E=I5, G=I5, f=np.ones(5))]),
           ([], [dict(kind="trajectory_bound", lower=np.ones(3), upper=np.ones(3))]),
           ([], [dict(kind="control_bound", lower=np.ones(3), upper=np.ones(3))])]
    for costs, cstrs in bad:
        with pytest.raises(_capi.CopraDomainError):
            emu.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], costs, cstrs)
    for costs, cstrs in [([dict(kind="trajectory", M=I5, p=np.ones(2))], []),
                         ([], [dict(kind="control_bound", lower=np.ones(3), upper=np.ones(2))])]:
        with pytest.raises(_capi.CopraDomainError):  # rows mismatch (try autoSpan)
            emu.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], costs, cstrs)
    two_bounds = [dict(kind="control_bound", lower=[-1.0], upper=[1.0])] * 2  # TestLMPC.cpp:1084-1086
    with pytest.raises(_capi.CopraRuntimeError):
        emu.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], [], two_bounds)


def test_two_tier_execution_overflow_queue(emu, oracle):
    """Occupancy design: the first launch uses a compact <= 40 KiB LDS layout whose R holds `rcap` < n active
    constraints; instances that need more are queued and redone by a second launch with the full layout.  A workload
    with very tight bounds forces that path; results must still match the oracle (incl. infeasible instances)."""
    from copra_amd import workloads
    wl = workloads.com_preview(24, v_max=0.12, u_max=0.8, seed=9)
    re = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    ro = oracle.lmpc_solve_batch(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    assert re["lds_bytes"] <= 40 * 1024 and re["rcap"] < 60
    assert re["overflowed"] > 0  # the second tier really ran
    assert (re["status"] == ro["status"]).all() and (re["iter"] == ro["iter"]).all()
    ok = ro["status"] == 0
    assert ok.any() and _rel(re["control"][ok], ro["control"][ok]) <= RTOL


def test_dense_mfma_hessian_equals_structured_path(emu, oracle, full_size_paths):
    """The headline TrajectoryCost handed over as a full-size entry (M = blockdiag, 126 x 126) must give the same QP
    as the per-step entry: dense v_mfma_f64_16x16x4 contraction vs block-diagonal prefix sums vs the oracle."""
    from copra_amd import workloads
    from copra_amd.autospan import autospan_cost
    wl = workloads.com_preview(3)
    dense = [autospan_cost(dict(wl["costs"][0], p=np.tile(wl["costs"][0]["p"], 21))), wl["costs"][1]]
    rd = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], dense, wl["cstrs"], dump_instance=1)
    rs = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"], dump_instance=1)
    qp = oracle.lmpc_build(wl["A"][1], wl["B"][1], wl["d"][1], wl["x0"][1], wl["N"], dense, wl["cstrs"])
    scale = np.abs(qp["Q"]).max()
    assert np.abs(rd["Q"] - qp["Q"]).max() <= 1e-12 * scale
    assert np.abs(rd["Q"] - rs["Q"]).max() <= 1e-12 * scale
    assert np.abs(rd["c"] - qp["c"]).max() <= 1e-11 * max(1.0, np.abs(qp["c"]).max())
    assert _rel(rd["control"], rs["control"]) <= 1e-9


@pytest.mark.parametrize("q1", ["registers", "lds"])
def test_dense_mfma_path_first_tier_layouts(emu, oracle, full_size_paths, monkeypatch, q1):
    """the first tier of that path as copra_batch_solve runs it (no dump): five columns of Q1 in REGISTERS (7 instances per CU; round 4)
    or eleven in LDS (6 per CU, option no_q1regs) -- a constrained batch, results, statuses and both iteration counters == oracle,
    instances that outgrow five columns finish in the second tier"""
    from copra_amd import workloads
    from copra_amd.autospan import autospan_cost
    if q1 == "lds":
        monkeypatch.setitem(OPTIONS, "no_q1regs", 1)
    wl = workloads.com_preview(20, v_max=0.3, u_max=1.5, seed=9)
    dense = [autospan_cost(dict(wl["costs"][0], p=np.tile(wl["costs"][0]["p"], 21))), wl["costs"][1]]
    re = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], dense, wl["cstrs"])
    ro = oracle.lmpc_solve_batch(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], dense, wl["cstrs"])
    assert re["factor_only"] and re["rcap"] == (5 if q1 == "registers" else 11)
    assert re["lds_bytes"] * (7 if q1 == "registers" else 6) <= 160 * 1024
    assert (re["status"] == ro["status"]).all() and (re["iter"] == ro["iter"]).all()
    assert ro["iter"][:, 0].max() > 6 and (q1 == "lds" or re["overflowed"] > 0)
    assert _rel(re["control"], ro["control"]) <= RTOL and _rel(re["trajectory"], ro["trajectory"]) <= RTOL


def _is_problem_batch(b=5):
    pb = F.bounded_system("trajectory", N=12)
    rng = np.random.default_rng(1)
    x0 = np.tile(pb["x0"], (b, 1))
    x0[:, 1] += rng.uniform(-0.5, 0.5, b)
    A, B, d = np.tile(pb["A"], (b, 1, 1)), np.tile(pb["B"], (b, 1, 1)), np.tile(pb["d"], (b, 1))
    ist = dict(R=10.0 * np.eye(2), r=np.array([0.1, -0.2]), x0lb=x0 - 0.05, x0ub=x0 + 0.05)
    return pb, A, B, d, x0, ist


def test_initial_state_lmpc_batch(emu, oracle):
    """InitialStateLMPC (src/InitialStateLMPC.cpp:77-128) on the device: decision vector [x0; U], per-instance x0
    bounds; control, trajectory and x0* against the oracle"""
    pb, A, B, d, x0, ist = _is_problem_batch()
    re = emu.lmpc_solve(A, B, d, x0, 12, pb["costs"], pb["cstrs"], initial_state=ist)
    for k in range(len(x0)):
        ro = oracle.lmpc_solve(A[k], B[k], d[k], x0[k], 12, pb["costs"], pb["cstrs"],
                               initial_state=dict(R=ist["R"], r=ist["r"], x0lb=ist["x0lb"][k], x0ub=ist["x0ub"][k]))
        assert ro["status"] == re["status"][k] == 0
        assert _rel(re["control"][k], ro["control"]) <= RTOL
        assert _rel(re["trajectory"][k], ro["trajectory"]) <= RTOL
        assert _rel(re["x0_opt"][k], ro["x0_opt"]) <= RTOL
        assert (re["x0_opt"][k] <= ist["x0ub"][k] + 1e-6).all() and (re["x0_opt"][k] >= ist["x0lb"][k] - 1e-6).all()


@pytest.mark.parametrize("full_size", [False, True])
def test_initial_state_lmpc_reference_test_problem(emu, oracle, full_size):
    """tests/TestLMPC_InitialState.cpp:266-403 (all nine classes, x0 free in [-1,1], R = 1e-6 I; run_optimization_test
    with per-step and with full-size entries) and :29-260 (trailing
    blocks of the InitialStateLMPC QP equal the LMPC QP).  With R = 1e-6 the Hessian's Schur complement is 1e-6
    against blocks of 1e6, so U itself is not determined to 1e-6 by ANY arithmetic; like the reference's test we check
    solve() == true and x0* within its bounds, plus x0* and the QP against the oracle."""
    pb = F.initial_state_problem(full_size)
    ist = dict(R=1e-6 * np.eye(2), r=np.zeros(2), x0lb=-np.ones(2), x0ub=np.ones(2))
    re = emu.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"], initial_state=ist,
                        dump_instance=0)
    ro = oracle.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"], initial_state=ist)
    assert re["status"][0] == ro["status"] == 0
    x0s = re["x0_opt"][0]
    assert (x0s <= 1 + 1e-6).all() and (x0s >= -1 - 1e-6).all()
    assert np.abs(x0s - ro["x0_opt"]).max() <= 1e-6
    qb = oracle.lmpc_build(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"], initial_state=ist)
    qa = oracle.lmpc_build(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"])
    scale = np.abs(qb["Q"]).max()
    assert np.abs(re["Q"] - qb["Q"]).max() <= 1e-9 * scale
    assert np.abs(re["c"] - qb["c"]).max() <= 1e-9 * max(1.0, np.abs(qb["c"]).max())
    assert np.abs(re["Aineq"] - qb["Aineq"]).max() <= 1e-10 * max(1.0, np.abs(qb["Aineq"]).max())
    assert np.abs(re["bineq"] - qb["bineq"]).max() <= 1e-10
    # TestLMPC_InitialState.cpp:212-226 on the device-built matrices
    assert np.abs(re["Q"][2:, 2:] - qa["Q"]).max() <= 1e-6
    assert np.abs(re["Aineq"][:, 2:] - qa["Aineq"]).max() <= 1e-6
    # better conditioned variant: U must match too
    ist2 = dict(R=np.eye(2), r=np.array([0.3, -0.2]), x0lb=-np.ones(2), x0ub=np.ones(2))
    re2 = emu.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"], initial_state=ist2)
    ro2 = oracle.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"], initial_state=ist2)
    assert re2["status"][0] == ro2["status"] == 0 and _rel(re2["control"][0], ro2["control"]) <= RTOL


def test_edge_cases_of_the_reference_path(emu, oracle):
    """tests/edge_cases.py on the kernel bodies: quirk Q1 with a finite lower bound, duplicate rows, the linearly
    dependent opposite pair (status parity), opposite state rows that x0 violates, default x0 bounds (x0lb == x0ub)"""
    import edge_cases as E
    pb, quirk, _ = E.finite_lower_trajectory_bound()
    re, ro = _compare(emu, oracle, pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], quirk)
    assert ro["status"][0] == 0 and re["trajectory"][0].reshape(-1, 2)[:, 1].max() <= -4.0 + 1e-9
    dup, opp = E.duplicate_and_opposite_rows()
    _compare(emu, oracle, dup["A"], dup["B"], dup["d"], dup["x0"], dup["N"], dup["costs"], dup["cstrs"], same_iters=False)
    re, ro = _compare(emu, oracle, opp["A"], opp["B"], opp["d"], opp["x0"], opp["N"], opp["costs"], opp["cstrs"],
                      same_iters=False)
    assert ro["status"][0] == 1
    q = E.opposite_state_rows_infeasible()
    re, ro = _compare(emu, oracle, q["A"], q["B"], q["d"], q["x0"], q["N"], q["costs"], q["cstrs"], same_iters=False)
    assert ro["status"][0] == 1
    pb = F.bounded_system("trajectory", N=12)
    ist = dict(R=10.0 * np.eye(2), r=np.array([0.1, -0.2]), x0lb=pb["x0"][None], x0ub=pb["x0"][None])
    re = emu.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], 12, pb["costs"], pb["cstrs"], initial_state=ist)
    ro = oracle.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], 12, pb["costs"], pb["cstrs"])  # the plain LMPC
    assert re["status"][0] == 0 and np.abs(re["x0_opt"][0] - pb["x0"]).max() < 1e-12
    assert _rel(re["control"][0], ro["control"]) <= RTOL


def _dense_twins(oracle, N):
    """a TrajectoryCost and a TrajectoryConstraint of the falling-mass fixture evaluated on the host (what a user
    subclass's update() does, costFunctions.cpp:63-82 / constraints.cpp:66-84) -> COPRA_COST_DENSE / COPRA_CSTR_DENSE"""
    pb = F.ineq_system("trajectory", N=N)
    A, B, d, x0 = pb["A"], pb["B"], pb["d"], pb["x0"]
    Phi, Psi, xi = oracle.preview(A, B, d, N)
    c0, k0 = pb["costs"][0], pb["cstrs"][0]
    M, p, w = np.atleast_2d(c0["M"]), np.asarray(c0["p"], float), np.asarray(c0["weights"], float)
    Q, E, f = np.zeros((N, N)), np.zeros((2, N)), np.zeros(N)
    for i in range(N + 1):
        tmp = M @ Psi[2 * i:2 * i + 2]
        Q += tmp.T @ (w[:, None] * tmp)
        E += (M @ Phi[2 * i:2 * i + 2]).T @ (w[:, None] * tmp)
        f += (M @ xi[2 * i:2 * i + 2] - p) @ (w[:, None] * tmp)
    Em, fm = np.atleast_2d(k0["E"]), np.asarray(k0["f"], float)
    Ar = np.vstack([Em @ Psi[2 * i:2 * i + 2] for i in range(N + 1)])
    Y = np.vstack([Em @ Phi[2 * i:2 * i + 2] for i in range(N + 1)])
    z = np.concatenate([fm - Em @ xi[2 * i:2 * i + 2] for i in range(N + 1)])
    costs = [dict(kind="dense", Q=Q, c=E.T @ x0 + f, E=E, f=f), pb["costs"][1]]
    cstrs = [dict(kind="dense", A=Ar, b=z - Y @ x0, Y=Y, z=z), pb["cstrs"][1]]
    return pb, costs, cstrs


@pytest.mark.parametrize("N", [12, 80])
def test_host_evaluated_user_pieces_dense_kinds(emu, oracle, N):
    """COPRA_COST_DENSE / COPRA_CSTR_DENSE (plug-in point 2: a user subclass evaluated on the host) in the one-wave kernels
    (N = 12) and the workgroup kernel (N = 80), LMPC and InitialStateLMPC: same QP and same solution as the built-in twin"""
    pb, costs, cstrs = _dense_twins(oracle, N)
    args = (pb["A"], pb["B"], pb["d"], pb["x0"], N)
    ro = oracle.lmpc_solve(*args, pb["costs"], pb["cstrs"])
    qo = oracle.lmpc_build(*args, pb["costs"], pb["cstrs"])
    re = emu.lmpc_solve(*args, costs, cstrs, dump_instance=0)
    assert re["status"][0] == ro["status"] == 0 and tuple(re["iter"][0]) == tuple(ro["iter"])
    assert _rel(re["control"][0], ro["control"]) <= RTOL
    assert np.abs(re["Q"] - qo["Q"]).max() <= 1e-12 * np.abs(qo["Q"]).max()
    assert np.abs(re["c"] - qo["c"]).max() <= 1e-10 and np.abs(re["Aineq"] - qo["Aineq"]).max() <= 1e-12
    ist = dict(R=10.0 * np.eye(2), r=np.array([0.1, -0.2]), x0lb=pb["x0"] - 0.05, x0ub=pb["x0"] + 0.05)
    ro = oracle.lmpc_solve(*args, pb["costs"], pb["cstrs"], initial_state=ist)
    re = emu.lmpc_solve(*args, costs, cstrs, initial_state=ist)
    assert re["status"][0] == ro["status"] == 0
    assert _rel(re["control"][0], ro["control"]) <= RTOL and np.abs(re["x0_opt"][0] - ro["x0_opt"]).max() <= 1e-9
    assert emu.lmpc_solve_riccati(*args, costs, cstrs) is None  # dense pieces couple all steps: not stage-wise


def test_warm_start_across_receding_horizon_ticks(emu, oracle):
    """copra_batch_set_warm_start (lmpc_shared.hpp + gi_core.hpp): the previous tick's active set, moved one step towards
    the present, is activated first.  Same U / status as a cold start and as the oracle on every tick; fewer constraint
    scans (iterations count forced activations + ONE closing scan instead of one scan per activation)"""
    from copra_amd import workloads
    wl = workloads.com_preview(6, v_max=0.25, u_max=1.2, seed=4)
    A, B, d = wl["A"][2], wl["B"][2], wl["d"][2]
    x = wl["x0"].copy()
    warm = np.full((6, emu.WARM_CAP), -1, dtype=np.int32)
    rng = np.random.default_rng(0)
    it_cold = it_warm = 0
    reused = 0
    for tick in range(6):
        rw = emu.lmpc_solve_shared(A, B, d, x, wl["N"], wl["costs"], wl["cstrs"], warm=warm)
        rc = emu.lmpc_solve_shared(A, B, d, x, wl["N"], wl["costs"], wl["cstrs"])
        for k in range(6):
            ro = oracle.lmpc_solve(A, B, d, x[k], wl["N"], wl["costs"], wl["cstrs"])
            assert rw["status"][k] == rc["status"][k] == ro["status"]
            if ro["status"] == 0:
                assert _rel(rw["control"][k], ro["control"]) <= RTOL and _rel(rc["control"][k], ro["control"]) <= RTOL
        ok = rw["status"] == 0
        it_cold += int(rc["iter"][ok, 0].sum())
        it_warm += int(rw["iter"][ok, 0].sum())
        reused += int((warm >= 0).sum())
        nxt = rw["trajectory"][:, 6:12].copy()
        nxt[:, :3] += 0.002 * rng.standard_normal((6, 3))
        x = np.where(ok[:, None], nxt, x)
    assert reused > 0  # active sets were carried over ...
    assert it_warm <= it_cold + 6 * 6  # ... and never cost more than the closing scan per solve


def test_riccati_factor_tier_body_with_one_control(emu, oracle, monkeypatch):
    """... and with ONE control (the reference's falling-mass fixtures, tests/systems.h: 40 steps; BASELINE configs[1]'s shape
    (2, 1, 10)): the stacked layout of the recursions and the 1 x 1 control block work as they are"""
    from copra_amd import workloads
    monkeypatch.setenv("COPRA_EMU_WANT_RIC", "1")
    wl = workloads.double_integrator(6)
    _, _ = _compare(emu, oracle, wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    for system, xcost in (("bounded", "trajectory"), ("ineq", "mixed"), ("mixed", "target")):
        pb = getattr(F, system + "_system")(xcost, N=40)
        re = emu.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"])
        ro = oracle.lmpc_solve(pb["A"], pb["B"], pb["d"], pb["x0"], pb["N"], pb["costs"], pb["cstrs"])
        assert re["riccati_factor"] and re["status"][0] == ro["status"] == 0 and tuple(re["iter"][0]) == tuple(ro["iter"])
        assert _rel(re["control"][0], ro["control"]) <= RTOL


@pytest.mark.parametrize("shape", ["com12", "planar16", "five_states"])
def test_riccati_factor_tier_body_on_other_shapes(emu, oracle, shape, monkeypatch):
    """the body of the Riccati-factor tier instantiated for shapes beyond the library's three CoM horizons -- what
    copra_batch_specialise compiles at run time: (6, 3, 12), (4, 2, 16), (5, 3, 12); the emulator takes the tier's layout the
    way the library does after the compilation (take_ric_layout).  Same statuses, iteration counts and controls as the oracle,
    through the overflow tier where the active set outgrows the five register columns"""
    from copra_amd import workloads
    monkeypatch.setenv("COPRA_EMU_WANT_RIC", "1")
    rng = np.random.default_rng(1)
    b = 6
    if shape == "com12":
        wl = workloads.com_preview(b, N=12, v_max=0.3, u_max=1.5, seed=3)
        args = (wl["A"], wl["B"], wl["d"], wl["x0"], 12, wl["costs"], wl["cstrs"])
    elif shape == "planar16":
        T = 0.1
        A = np.tile(np.block([[np.eye(2), T * np.eye(2)], [np.zeros((2, 2)), np.eye(2)]]), (b, 1, 1))
        B = np.tile(np.vstack([0.5 * T * T * np.eye(2), T * np.eye(2)]), (b, 1, 1))
        x0 = np.hstack([rng.normal(0, 0.3, (b, 2)), rng.uniform(-0.2, 0.2, (b, 2))])
        costs = [dict(kind="trajectory", M=np.eye(4), p=np.array([1.0, 0.5, 0.0, 0.0]), weights=[10, 10, 1, 1]),
                 dict(kind="control", N=np.eye(2), p=np.zeros(2), weights=[1e-3] * 2)]
        cstrs = [dict(kind="trajectory_bound", lower=[-np.inf] * 4, upper=[np.inf, np.inf, 0.4, 0.4]),
                 dict(kind="control_bound", lower=[-1.5] * 2, upper=[1.5] * 2)]
        args = (A, B, np.zeros((b, 4)), x0, 16, costs, cstrs)
    else:
        Q, _ = np.linalg.qr(rng.standard_normal((5, 5)))
        A = np.tile(0.95 * Q, (b, 1, 1))
        B = np.tile(0.4 * rng.standard_normal((5, 3)), (b, 1, 1))
        x0 = rng.standard_normal((b, 5))
        costs = [dict(kind="trajectory", M=np.eye(5), p=np.zeros(5), weights=[5.0] * 5),
                 dict(kind="control", N=np.eye(3), p=np.zeros(3), weights=[1e-2] * 3)]
        cstrs = [dict(kind="control_bound", lower=[-0.4] * 3, upper=[0.4] * 3)]
        args = (A, B, 0.01 * rng.standard_normal((b, 5)), x0, 12, costs, cstrs)
    re = emu.lmpc_solve(*args)
    assert re["riccati_factor"]
    ro = oracle.lmpc_solve_batch(*args)
    assert (re["status"] == ro["status"]).all() and (re["iter"] == ro["iter"]).all()
    ok = ro["status"] == 0
    assert ok.any() and _rel(re["control"][ok], ro["control"][ok]) <= RTOL and _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= RTOL
    assert (re["iter"][:, 0] > 1).any()


@pytest.mark.parametrize("steps", [1, 2])
def test_riccati_factor_tier_ladder_steps(emu, oracle, steps, monkeypatch):
    """the layout ladder of the headline's tier (Q1 moves to LDS, more columns per step: what adapt_layout does after solves that
    overflowed): the tight workload (3 .. 22 active constraints) on its first two steps, against the oracle incl. iteration counts"""
    from copra_amd import workloads
    wl = workloads.com_preview(10, v_max=0.25, u_max=1.2, seed=4)
    base = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    monkeypatch.setenv("COPRA_EMU_LADDER_STEPS", str(steps))
    re, ro = _compare(emu, oracle, wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    assert re["riccati_factor"] and re["rcap"] > base["rcap"] and re["overflowed"] <= base["overflowed"]


@pytest.mark.parametrize("mode", ["spec", "handover", "filter_only", "off"])
@pytest.mark.parametrize("batch,N,vmax,umax", [(150, 20, 0.6, 3.0), (70, 15, 0.3, 1.5), (64, 10, 0.6, 3.0)])
def test_one_instance_per_lane_pass(emu, oracle, monkeypatch, mode, batch, N, vmax, umax, no_axis):
    """lmpc_lane.hpp in front of the Riccati-factor tier: LQ sweep + roll-out with one instance per LANE (64 instances per wave; the last
    wave of the batch is ragged).  It must finish EXACTLY the instances whose unconstrained minimiser violates nothing (the oracle's
    iteration count (1, 0): qpgen2's first scan finds nothing) and leave the others to the first tier, which takes the factor over
    (`handover`: K | kv | Lam^-1 and the row-norm sums come from the pass, no sweep and no roll-out there) or sweeps itself
    (`filter_only`); `off`: the tier alone.  Statuses, BOTH iteration counters, U and X against the oracle in all three."""
    from copra_amd import workloads
    if mode == "off":
        monkeypatch.setitem(OPTIONS, "no_lane_pass", 1)
    if mode != "spec":  # (round 5: by default the pass takes the first steps of the iteration itself and hands nothing over)
        monkeypatch.setitem(OPTIONS, "no_lane_spec", 1)
    if mode == "filter_only":
        monkeypatch.setitem(OPTIONS, "no_lane_handover", 1)
    wl = workloads.com_preview(batch, N=N, v_max=vmax, u_max=umax, seed=9)
    args = (wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    re = emu.lmpc_solve(*args)
    ro = oracle.lmpc_solve_batch(*args, nthreads=8)
    assert re["riccati_factor"]
    assert (re["status"] == ro["status"]).all() and (re["iter"] == ro["iter"]).all()
    ok = ro["status"] == 0
    assert _rel(re["control"][ok], ro["control"][ok]) <= 1e-9 and _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= 1e-9
    at_minimiser = int(((ro["iter"][:, 0] == 1) & ok).sum())
    one_bound = int(((ro["iter"][:, 0] >= 2) & (ro["iter"][:, 0] <= 4) & (ro["iter"][:, 1] == 0) & ok).sum())  # (up to one bound on u_0 per axis: (2, 0) ... (4, 0))
    if mode == "spec":  # (round 5: the pass also takes the first TWO steps of the iteration where bounds on u_0 are the picks)
        assert at_minimiser <= re["lane_pass_finished"] <= at_minimiser + one_bound
    else:
        assert re["lane_pass_finished"] == (-1 if mode == "off" else at_minimiser)
    if vmax == 0.6:
        assert 0 < at_minimiser < batch  # (both kinds of instance in the batch)


def test_one_instance_per_lane_pass_own_bounds_and_skips(emu, oracle, no_axis):
    """per-instance control bounds, cost references and right-hand sides all go through the pass (every lane reads its own bounds and
    its own row of right-hand sides, rebuilds its own affine cost terms)"""
    from copra_amd import workloads
    b = 24
    wl = workloads.com_preview(b, v_max=0.6, u_max=3.0, seed=12)
    rng = np.random.default_rng(3)
    n = 3 * wl["N"]
    umax = 3.0 * rng.uniform(0.3, 1.2, b)
    ub = np.repeat(umax[:, None], n, axis=1)
    lb = -0.8 * ub
    args = (wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    re = emu.lmpc_solve(*args, bounds=(lb, ub))
    finished = one_bound = 0
    for k in range(b):
        cs = [wl["cstrs"][0], dict(wl["cstrs"][1], lower=[-0.8 * umax[k]] * 3, upper=[umax[k]] * 3)]
        ro = oracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], wl["costs"], cs)
        assert re["status"][k] == ro["status"]
        if ro["status"] == 0:
            assert tuple(re["iter"][k]) == tuple(ro["iter"]) and _rel(re["control"][k], ro["control"]) <= 1e-9
            finished += int(ro["iter"][0] == 1)
            one_bound += int(tuple(ro["iter"]) in ((2, 0), (3, 0), (4, 0)))
    assert 0 < finished <= re["lane_pass_finished"] <= finished + one_bound  # (+ the first steps the pass takes itself: round 5)
    # per-instance cost references (every instance its own goal): the pass rebuilds its affine terms per lane from the plan's coefficient
    # table
    refs = {0: np.tile(wl["costs"][0]["p"], (b, 1)) + 0.03 * rng.standard_normal((b, 6))}
    re2 = emu.lmpc_solve(*args, cost_refs=refs)
    finished = one_bound = 0
    for k in range(b):
        cs = [dict(wl["costs"][0], p=refs[0][k]), wl["costs"][1]]
        ro = oracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], cs, wl["cstrs"])
        assert re2["status"][k] == ro["status"]
        if ro["status"] == 0:
            assert tuple(re2["iter"][k]) == tuple(ro["iter"]) and _rel(re2["control"][k], ro["control"]) <= 1e-9
            finished += int(ro["iter"][0] == 1)
            one_bound += int(tuple(ro["iter"]) in ((2, 0), (3, 0), (4, 0)))
    assert 0 < finished <= re2["lane_pass_finished"] <= finished + one_bound  # (+ the first steps the pass takes itself: round 5)
    # ... and per-instance right-hand sides (every instance its own velocity limit): every lane reads its own row of the table
    vlim = 0.6 * rng.uniform(0.7, 1.2, b)
    re3 = emu.lmpc_solve(*args, row_rhs=np.repeat(vlim[:, None], 63, axis=1))
    finished = one_bound = 0
    inf = np.inf
    for k in range(b):
        cs = [dict(wl["cstrs"][0], upper=[inf, inf, inf, vlim[k], vlim[k], vlim[k]]), wl["cstrs"][1]]
        ro = oracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], wl["costs"], cs)
        assert re3["status"][k] == ro["status"]
        if ro["status"] == 0:
            assert tuple(re3["iter"][k]) == tuple(ro["iter"]) and _rel(re3["control"][k], ro["control"]) <= 1e-9
            finished += int(ro["iter"][0] == 1)
            one_bound += int(tuple(ro["iter"]) in ((2, 0), (3, 0), (4, 0)))
    assert 0 < finished <= re3["lane_pass_finished"] <= finished + one_bound  # (+ the first steps the pass takes itself: round 5)


def test_lane_pass_skips_the_gains_between_decoupled_axes_only_where_a_whole_wave_is_decoupled(emu, oracle, monkeypatch, no_axis):
    """the CoM model is three decoupled double integrators and its costs couple no two axes (FusedPlan::lane_axes): the pass neither writes
    nor reads the twelve gains K(c, j), j % 3 != c, which are exactly zero then.  The systems are checked per WAVE: with ONE instance whose
    A or B couples two axes its wave keeps every entry, the other waves do not -- results equal to the oracle's either way, and equal to
    the decoupled batch's on the instances that were not touched"""
    from copra_amd import workloads
    b = 130  # (three waves)
    wl = workloads.com_preview(b, seed=41)
    args = lambda A, B: (A, B, wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])  # noqa: E731
    base = emu.lmpc_solve(*args(wl["A"], wl["B"]))
    # the structured sweep and roll-out leave out products that are exactly zero, nothing else: bit for bit the dense ones' results
    monkeypatch.setitem(OPTIONS, "no_lane_axes", 1)
    dense = emu.lmpc_solve(*args(wl["A"], wl["B"]))
    monkeypatch.setitem(OPTIONS, "no_lane_axes", 0)
    # (what ends in the PASS in both runs: the tier behind it sweeps axis by axis too on decoupled systems -- scalar arithmetic instead of
    #  matrix instructions, equal to rounding -- and a bound on u_0 in every axis ends in the pass on decoupled axes only)
    differ = (dense["control"] != base["control"]).any(axis=1) | (dense["trajectory"] != base["trajectory"]).any(axis=1)
    assert differ.sum() <= b - dense["lane_pass_finished"] and dense["lane_pass_finished"] > b // 2
    both = ~differ
    assert (dense["iter"] == base["iter"]).all() and (dense["status"] == base["status"]).all()
    assert _rel(dense["control"], base["control"]) <= 1e-9 and _rel(dense["trajectory"], base["trajectory"]) <= 1e-9
    A2, B2 = wl["A"].copy(), wl["B"].copy()
    A2[5, 0, 4] = 0.03  # (x position picks up y velocity: instance 5, first wave)
    B2[70, 3, 1] = 0.02  # (x velocity driven by the y control: instance 70, second wave)
    re = emu.lmpc_solve(*args(A2, B2))
    ro = oracle.lmpc_solve_batch(A2, B2, wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"], nthreads=8)
    ok = ro["status"] == 0
    assert (re["status"] == ro["status"]).all() and ok.sum() >= b - 2
    assert (re["iter"][ok] == ro["iter"][ok]).all()
    assert _rel(re["control"][ok], ro["control"][ok]) <= 1e-9 and _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= 1e-9
    same = np.ones(b, dtype=bool)
    same[[5, 70]] = False
    third = np.arange(b) >= 128  # (the wave nobody touched: bit for bit; in the two others a third bound on u_0 ends in the tier now)
    assert np.array_equal(re["control"][third], base["control"][third]) and (re["iter"][same] == base["iter"][same]).all()
    assert ((re["control"][same] != base["control"][same]).any(axis=1)).sum() <= b - re["lane_pass_finished"]
    assert _rel(re["control"][same], base["control"][same]) <= 1e-9
    assert not np.allclose(re["control"][5], base["control"][5]) and not np.allclose(re["control"][70], base["control"][70])
    # ... and a cost that couples the axes (a dense output map): the plan says so, nothing is skipped anywhere
    rng = np.random.default_rng(3)
    costs = [dict(wl["costs"][0], M=np.eye(6) + 0.1 * rng.standard_normal((6, 6))), wl["costs"][1]]
    rc = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], costs, wl["cstrs"])
    rco = oracle.lmpc_solve_batch(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], costs, wl["cstrs"], nthreads=8)
    okc = rco["status"] == 0
    assert (rc["status"] == rco["status"]).all() and (rc["iter"][okc] == rco["iter"][okc]).all()
    assert _rel(rc["control"][okc], rco["control"][okc]) <= RTOL  # (a general output map: conditioning, as everywhere in this file)


def test_a_row_that_couples_two_axes_keeps_the_pass_off_the_axis_by_axis_steps(emu, oracle):
    """on decoupled axes the pass takes its speculative steps axis by axis, all in one trajectory (lane_spec_axes) -- right as long as the iterates
    BETWEEN the axes' steps cannot violate anything.  A control row u_x + u_y <= c can hold at the minimiser and after both steps and fail in
    between, where qpgen2 would add it: the plan builder sees the row and the controller runs the dense builds with the step-by-step levels
    (FusedPlan::lane_axes = 0).  Counters and results against the oracle, with bounds tight enough that two axes saturate at once"""
    from copra_amd import workloads
    b = 128
    wl = workloads.com_preview(b, seed=47, u_max=1.5)
    ro0 = oracle.lmpc_solve_batch(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"], nthreads=8)
    assert ((ro0["iter"][:, 0] >= 3) & (ro0["iter"][:, 1] == 0)).sum() >= 8  # (instances with bounds on u_0 in two axes)
    for c in (2.2, 2.6, 2.9):
        cstrs = list(wl["cstrs"]) + [dict(kind="control", G=[[1.0, 1.0, 0.0]], f=[c]), dict(kind="control", G=[[-1.0, -1.0, 0.0]], f=[c])]
        args = (wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], cstrs)
        re = emu.lmpc_solve(*args)
        ro = oracle.lmpc_solve_batch(*args, nthreads=8)
        ok = ro["status"] == 0
        assert (re["status"] == ro["status"]).all() and ok.sum() >= b - 2
        assert (re["iter"][ok] == ro["iter"][ok]).all(), c
        assert _rel(re["control"][ok], ro["control"][ok]) <= 1e-9 and _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= 1e-9


def test_lane_pass_with_more_rows_per_step_than_its_prefetch_buffer_holds(emu, oracle, no_axis):
    """six rows per step (upper AND lower velocity limits as rows) in front of the pass, whose stage buffer carries the right-hand sides of
    the first four (lmpc_lane.hpp: RQ): the others are read in place -- with the controller's right-hand sides and with every instance's own"""
    from copra_amd import workloads
    b = 64
    wl = workloads.com_preview(b, seed=33, v_max=0.45)
    vsel = np.hstack([np.zeros((3, 3)), np.eye(3)])
    cstrs = list(wl["cstrs"]) + [dict(kind="trajectory", E=-vsel, f=[0.45] * 3, ineq=True)]
    args = (wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], cstrs)
    rng = np.random.default_rng(5)
    for own in (False, True):
        vup, vlo = (0.45 * rng.uniform(0.8, 1.2, b), 0.45 * rng.uniform(0.8, 1.2, b)) if own else (np.full(b, 0.45), np.full(b, 0.45))
        # stacked order of the rows: the trajectory bound's (N + 1) x 3, then the row constraint's (N + 1) x 3
        rhs = np.hstack([np.repeat(vup[:, None], 63, axis=1), np.repeat(vlo[:, None], 63, axis=1)]) if own else None
        re = emu.lmpc_solve(*args, row_rhs=rhs)
        nsolved = 0
        inf = np.inf
        for k in range(b):
            cs = [dict(cstrs[0], upper=[inf, inf, inf, vup[k], vup[k], vup[k]]), cstrs[1], dict(cstrs[2], f=[vlo[k]] * 3)]
            ro = oracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], wl["costs"], cs)
            assert re["status"][k] == ro["status"], (own, k)
            if ro["status"] == 0:
                nsolved += 1
                assert tuple(re["iter"][k]) == tuple(ro["iter"]), (own, k)
                assert _rel(re["control"][k], ro["control"]) <= 1e-9 and _rel(re["trajectory"][k], ro["trajectory"]) <= 1e-9
        assert nsolved >= b // 2 and 0 < re["lane_pass_finished"] < b


def test_one_instance_per_lane_pass_shared_model(emu, oracle, monkeypatch):
    """the shared-model form of the pass (lmpc_lane_shared_body: the batch-wide stage records as scalar operands, only the roll-out from
    each x0 is left) in front of the Riccati-factor tier in shared-model mode: it finishes exactly the instances at their unconstrained
    minimiser, the tier starts the others from the U and X it left; against the oracle and against the tier alone"""
    from copra_amd import workloads
    b = 70
    wl = workloads.com_preview(b, seed=21)
    A, B, d = wl["A"][3], wl["B"][3], wl["d"][3]
    ro = oracle.lmpc_solve_batch(np.tile(A, (b, 1, 1)), np.tile(B, (b, 1, 1)), np.tile(d, (b, 1)), wl["x0"], wl["N"], wl["costs"], wl["cstrs"],
                                 nthreads=8)
    re = emu.lmpc_solve_shared(A, B, d, wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    ok = ro["status"] == 0
    assert re["riccati_factor"] and (re["status"] == ro["status"]).all() and (re["iter"][ok] == ro["iter"][ok]).all()
    assert _rel(re["control"][ok], ro["control"][ok]) <= 1e-9 and _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= 1e-9
    at_minimiser = int(((ro["iter"][:, 0] == 1) & ok).sum())
    steps = int(((ro["iter"][:, 0] >= 2) & (ro["iter"][:, 0] <= 3) & (ro["iter"][:, 1] == 0) & ok).sum())
    # (round 5: this form of the pass, too, takes the first steps of the iteration itself where bounds on u_0 are the picks)
    assert 0 < at_minimiser < re["lane_pass_finished"] <= at_minimiser + steps
    monkeypatch.setitem(OPTIONS, "no_lane_spec", 1)
    rn = emu.lmpc_solve_shared(A, B, d, wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    assert rn["lane_pass_finished"] == at_minimiser and (rn["status"] == re["status"]).all() and (rn["iter"] == re["iter"]).all()
    assert _rel(rn["control"][ok], re["control"][ok]) <= 1e-11
    monkeypatch.setitem(OPTIONS, "no_lane_spec", 0)
    monkeypatch.setitem(OPTIONS, "no_lane_pass", 1)
    r0 = emu.lmpc_solve_shared(A, B, d, wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    assert r0["lane_pass_finished"] == -1 and (r0["status"] == re["status"]).all() and (r0["iter"] == re["iter"]).all()
    assert _rel(r0["control"][ok], re["control"][ok]) <= 1e-11


@pytest.mark.parametrize("case", ["falling_mass_32", "falling_mass_20", "com_12", "com_20_generic"])
def test_one_instance_per_lane_pass_filters_for_the_other_tiers(emu, oracle, monkeypatch, case, no_axis):
    """in front of a first tier that is not the Riccati-factor tier (shapes the library has no instantiation of it for: the run-time-shape
    and factor-only kernels) the pass only FILTERS: the instances at their unconstrained minimiser end in it, the tier solves the others
    from scratch.  (2, 1) and (6, 3) lanes, against the oracle."""
    from copra_amd import workloads
    monkeypatch.setenv("COPRA_EMU_LANE_FILTER", "1")
    if case == "com_12":  # (since round 4 the library holds the Riccati-factor tier for (6, 3) at every horizon: switched off to keep
        monkeypatch.setitem(OPTIONS, "no_ric", 1)  # covering the pass as a filter in front of the factor-only tier)
    b = 70
    wl = {"falling_mass_32": lambda: workloads.double_integrator(b, N=32), "falling_mass_20": lambda: workloads.double_integrator(b, N=20),
          "com_12": lambda: workloads.com_preview(b, N=12, seed=5), "com_20_generic": lambda: workloads.com_preview(b, N=20, seed=5)}[case]()
    args = (wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    re = emu.lmpc_solve(*args, specialised=(case != "com_20_generic"))
    ro = oracle.lmpc_solve_batch(*args, nthreads=8)
    ok = ro["status"] == 0
    assert (re["status"] == ro["status"]).all() and (re["iter"][ok] == ro["iter"][ok]).all()
    assert _rel(re["control"][ok], ro["control"][ok]) <= 1e-9
    assert re["lane_pass_finished"] == int(((ro["iter"][:, 0] == 1) & ok).sum()) > 0
    assert re["riccati_factor"] == (case == "com_20_generic")


@pytest.mark.parametrize("specialised", [True, False])
def test_selection_rows_of_a_trajectory_constraint(emu, oracle, monkeypatch, specialised):
    """|v| <= v_max written as TrajectoryConstraint(E = [S; -S], f) with S a selection matrix: the plan builder classifies such rows as
    +- one component of one state (the rows of +-Psi, like TrajectoryBoundConstraint's), the controller keeps the compact variant of the
    Riccati-factor tier; against the oracle, and identical to the dense-row classification (option no_selection_rows)"""
    from copra_amd import workloads
    b = 40
    wl = workloads.com_preview(b, v_max=0.5, u_max=2.5, seed=41)
    S3 = np.hstack([np.zeros((3, 3)), np.eye(3)])
    cstrs = [dict(kind="trajectory", E=np.vstack([S3, -S3]), f=[0.5, 0.5, 0.5, 0.12, 0.12, 0.12], ineq=True), wl["cstrs"][1]]
    args = (wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], cstrs)
    re = emu.lmpc_solve(*args, specialised=specialised)
    ro = oracle.lmpc_solve_batch(*args, nthreads=8)
    ok = ro["status"] == 0
    # (initial velocities lie in [-0.2, 0.2]: some instances start below the lower limit -- infeasible, status 1 --, in others it binds)
    assert ok.sum() >= 8 and (~ok).sum() >= 3 and (re["status"] == ro["status"]).all() and (re["iter"][ok] == ro["iter"][ok]).all()
    assert _rel(re["control"][ok], ro["control"][ok]) <= 1e-9 and _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= 1e-9
    assert (ro["iter"][ok, 0] > 1).any()
    monkeypatch.setitem(OPTIONS, "no_selection_rows", 1)
    rd = emu.lmpc_solve(*args, specialised=specialised)
    assert (rd["status"] == re["status"]).all() and (rd["iter"][ok] == re["iter"][ok]).all()
    assert _rel(rd["control"][ok], re["control"][ok]) <= 1e-10
    if specialised:
        assert re["lds_bytes"] < rd["lds_bytes"]  # (compact variant of the tier against its general one)


@pytest.mark.parametrize("kind", ["terminal_velocity", "terminal_box_and_first_control", "mixed_one_step"])
def test_full_size_rows_that_touch_one_step(emu, oracle, monkeypatch, kind):
    """a terminal constraint written the reference's way -- a FULL-SIZE E (rows over the whole trajectory) that is non-zero in the last
    state only -- is classified as a per-step row of that step (selection rows as +- one component); likewise full-size G / mixed rows
    inside one step.  Against the oracle and against the full-row classification (option no_step_rows)."""
    from copra_amd import workloads
    b = 24
    wl = workloads.com_preview(b, v_max=0.6, u_max=3.0, seed=51)
    N, nx, nu = wl["N"], 6, 3
    X, U = nx * (N + 1), nu * N
    rng = np.random.default_rng(5)
    if kind == "terminal_velocity":  # |v_N| <= 0.02: six selection rows in the last block
        E = np.zeros((6, X))
        E[:3, X - 3:] = np.eye(3)
        E[3:, X - 3:] = -np.eye(3)
        extra = [dict(kind="trajectory", E=E, f=[0.02] * 6, ineq=True)]
    elif kind == "terminal_box_and_first_control":  # a dense terminal row + a full-size control row inside step 0
        E = np.zeros((2, X))
        E[:, X - 6:] = rng.standard_normal((2, 6))
        G = np.zeros((1, U))
        G[0, :3] = [1.0, -0.5, 0.25]
        extra = [dict(kind="trajectory", E=E, f=[2.0, 2.5], ineq=True), dict(kind="control", G=G, f=[0.4], ineq=True)]
    else:  # mixed row inside step 3
        E = np.zeros((1, X))
        G = np.zeros((1, U))
        E[0, 3 * nx + 3: 3 * nx + 6] = [1.0, 1.0, 1.0]
        G[0, 3 * nu: 3 * nu + 3] = [0.1, 0.1, 0.1]
        extra = [dict(kind="mixed", E=E, G=G, f=[0.5], ineq=True)]
    cstrs = wl["cstrs"] + extra
    args = (wl["A"], wl["B"], wl["d"], wl["x0"], N, wl["costs"], cstrs)
    re = emu.lmpc_solve(*args)
    ro = oracle.lmpc_solve_batch(*args, nthreads=8)
    ok = ro["status"] == 0
    assert ok.sum() >= b // 2 and (re["status"] == ro["status"]).all() and (re["iter"][ok] == ro["iter"][ok]).all()
    assert _rel(re["control"][ok], ro["control"][ok]) <= 1e-9 and _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= 1e-9
    assert (ro["iter"][ok, 0] > 1).any()
    monkeypatch.setitem(OPTIONS, "no_step_rows", 1)
    rf = emu.lmpc_solve(*args)
    assert (rf["status"] == re["status"]).all() and (rf["iter"][ok] == re["iter"][ok]).all() and _rel(rf["control"][ok], re["control"][ok]) <= 1e-9
    if kind == "terminal_velocity":
        assert re["riccati_factor"] and re["lds_bytes"] < rf["lds_bytes"]  # (the compact variant of the headline's tier is kept)


@pytest.mark.parametrize("specialised", [True, False])
@pytest.mark.parametrize("what", ["state_reference", "state_and_control_reference"])
def test_reference_trajectory_costs(emu, oracle, monkeypatch, specialised, what):
    """a reference that changes along the horizon -- the reference's API can only express it as a FULL-SIZE entry, M = blkdiag(M0 .. M0),
    stacked p (costFunctions.cpp:63-82) -- is classified as a per-step entry with the reference of the step (CostTerm::pstride) and runs on
    the step-by-step cost phase instead of the dense contraction; against the oracle (which takes the full-size entry as it is) and
    against the dense path (option no_stage_refs), with controller-wide and with per-instance reference trajectories"""
    from copra_amd import workloads
    b = 12
    wl = workloads.com_preview(b, v_max=0.5, u_max=2.5, seed=71)
    N, nx, nu = wl["N"], 6, 3
    rng = np.random.default_rng(9)
    M0 = np.eye(6) if what == "state_reference" else np.eye(6)[:4] + 0.1 * rng.standard_normal((4, 6))
    r = M0.shape[0]
    Mf = np.kron(np.eye(N + 1), M0)
    ts = np.linspace(0.0, 1.0, N + 1)
    pos = workloads.COM_X_INIT[:3][None, :] + ts[:, None] * (workloads.COM_X_GOAL[:3] - workloads.COM_X_INIT[:3])[None, :]
    xref = np.hstack([pos, 0.05 * np.ones((N + 1, 3))])  # a straight-line reference with a constant velocity
    pf = (xref @ M0.T).reshape(-1)
    wf = np.tile(np.array([10.0, 10.0, 10.0, 1.0, 1.0, 1.0])[:r], N + 1)
    costs = [dict(kind="trajectory", M=Mf, p=pf, weights=wf)]
    if what == "state_and_control_reference":
        uref = 0.2 * np.sin(np.arange(N))[:, None] * np.ones((1, nu))
        costs.append(dict(kind="control", N=np.kron(np.eye(N), np.eye(nu)), p=uref.reshape(-1), weights=np.full(nu * N, 1e-2)))
    else:
        costs.append(wl["costs"][1])
    args = (wl["A"], wl["B"], wl["d"], wl["x0"], N, costs, wl["cstrs"])
    re = emu.lmpc_solve(*args, specialised=specialised)
    ro = oracle.lmpc_solve_batch(*args, nthreads=8)
    ok = ro["status"] == 0
    assert ok.sum() >= b - 2 and (re["status"] == ro["status"]).all() and (re["iter"][ok] == ro["iter"][ok]).all()
    assert _rel(re["control"][ok], ro["control"][ok]) <= 1e-8 and _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= 1e-8
    refs = {0: np.tile(pf, (b, 1)) + 0.02 * rng.standard_normal((b, pf.size))}  # every instance its own reference trajectory
    re2 = emu.lmpc_solve(*args, specialised=specialised, cost_refs=refs)
    monkeypatch.setitem(OPTIONS, "no_stage_refs", 1)
    rd = emu.lmpc_solve(*args, specialised=specialised)
    rd2 = emu.lmpc_solve(*args, specialised=specialised, cost_refs=refs)
    assert (rd["status"] == re["status"]).all() and _rel(rd["control"][ok], re["control"][ok]) <= 1e-8
    ok2 = rd2["status"] == 0
    assert (rd2["status"] == re2["status"]).all() and ok2.sum() >= b - 3 and _rel(rd2["control"][ok2], re2["control"][ok2]) <= 1e-8
    assert np.abs(re2["control"][ok2 & ok] - re["control"][ok2 & ok]).max() > 1e-4  # (the per-instance references did something)


@pytest.mark.parametrize("mode", ["lane_pass_and_handover", "own_sweep", "lane_pass_filter_only"])
def test_reference_trajectory_on_the_riccati_factor_tier(emu, oracle, monkeypatch, mode):
    """reference trajectories keep the headline's kernels: the affine term of the stage cost, h_k = -sum_t [M N]_t' W_t p_t[k], changes
    along the horizon -- rebuilt per stage by the one-instance-per-lane pass (lmpc_lane.hpp: stage_h), formed once per instance by the
    tier's own sweep (lmpc_fused_ric.hpp, in the place of record k) -- with controller-wide and per-instance references, a state reference
    alone and together with a control reference"""
    from copra_amd import workloads
    b = 70
    wl = workloads.com_preview(b, v_max=0.5, u_max=2.5, seed=71)
    N, nu = wl["N"], 3
    rng = np.random.default_rng(9)
    ts = np.linspace(0.0, 1.0, N + 1)
    pos = workloads.COM_X_INIT[:3][None, :] + ts[:, None] * (workloads.COM_X_GOAL[:3] - workloads.COM_X_INIT[:3])[None, :]
    xref = np.hstack([pos, 0.05 * np.ones((N + 1, 3))])
    pf = xref.reshape(-1)
    uref = 0.2 * np.sin(np.arange(N))[:, None] * np.ones((1, nu))
    track = dict(kind="trajectory", M=np.kron(np.eye(N + 1), np.eye(6)), p=pf, weights=np.tile([10.0, 10.0, 10.0, 1.0, 1.0, 1.0], N + 1))
    if mode == "own_sweep":
        monkeypatch.setitem(OPTIONS, "no_lane_pass", 1)
    if mode == "lane_pass_filter_only":
        monkeypatch.setitem(OPTIONS, "no_lane_handover", 1)
    for second in (wl["costs"][1], dict(kind="control", N=np.kron(np.eye(N), np.eye(nu)), p=uref.reshape(-1), weights=np.full(nu * N, 1e-2))):
        args = (wl["A"], wl["B"], wl["d"], wl["x0"], N, [track, second], wl["cstrs"])
        ro = oracle.lmpc_solve_batch(*args, nthreads=8)
        ok = ro["status"] == 0
        re = emu.lmpc_solve(*args)
        assert re["riccati_factor"] and (re["lane_pass_finished"] > 0) == (mode != "own_sweep")
        assert ok.sum() >= b - 4 and (re["status"] == ro["status"]).all() and (re["iter"][ok] == ro["iter"][ok]).all()
        assert _rel(re["control"][ok], ro["control"][ok]) <= 1e-9 and _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= 1e-9
        refs = {0: np.tile(pf, (b, 1)) + 0.02 * rng.standard_normal((b, pf.size))}  # every instance its own reference trajectory
        re2 = emu.lmpc_solve(*args, cost_refs=refs)
        monkeypatch.setitem(OPTIONS, "no_stage_refs", 1)
        rd2 = emu.lmpc_solve(*args, cost_refs=refs)  # (the full-size entry as it is: dense contraction)
        monkeypatch.setitem(OPTIONS, "no_stage_refs", 0)
        ok2 = rd2["status"] == 0
        assert re2["riccati_factor"] and not rd2["riccati_factor"]
        assert ok2.sum() >= b - 6 and (re2["status"] == rd2["status"]).all() and (re2["iter"][ok2] == rd2["iter"][ok2]).all()
        assert _rel(re2["control"][ok2], rd2["control"][ok2]) <= 1e-9
        assert np.abs(re2["control"][ok2 & ok] - re["control"][ok2 & ok]).max() > 1e-4


def test_reference_trajectory_shared_model(emu, oracle, monkeypatch):
    """one model, one reference trajectory, a batch of measured states: the prepare run of the shared-model mode does the sweep with the
    stage-varying affine term once, the lane pass and the tier read its records"""
    from copra_amd import workloads
    b = 70
    wl = workloads.com_preview(b, v_max=0.5, u_max=2.5, seed=23)
    A, B, d, N = wl["A"][3], wl["B"][3], wl["d"][3], wl["N"]
    ts = np.linspace(0.0, 1.0, N + 1)
    xref = workloads.COM_X_INIT[None, :] + ts[:, None] * (workloads.COM_X_GOAL - workloads.COM_X_INIT)[None, :]
    costs = [dict(kind="trajectory", M=np.kron(np.eye(N + 1), np.eye(6)), p=xref.reshape(-1), weights=np.tile([10.0, 10.0, 10.0, 1.0, 1.0, 1.0], N + 1)),
             wl["costs"][1]]
    ro = oracle.lmpc_solve_batch(np.tile(A, (b, 1, 1)), np.tile(B, (b, 1, 1)), np.tile(d, (b, 1)), wl["x0"], N, costs, wl["cstrs"], nthreads=8)
    ok = ro["status"] == 0
    for no_pass in (False, True):
        if no_pass:
            monkeypatch.setitem(OPTIONS, "no_lane_pass", 1)
        re = emu.lmpc_solve_shared(A, B, d, wl["x0"], N, costs, wl["cstrs"])
        assert re["riccati_factor"] and ok.sum() >= b - 4 and (re["status"] == ro["status"]).all() and (re["iter"][ok] == ro["iter"][ok]).all()
        assert _rel(re["control"][ok], ro["control"][ok]) <= 1e-9 and _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= 1e-9
        assert (re["lane_pass_finished"] > 0) == (not no_pass)


@pytest.mark.parametrize("mode", ["lane_pass_and_handover", "own_sweep", "factor_only_tier"])
def test_mixed_cost_reference_trajectory(emu, oracle, monkeypatch, mode):
    """MixedCost with a reference that changes along the horizon: the full-size entry of costFunctions.cpp:173-210 -- M with fullXDim
    columns (those of x_N zero), N with fullUDim, both repeating their block over the N steps that have a control -- is a per-step entry
    with the reference of the step too; on the Riccati-factor tier behind the lane pass, on its own sweep and on the factor-only tier's
    step-by-step cost phase, controller-wide and per-instance references; against the oracle, which takes the full-size entry as it is"""
    from copra_amd import workloads
    b = 70
    wl = workloads.com_preview(b, v_max=0.5, u_max=2.5, seed=33)
    N, nx, nu = wl["N"], 6, 3
    rng = np.random.default_rng(2)
    M0 = np.hstack([np.zeros((3, 3)), np.eye(3)])  # the velocity ...
    N0 = 0.05 * np.eye(3)  # ... plus a share of the control follows a reference
    Mf = np.hstack([np.kron(np.eye(N), M0), np.zeros((3 * N, nx))])
    Nf = np.kron(np.eye(N), N0)
    pk = 0.05 * np.sin(0.4 * np.arange(N))[:, None] * np.array([1.0, -0.5, 0.3])[None, :]
    mixed = dict(kind="mixed", M=Mf, N=Nf, p=pk.reshape(-1), weights=np.tile([2.0, 3.0, 1.5], N))
    costs = [wl["costs"][0], mixed, wl["costs"][1]]
    if mode == "own_sweep":
        monkeypatch.setitem(OPTIONS, "no_lane_pass", 1)
    if mode == "factor_only_tier":
        monkeypatch.setitem(OPTIONS, "no_ric", 1)
    args = (wl["A"], wl["B"], wl["d"], wl["x0"], N, costs, wl["cstrs"])
    ro = oracle.lmpc_solve_batch(*args, nthreads=8)
    ok = ro["status"] == 0
    re = emu.lmpc_solve(*args)
    assert re["riccati_factor"] == (mode != "factor_only_tier")
    assert ok.sum() >= b - 4 and (re["status"] == ro["status"]).all() and (re["iter"][ok] == ro["iter"][ok]).all()
    assert _rel(re["control"][ok], ro["control"][ok]) <= 1e-9 and _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= 1e-9
    refs = {1: np.tile(pk.reshape(-1), (b, 1)) + 0.02 * rng.standard_normal((b, pk.size))}
    re2 = emu.lmpc_solve(*args, cost_refs=refs)
    monkeypatch.setitem(OPTIONS, "no_stage_refs", 1)
    rd = emu.lmpc_solve(*args)
    rd2 = emu.lmpc_solve(*args, cost_refs=refs)  # (the full-size entry as it is: dense contraction)
    assert not rd["riccati_factor"] and (rd["status"] == re["status"]).all() and _rel(rd["control"][ok], re["control"][ok]) <= 1e-9
    ok2 = rd2["status"] == 0
    assert ok2.sum() >= b - 6 and (re2["status"] == rd2["status"]).all() and (re2["iter"][ok2] == rd2["iter"][ok2]).all()
    assert _rel(re2["control"][ok2], rd2["control"][ok2]) <= 1e-9
    assert np.abs(re2["control"][ok2 & ok] - re["control"][ok2 & ok]).max() > 1e-5


@pytest.mark.parametrize("name", sorted(__import__("published_qps").PUBLISHED))
def test_published_qp_examples_on_the_kernel_body(oracle, emu, name):
    """tests/published_qps.py (R solve.QP, Goldfarb & Idnani 1983, MathWorks, Nocedal & Wright, CVXOPT, Scilab, Hock-Schittkowski)
    through the dense-QP kernel body (qp_dense.hpp + gi_core.hpp) on the wave emulator: the published digits, and the oracle's
    iteration counts"""
    import published_qps as PQ
    qp = PQ.PUBLISHED[name]
    args = (qp["Q"], qp["c"], qp["Aeq"], qp["beq"], qp["Aineq"], qp["bineq"], qp["XL"], qp["XU"])
    x, fail, it = emu.qp_dense(*args)
    xo, fo, ito = oracle.quadprog_dense(*args)
    assert fail[0] == fo == 0 and tuple(it[0]) == tuple(ito)
    assert np.abs(x[0] - qp["x_star"]).max() <= qp["tol"] and np.abs(x[0] - xo).max() <= 1e-12 * (1.0 + np.abs(xo).max())
    if qp["iterations"] is not None:
        assert tuple(it[0]) == qp["iterations"]


def test_lane_pass_counts_the_rows_the_unconstrained_minimiser_violates(oracle, emu, no_axis):
    """FusedPlan::lane_hist: on the first solve of a controller the one-instance-per-lane pass histograms, over the instances it leaves
    to the first tier, how many rows and bounds the unconstrained minimiser violates -- copra_batch_solve picks the tier's starting
    layout from it (the final active set is ~ 1.1 x that count).  Against numpy: -Q^-1 c of the oracle's condensed QP, rows and bounds
    counted with qpgen2's test."""
    from copra_amd import workloads
    b = 96
    wl = workloads.com_preview(b, v_max=0.25, u_max=1.2)
    re = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    hist = emu.last_lane_hist()
    want = np.zeros(32, dtype=int)
    active = []
    for k in range(b):
        qp = oracle.lmpc_build(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], wl["costs"], wl["cstrs"])
        u0 = -np.linalg.solve(qp["Q"], qp["c"])
        nv = int((qp["Aineq"] @ u0 - qp["bineq"] > 1e-12).sum() + (u0 > qp["ub"] + 1e-12).sum() + (u0 < qp["lb"] - 1e-12).sum())
        if nv:
            want[min(nv, 31)] += 1
            active.append((nv, re["iter"][k, 0] - 1 - 2 * re["iter"][k, 1]))
    assert hist.sum() == want.sum() > b // 2 and (hist == want).all()
    nv, na = np.array(active).T
    assert np.corrcoef(nv, na)[0, 1] > 0.8 and 0.9 <= na.mean() / nv.mean() <= 1.35  # (what the predictor b + b / 8 rests on)


@pytest.mark.parametrize("shape", ["com_12", "com_5", "com_21", "com_18_tight", "planar_16", "planar_30", "fallingmass_48", "fallingmass_64"])
def test_riccati_factor_tier_with_a_run_time_horizon(emu, oracle, shape):
    """round-3 verdict, missing #5: the headline's two kernels existed for (6, 3) at N = 10, 15, 20 only; every other horizon ran 2.5 x
    slower unless the USER's box had hipcc.  The library now holds the Riccati-factor tier and the one-instance-per-lane pass with the
    horizon as a RUN-TIME value (NH == 0 builds, copra_hip_ric.hip) for the double integrators in one, two and three dimensions
    (plan_builder.hpp::ric_aot_shape): the plan takes the tier by itself, statuses and BOTH iteration counters equal the oracle's,
    controls to 1e-9 -- incl. the lane pass's hand-over, a tight workload that steps down to the LDS-Q1 layout, and the tier alone."""
    from copra_amd import workloads
    import test_gpu_parity as G
    b = 24 if shape.startswith("planar") else 70  # (the planar cases walk long active-set paths: 20 - 40 iterations per instance)
    wl = {"com_12": lambda: workloads.com_preview(b, N=12, seed=5), "com_5": lambda: workloads.com_preview(b, N=5, seed=6),
          "com_21": lambda: workloads.com_preview(b, N=21, seed=7),
          "com_18_tight": lambda: workloads.com_preview(b, N=18, seed=8, v_max=0.3, u_max=1.5),
          "planar_16": lambda: G._planar_integrator(b, 16), "planar_30": lambda: G._planar_integrator(b, 30, seed=3),
          "fallingmass_48": lambda: workloads.double_integrator(b, N=48), "fallingmass_64": lambda: workloads.double_integrator(b, N=64)}[shape]()
    args = (wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    ro = oracle.lmpc_solve_batch(*args, nthreads=8)
    ok = ro["status"] == 0
    for opts in ({}, {"no_lane_pass": 1}, {"no_lane_handover": 1}) if shape != "planar_30" else ({},):
        OPTIONS.update(opts)
        try:
            re = emu.lmpc_solve(*args)
        finally:
            for k in opts:
                OPTIONS.pop(k, None)
        assert re["riccati_factor"], shape
        assert (re["status"] == ro["status"]).all() and (re["iter"][ok] == ro["iter"][ok]).all(), (shape, opts)
        assert _rel(re["control"][ok], ro["control"][ok]) <= 1e-9 and _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= 1e-9


@pytest.mark.parametrize("b,vmax,umax", [(70, 0.6, 2.0), (33, 0.3, 1.5), (96, 0.25, 1.0)])
def test_lane_pass_takes_the_first_step_of_the_iteration(emu, oracle, monkeypatch, b, vmax, umax, no_axis):
    """Round 5: where a bound on u_0 is qpgen2's first pick, the one-instance-per-lane pass takes that step itself (closed form in the
    quantities of its roll-out: lmpc_lane.hpp) and finishes the instance when the new iterate violates nothing -- iterations (2, 0), as
    the oracle counts them.  With the speculation switched off the same instances go through the first tier: same statuses, same
    counters, same U and X; with it, strictly more instances end in the pass.  Ragged last wave included."""
    from copra_amd import workloads
    wl = workloads.com_preview(b, v_max=vmax, u_max=umax, seed=9)
    args = (wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    ro = oracle.lmpc_solve_batch(*args, nthreads=8)
    rs = emu.lmpc_solve(*args)
    monkeypatch.setitem(OPTIONS, "no_lane_spec", 1)
    rn = emu.lmpc_solve(*args)
    ok = ro["status"] == 0
    for r in (rs, rn):
        assert (r["status"] == ro["status"]).all() and (r["iter"][ok] == ro["iter"][ok]).all()
        assert _rel(r["control"][ok], ro["control"][ok]) <= 1e-9 and _rel(r["trajectory"][ok], ro["trajectory"][ok]) <= 1e-9
    at_minimiser = int(((ro["iter"][:, 0] == 1) & ok).sum())
    assert rn["lane_pass_finished"] == at_minimiser
    one_bound = int(((ro["iter"][:, 0] == 2) & (ro["iter"][:, 1] == 0) & ok).sum())
    two_bounds = int(((ro["iter"][:, 0] >= 3) & (ro["iter"][:, 0] <= 4) & (ro["iter"][:, 1] == 0) & ok).sum())  # (two, on decoupled axes three, bounds on u_0)
    assert at_minimiser <= rs["lane_pass_finished"] <= at_minimiser + one_bound + two_bounds
    if vmax >= 0.6:  # (loose velocity rows: every first pick is a bound on u_0 -- every one-constraint instance ends in the pass, and
        #  the two-constraint ones whose second pick is another bound on u_0)
        assert rs["lane_pass_finished"] >= at_minimiser + one_bound and one_bound > 0


@pytest.mark.parametrize("first", [0, 12, 24, 36, 48, 60, 72, 84])
def test_random_controllers_against_the_oracle(emu, oracle, first):
    """tests/random_controllers.py: random shapes (nx 1..7, nu 1..3, N 2..24), random per-instance systems and a random mix of the four
    cost classes and five constraint classes as per-step and as full-size entries -- the kernel bodies against the oracle: statuses equal,
    U and X within 1e-6 entry-wise; the iteration counters equal except where a tie is broken at rounding level (rare: counted)."""
    import random_controllers as RC
    ndiff = ninst = 0
    for seed in range(first, first + 12):
        c = RC.make(seed, batch=4)
        ro = oracle.lmpc_solve_batch(c["A"], c["B"], c["d"], c["x0"], c["N"], c["costs"], c["cstrs"])
        re = emu.lmpc_solve(c["A"], c["B"], c["d"], c["x0"], c["N"], c["costs"], c["cstrs"])
        what = "seed %d (%d, %d, %d) %s" % (seed, c["nx"], c["nu"], c["N"], c["forms"])
        assert (re["status"] == ro["status"]).all(), what
        ok = ro["status"] == 0
        if ok.any():
            assert _rel(re["control"][ok], ro["control"][ok]) <= RTOL and _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= RTOL, what
        ndiff += int((re["iter"][ok] != ro["iter"][ok]).any(axis=1).sum())  # (status 0: an infeasible exit is reached through multipliers at
        ninst += int(ok.sum())  #  rounding level, where the drop counters of two arithmetics differ)
    assert ndiff <= ninst // 16


@pytest.mark.parametrize("first", [0, 6, 12])
def test_random_integrator_controllers_on_the_headline_kernels(emu, oracle, first):
    """tests/random_controllers.py::make_integrator in the emulator, 70 instances each (the one-instance-per-lane pass runs in front of the
    Riccati-factor tier from 64 instances on; general rows take the tier's own sweep): statuses, both iteration counters, U and X."""
    import random_controllers as RC
    for seed in range(first, first + 6):
        c = RC.make_integrator(seed, 70)
        ro = oracle.lmpc_solve_batch(c["A"], c["B"], c["d"], c["x0"], c["N"], c["costs"], c["cstrs"], nthreads=4)
        re = emu.lmpc_solve(c["A"], c["B"], c["d"], c["x0"], c["N"], c["costs"], c["cstrs"])
        what = "seed %d (%d, %d, %d) %s" % (seed, c["nx"], c["nu"], c["N"], c["forms"])
        ok = ro["status"] == 0
        assert (re["status"] == ro["status"]).all() and (re["iter"][ok] == ro["iter"][ok]).all(), what
        assert _rel(re["control"][ok], ro["control"][ok]) <= RTOL and _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= RTOL, what


@pytest.mark.parametrize("shape", [(12, 6), (5, 3), (4, 2)])
def test_random_controllers_on_the_interior_point_kernels(emu, oracle, shape):
    """beyond 64 variables: the stage-wise interior-point kernels (lmpc_riccati_mfma.hpp for (12, 6), lmpc_riccati.hpp otherwise) on random
    controllers -- an instance the kernel accepts (status 0; the others go to the Goldfarb-Idnani kernel on the device) is within 1e-6 of the
    certified optimum where it is further than that from the oracle"""
    import random_controllers as RC
    import truth
    nx, nu = shape
    naccepted = 0
    for seed in range(8):
        N = 64 // nu + 1 + seed
        c = RC.make(seed, batch=2, shape=(nx, nu, N))
        ro = oracle.lmpc_solve_batch(c["A"], c["B"], c["d"], c["x0"], N, c["costs"], c["cstrs"])
        re = emu.lmpc_solve_riccati(c["A"], c["B"], c["d"], c["x0"], N, c["costs"], c["cstrs"])
        if re is None:  # (not stage-wise -- a cost row that couples the steps: the controller runs on the Goldfarb-Idnani kernels)
            continue
        for k in range(2):
            if re["status"][k] != 0:
                continue
            assert ro["status"][k] == 0, (shape, seed, k)
            naccepted += 1
            if _rel(re["control"][k], ro["control"][k]) <= RTOL and _rel(re["trajectory"][k], ro["trajectory"][k]) <= RTOL:
                continue
            t = truth.solve(c["A"][k], c["B"][k], c["d"][k], c["x0"][k], N, c["costs"], c["cstrs"], ro["control"][k])
            assert _rel(re["control"][k], t["control"]) <= RTOL and _rel(re["trajectory"][k], t["trajectory"]) <= RTOL, (shape, seed, k, c["forms"])
    assert naccepted >= 2  # (the interior-point iteration breaks down on many RANDOM controllers: those go to the other kernel)


def test_shared_model_riccati_factor_tier_with_general_rows(emu, oracle):
    """the Riccati-factor tier in shared-model mode on controllers with GENERAL rows (dense state rows, mixed rows, control rows): the rows
    that go through the free response of the preview need it rebuilt from each instance's x0 (the stage records come from one prepare run
    at x0 = 0) -- round 4 had found statuses and U off there and taken general rows out of the mode.  Against the oracle, incl. the
    iteration counters"""
    import random_controllers as RC
    b = 12
    for seed in (1, 3, 4, 5):  # (1, 3, 4: the seeds that were wrong)
        wl, cstrs = RC.com_preview_with_general_rows(seed, b)
        A, B, d = wl["A"][3], wl["B"][3], wl["d"][3]
        ro = oracle.lmpc_solve_batch(np.tile(A, (b, 1, 1)), np.tile(B, (b, 1, 1)), np.tile(d, (b, 1)), wl["x0"], wl["N"], wl["costs"], cstrs, nthreads=8)
        re = emu.lmpc_solve_shared(A, B, d, wl["x0"], wl["N"], wl["costs"], cstrs)
        ok = ro["status"] == 0
        assert re["riccati_factor"] and (re["status"] == ro["status"]).all() and (re["iter"][ok] == ro["iter"][ok]).all(), seed
        assert _rel(re["control"][ok], ro["control"][ok]) <= 1e-7 and _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= 1e-7, seed


def test_shared_model_riccati_factor_tier_run_time_horizons(emu, oracle):
    """the Riccati-factor tier's shared-model mode on its RUN-TIME-horizon builds (random_controllers.py: make_integrator -- (6, 3) and (4, 2)
    at random horizons, reference trajectories, bound / row / mixed constraints): until round 4 only the compile-time horizons 10, 15, 20 of
    (6, 3) took the mode.  Against the oracle incl. the iteration counters; at least half of the controllers must have run the tier"""
    import random_controllers as RC
    b, nric = 10, 0
    for seed in range(0, 24):
        c = RC.make_integrator(seed, b)
        A, B, d = c["A"][0], c["B"][0], c["d"][0]
        ro = oracle.lmpc_solve_batch(np.tile(A, (b, 1, 1)), np.tile(B, (b, 1, 1)), np.tile(d, (b, 1)), c["x0"], c["N"], c["costs"], c["cstrs"], nthreads=8)
        re = emu.lmpc_solve_shared(A, B, d, c["x0"], c["N"], c["costs"], c["cstrs"])
        ok = ro["status"] == 0
        nric += int(bool(re["riccati_factor"]))
        assert (re["status"] == ro["status"]).all() and (re["iter"][ok] == ro["iter"][ok]).all(), seed
        if ok.any():
            assert _rel(re["control"][ok], ro["control"][ok]) <= 1e-7 and _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= 1e-7, seed
    assert nric >= 12


def test_random_dense_qps_with_awkward_cases(emu, oracle):
    """the dense-QP kernel body (qp_dense.hpp: plug-in point 1) on random problems up to 64 variables with the awkward cases mixed in
    (tests/fuzz/fuzz_dense_qp.py: contradictions, zero rows, infinite bounds, indefinite Hessians, pinned variables, duplicated rows):
    statuses, iteration counters and x against the oracle"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "fuzz"))
    import fuzz_dense_qp as FQ
    bad, tot, seen = FQ.run(0, 30, emu=True, verbose=True)
    assert bad == 0 and tot == 480 and len(seen) >= 7


@pytest.mark.parametrize("case", ["goals", "trajectories", "goals_N13"])
def test_shared_model_records_tier_with_per_instance_references(emu, oracle, case):
    """one model for the batch, every instance its own goal / reference trajectory: the shared lane pass adds the DELTA of the instance's
    feed-forward terms to the batch-wide records (lmpc_lane_shared_body: dkv_k = -Lam_k^-1 (dh_u + B' dpv+), dpv_k = dh_x + K_k' dh_u +
    Acl_k' dpv+) and hands U and X to the tier.  Against the oracle instance by instance incl. the iteration counters; compile-time
    horizon 20 and run-time horizon 13"""
    from copra_amd import workloads
    b = 20
    rng = np.random.default_rng(3)
    N = 13 if case.endswith("13") else 20
    wl = workloads.com_preview(b, N=N, seed=5)
    A, B, d = wl["A"][3], wl["B"][3], wl["d"][3]
    if case.startswith("goals"):
        costs = wl["costs"]
        refs = workloads.COM_X_GOAL[None, :] + 0.08 * rng.standard_normal((b, 6))
    else:
        ts = np.linspace(0, 1, N + 1)
        xref = workloads.COM_X_INIT[None, :] + ts[:, None] * (workloads.COM_X_GOAL - workloads.COM_X_INIT)[None, :]
        costs = [dict(kind="trajectory", M=np.kron(np.eye(N + 1), np.eye(6)), p=xref.reshape(-1), weights=np.tile([10.0, 10, 10, 1, 1, 1], N + 1)), wl["costs"][1]]
        refs = np.tile(xref.reshape(-1), (b, 1)) + 0.05 * rng.standard_normal((b, 6 * (N + 1)))
    re = emu.lmpc_solve_shared(A, B, d, wl["x0"], N, costs, wl["cstrs"], cost_refs={0: refs})
    assert re["riccati_factor"] and 0 < re["lane_pass_finished"] < b
    for k in range(b):
        ro = oracle.lmpc_solve(A, B, d, wl["x0"][k], N, [dict(costs[0], p=refs[k]), costs[1]], wl["cstrs"])
        assert re["status"][k] == ro["status"] == 0 and tuple(re["iter"][k]) == tuple(ro["iter"])
        assert _rel(re["control"][k], ro["control"]) <= 1e-7 and _rel(re["trajectory"][k], ro["trajectory"]) <= 1e-7


# ---- the one-(instance, axis)-per-lane solver (lmpc_axis.hpp; round 6) ----

def _axis_case(emu, oracle, wl, cstrs=None, what=""):
    cs = wl["cstrs"] if cstrs is None else cstrs
    re = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], cs)
    ro = oracle.lmpc_solve_batch(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], cs, nthreads=4)
    ok = ro["status"] == 0
    assert (re["status"] == ro["status"]).all(), what
    assert (re["iter"][ok] == ro["iter"][ok]).all(), what
    assert _rel(re["control"][ok], ro["control"][ok]) <= 1e-8 and _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= 1e-8, what
    return re, ro


@pytest.mark.parametrize("N,vmax,umax", [(20, 0.6, 3.0), (20, 0.25, 1.2), (15, 0.35, 1.8), (7, 0.3, 1.5), (21, 0.4, 2.0)])
def test_axis_solver_on_the_com_preview(emu, oracle, N, vmax, umax):
    """BASELINE configs[2]'s model is three decoupled double integrators: every (instance, axis) is solved by ONE lane -- sweep, roll-out and
    the Goldfarb-Idnani iteration in range-space form on the Riccati factor (lmpc_axis.hpp).  qpgen2's run on the whole problem is an
    interleaving of the axes' runs: statuses, BOTH iteration counters (the sums over the axes), U and X equal the oracle's.  N = 20: the build
    with the horizon compiled in; 7, 15: the run-time horizon of the same build.  Batches that are not a
    multiple of the 21 instances of a wave."""
    from copra_amd import workloads
    b = 50 if N != 20 else 85
    wl = workloads.com_preview(b, N=N, v_max=vmax, u_max=umax, seed=5 + N)
    re, ro = _axis_case(emu, oracle, wl, what=(N, vmax))
    # with six active constraints per axis nearly everything ends in it (the others: the first tier, from scratch)
    assert re["lane_pass_finished"] >= (b - 2 if vmax >= 0.35 else b // 2)
    assert ro["iter"][:, 0].max() >= 3


def test_axis_solver_hands_over_what_outgrows_its_lanes(emu, oracle, monkeypatch):
    """an axis whose active set outgrows the lane's room (here: a build with room for TWO constraints) sends its INSTANCE to a list.  The SECOND
    CHANCE walks that list: the same solver with room for twelve (S, its factor and the small vectors in the lane's LDS, loops with run-time
    trip counts), instances taken from the list -- it finishes them all here; without it (and for whatever it lists in turn) the first tier
    solves them from scratch -- same results, same counters either way"""
    from copra_amd import workloads
    monkeypatch.setenv("COPRA_EMU_AXIS_QMAX2", "1")
    wl = workloads.com_preview(64, v_max=0.3, u_max=1.5, seed=9)
    re, ro = _axis_case(emu, oracle, wl)
    assert re["lane_pass_finished"] == 64 and ro["iter"][:, 0].max() >= 8
    monkeypatch.setenv("COPRA_EMU_AXIS_NO_SECOND_CHANCE", "1")
    re, ro = _axis_case(emu, oracle, wl)
    assert 0 < re["lane_pass_finished"] < 64


def test_axis_solver_leaves_coupled_and_infeasible_instances_to_the_tier(emu, oracle):
    """the SYSTEMS are checked per instance: one whose A or B couples two axes goes to the tier (its neighbours in the wave do not); so does an
    instance whose x0 violates a state row (qpgen2: no step in primal space, "no solution": status 1 from the tier) and one with an empty box"""
    from copra_amd import workloads
    b = 45
    wl = workloads.com_preview(b, v_max=0.5, u_max=2.5, seed=3)
    A2, B2, x2 = wl["A"].copy(), wl["B"].copy(), wl["x0"].copy()
    A2[5, 0, 4] = 0.03  # (x position picks up y velocity)
    B2[30, 3, 1] = 0.02  # (x velocity driven by the y control)
    x2[12, 4] = 0.9  # (violates the velocity bound at step 0)
    wl2 = dict(wl, A=A2, B=B2, x0=x2)
    re, ro = _axis_case(emu, oracle, wl2)
    assert ro["status"][12] == 1 and re["lane_pass_finished"] <= b - 3
    same = np.ones(b, dtype=bool)
    same[[5, 12, 30]] = False
    base = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    assert (re["control"][same] == base["control"][same]).all()  # (nothing of a neighbour's result depends on them)


def test_axis_solver_with_per_instance_goals(emu, oracle):
    """copra_batch_set_cost_reference in front of the (instance, axis)-per-lane solver: every instance tracks its own goal (one
    TrajectoryCost(M, p_b) per LMPC in the reference, costFunctions.cpp:63-82) -- a lane rebuilds the affine terms of its axis from its
    instance's reference (FusedPlan::axis_cref).  Statuses, both iteration counters, U and X against the oracle run instance by instance with
    its own cost; the instances end in the solver, not in the tier"""
    from copra_amd import workloads
    rng = np.random.default_rng(21)
    for N, vmax, umax, b in ((20, 0.5, 2.5, 44), (12, 0.35, 1.8, 30)):
        wl = workloads.com_preview(b, N=N, v_max=vmax, u_max=umax, seed=11 + N)
        goals = wl["costs"][0]["p"][None, :] + 0.3 * rng.standard_normal((b, 6))
        re = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"], cost_refs={0: goals})
        assert re["lane_pass_finished"] >= b - 2
        for k in range(b):
            costs = [dict(wl["costs"][0], p=goals[k]), wl["costs"][1]]
            ro = oracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], costs, wl["cstrs"])
            assert re["status"][k] == ro["status"] == 0 and tuple(re["iter"][k]) == tuple(ro["iter"]), (N, k)
            assert _rel(re["control"][k], ro["control"]) <= 1e-8 and _rel(re["trajectory"][k], ro["trajectory"]) <= 1e-8, (N, k)


@pytest.mark.parametrize("nu,N,amax", [(3, 20, None), (3, 14, 2.0), (2, 20, None), (2, 9, 1.5)])
def test_axis_solver_on_chains_of_three_states(emu, oracle, nu, N, amax):
    """the jerk-controlled CoM model (position, velocity, acceleration per axis; the jerk as control: nx = 3 nu) on the (instance, axis)-per-lane
    solver's builds for chains of three states: tables in registers (one row per axis and step) and read from LDS (a bound on the acceleration
    as well: two rows).  Statuses, both counters, U and X against the oracle; the instances end in the solver"""
    from copra_amd import workloads
    b = 50
    wl = workloads.jerk_preview(b, nu=nu, N=N, seed=3 + N, v_max=0.3, j_max=6.0, a_max=amax)
    re, ro = _axis_case(emu, oracle, wl, what=(nu, N, amax))
    assert re["lane_pass_finished"] >= b - 3
    assert ro["iter"][:, 0].max() >= 3 and (ro["iter"][:, 0] >= 2).mean() >= 0.2  # (the constraints matter)


@pytest.mark.parametrize("axis", [True, False])
def test_reference_trajectory_on_chains_of_three_states(emu, oracle, monkeypatch, axis):
    """tracking on the jerk-controlled model: a full-size TrajectoryCost with NINE rows per step, classified as a per-step cost with the step's
    reference (plan_builder.hpp) -- on the (instance, axis)-per-lane solver, and (option no_axis_solver) on the general one-wave kernel, which
    evaluates the step's reference in its condense code; controller-wide and one per instance"""
    from copra_amd import workloads
    if not axis:
        monkeypatch.setitem(OPTIONS, "no_axis_solver", 1)
    rng = np.random.default_rng(41)
    b, N, nu = 24, 12, 3
    wl = workloads.jerk_preview(b, nu=nu, N=N, seed=8, v_max=0.35, j_max=8.0)
    nx = 3 * nu
    ts = np.linspace(0.0, 1.0, N + 1)
    xref = np.zeros((N + 1, nx))
    xref[:, :nu] = workloads.COM_X_INIT[:nu][None, :] + ts[:, None] * (workloads.COM_X_GOAL[:nu] - workloads.COM_X_INIT[:nu])[None, :]
    xref[:, nu:2 * nu] = 0.05
    track = dict(kind="trajectory", M=np.kron(np.eye(N + 1), np.eye(nx)), p=xref.reshape(-1), weights=np.tile(wl["costs"][0]["weights"], N + 1))
    args = (wl["A"], wl["B"], wl["d"], wl["x0"], N, [track, wl["costs"][1]], wl["cstrs"])
    ro = oracle.lmpc_solve_batch(*args, nthreads=4)
    ok = ro["status"] == 0
    re = emu.lmpc_solve(*args)
    assert ok.all() and (re["status"] == ro["status"]).all() and (re["iter"] == ro["iter"]).all()
    assert _rel(re["control"], ro["control"]) <= 1e-7 and _rel(re["trajectory"], ro["trajectory"]) <= 1e-7
    assert (re["lane_pass_finished"] >= b - 2) == axis
    own = np.tile(xref.reshape(-1), (b, 1)) + 0.03 * rng.standard_normal((b, xref.size))
    re2 = emu.lmpc_solve(*args, cost_refs={0: own})
    for k in range(0, b, 2):
        rk = oracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], N, [dict(track, p=own[k]), wl["costs"][1]], wl["cstrs"])
        assert re2["status"][k] == rk["status"] == 0 and tuple(re2["iter"][k]) == tuple(rk["iter"]) and _rel(re2["control"][k], rk["control"]) <= 1e-7


@pytest.mark.parametrize("model", ["com", "jerk"])
def test_axis_solver_with_states_in_axis_major_order(emu, oracle, model):
    """the same controllers with x = (p_x, v_x, p_y, v_y, ..) instead of (p, v): the engine sees the order of the states from the zero pattern
    of the first system it is given (plan_builder.hpp: axis_order_of) and the (instance, axis)-per-lane solver reads its lanes' axes through that
    order's index map and tables -- statuses, counters, U and X against the oracle run on the permuted controller; a goal per instance as well"""
    from copra_amd import workloads
    b = 45
    base = workloads.com_preview(b, v_max=0.4, u_max=2.0, seed=17) if model == "com" else workloads.jerk_preview(b, nu=3, N=16, seed=5, v_max=0.3, j_max=6.0)
    wl = workloads.axis_major(base)
    re, ro = _axis_case(emu, oracle, wl, what=model)
    assert re["lane_pass_finished"] >= b - 3
    nx = wl["A"].shape[1]
    goals = np.tile(wl["costs"][0]["p"], (b, 1)) + 0.1 * np.random.default_rng(2).standard_normal((b, nx)) * (np.asarray(wl["costs"][0]["p"]) != 0)
    re2 = emu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"], cost_refs={0: goals})
    assert re2["lane_pass_finished"] >= b - 3
    for k in range(0, b, 4):
        rk = oracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], [dict(wl["costs"][0], p=goals[k]), wl["costs"][1]], wl["cstrs"])
        assert re2["status"][k] == rk["status"] == 0 and tuple(re2["iter"][k]) == tuple(rk["iter"]) and _rel(re2["control"][k], rk["control"]) <= 1e-8


@pytest.mark.parametrize("nu,N", [(3, 20), (3, 13), (2, 20), (2, 29)])
def test_axis_solver_on_one_state_per_control(emu, oracle, nu, N):
    """a velocity-controlled point (nx = nu: the kinematic model of mobile-robot MPC) on the (instance, axis)-per-lane solver's builds for ONE
    state per control: statuses, both counters, U and X against the oracle; the instances end in the solver"""
    from copra_amd import workloads
    b = 50
    wl = workloads.kinematic_preview(b, nu=nu, N=N, seed=2 + N)
    re, ro = _axis_case(emu, oracle, wl, what=(nu, N))
    assert re["lane_pass_finished"] >= b - 3
    assert ro["iter"][:, 0].max() >= 3 and (ro["iter"][:, 0] >= 2).mean() >= 0.3  # (the constraints matter)


def test_axis_solver_with_per_instance_limits(emu, oracle):
    """every robot its own velocity and actuator limits (copra_batch_set_constraint_rhs, copra_batch_set_control_bounds) in front of the
    (instance, axis)-per-lane solver: the builds that keep bounds and right-hand sides in registers take the lane's own values where they are the
    same at every step of the horizon; an instance whose limits CHANGE along the horizon goes to the tier (same results).  And a controller
    without control bounds at all: an infinite box is not an empty one"""
    from copra_amd import workloads
    rng = np.random.default_rng(33)
    b, inf = 47, np.inf
    wl = workloads.com_preview(b, v_max=0.5, u_max=2.5, seed=13)
    N = wl["N"]
    vlim = 0.5 * rng.uniform(0.6, 1.3, b)
    ulim = 2.5 * rng.uniform(0.6, 1.3, b)
    args = (wl["A"], wl["B"], wl["d"], wl["x0"], N, wl["costs"], wl["cstrs"])
    rhs = np.repeat(vlim[:, None], 3 * (N + 1), axis=1)
    lo, hi = -np.repeat(ulim[:, None], 3 * N, axis=1), np.repeat(ulim[:, None], 3 * N, axis=1)
    hi[5, 3 * 7 + 1] *= 0.5  # (instance 5: a tighter bound on one control of step 7 -- its limits change along the horizon)
    rhs[9, 3 * 4 + 2] *= 0.8  # (instance 9: one row of step 4)
    re = emu.lmpc_solve(*args, row_rhs=rhs, bounds=(lo, hi))
    assert b - 6 <= re["lane_pass_finished"] <= b - 2
    for k in range(b):
        up = np.full((N + 1, 6), inf)
        up[:, 3:] = rhs[k].reshape(N + 1, 3)
        cs = [dict(kind="trajectory_bound", lower=np.full(6 * (N + 1), -inf), upper=up.reshape(-1)),
              dict(kind="control_bound", lower=lo[k], upper=hi[k])]
        ro = oracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], N, wl["costs"], cs)
        assert re["status"][k] == ro["status"], k
        if ro["status"] == 0:
            assert tuple(re["iter"][k]) == tuple(ro["iter"]) and _rel(re["control"][k], ro["control"]) <= 1e-8, k
    re2, ro2 = _axis_case(emu, oracle, wl, cstrs=[wl["cstrs"][0]])  # no control bounds at all
    assert re2["lane_pass_finished"] >= b - 2


def test_axis_solver_with_reference_trajectories(emu, oracle):
    """a TrajectoryCost as a full-size entry whose reference changes along the horizon (the only form the reference's API has for tracking,
    costFunctions.cpp:63-82 with AutoSpan) in front of the (instance, axis)-per-lane solver: the run-time-horizon builds rebuild h stage by
    stage from the step's reference -- controller-wide, per instance, and together with a control reference over N steps.  Statuses, both
    counters, U and X against the oracle; the instances end in the solver"""
    from copra_amd import workloads
    rng = np.random.default_rng(19)
    for N, b in ((20, 43), (11, 30)):
        wl = workloads.com_preview(b, N=N, v_max=0.5, u_max=2.5, seed=40 + N)
        ts = np.linspace(0.0, 1.0, N + 1)
        pos = workloads.COM_X_INIT[:3][None, :] + ts[:, None] * (workloads.COM_X_GOAL[:3] - workloads.COM_X_INIT[:3])[None, :]
        pf = np.hstack([pos, 0.05 * np.ones((N + 1, 3))]).reshape(-1)
        uref = 0.2 * np.sin(np.arange(N))[:, None] * np.ones((1, 3))
        track = dict(kind="trajectory", M=np.kron(np.eye(N + 1), np.eye(6)), p=pf, weights=np.tile([10.0, 10.0, 10.0, 1.0, 1.0, 1.0], N + 1))
        for second in (wl["costs"][1], dict(kind="control", N=np.kron(np.eye(N), np.eye(3)), p=uref.reshape(-1), weights=np.full(3 * N, 1e-2))):
            args = (wl["A"], wl["B"], wl["d"], wl["x0"], N, [track, second], wl["cstrs"])
            ro = oracle.lmpc_solve_batch(*args, nthreads=4)
            ok = ro["status"] == 0
            re = emu.lmpc_solve(*args)
            assert re["lane_pass_finished"] >= b - 2 and ok.sum() >= b - 2
            assert (re["status"] == ro["status"]).all() and (re["iter"][ok] == ro["iter"][ok]).all()
            assert _rel(re["control"][ok], ro["control"][ok]) <= 1e-8 and _rel(re["trajectory"][ok], ro["trajectory"][ok]) <= 1e-8
            own = np.tile(pf, (b, 1)) + 0.05 * rng.standard_normal((b, pf.size))  # every instance its own reference trajectory
            re2 = emu.lmpc_solve(*args, cost_refs={0: own})
            assert re2["lane_pass_finished"] >= b - 2
            for k in range(0, b, 3):
                rk = oracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], N, [dict(track, p=own[k]), second], wl["cstrs"])
                assert re2["status"][k] == rk["status"]
                if rk["status"] == 0:
                    assert tuple(re2["iter"][k]) == tuple(rk["iter"]) and _rel(re2["control"][k], rk["control"]) <= 1e-8


def test_axis_solver_with_rows_that_change_along_the_horizon(emu, oracle):
    """tables that are NOT the same at every step (FusedPlan::axis_const = 0: the builds that read them from LDS stage by stage): a mixed
    constraint v_k + 0.1 u_k <= v_max per axis (rows with a control part, none at step N), lower velocity limits as rows (two rows per axis and
    step), and control bounds that tighten along the horizon (a full-size ControlBoundConstraint)"""
    from copra_amd import workloads
    b, N = 40, 12
    wl = workloads.com_preview(b, N=N, v_max=0.35, u_max=1.8, seed=21)
    vsel = np.hstack([np.zeros((3, 3)), np.eye(3)])
    up = np.repeat(np.linspace(2.0, 1.2, N), 3)
    cs = [dict(kind="mixed", E=vsel, G=0.1 * np.eye(3), f=[0.35] * 3, ineq=True),
          dict(kind="trajectory", E=-vsel, f=[0.5] * 3, ineq=True),
          dict(kind="control_bound", lower=-up, upper=up)]
    re, ro = _axis_case(emu, oracle, wl, cstrs=cs)
    assert re["lane_pass_finished"] >= b // 2 and ro["iter"][:, 0].max() >= 3


@pytest.mark.parametrize("N", [14, 27])
def test_axis_solver_on_the_planar_point_mass(emu, oracle, N):
    """two dimensions: 32 instances per wave; N = 27: the build for up to 31 steps.  With a gravity-like bias d and a terminal cost."""
    rng = np.random.default_rng(N)
    dim = 2
    b, nx, nu = 70, 2 * dim, dim
    T = rng.uniform(0.08, 0.15, b)
    A = np.zeros((b, nx, nx))
    B = np.zeros((b, nx, nu))
    I = np.eye(dim)
    A[:, :dim, :dim] = I
    A[:, dim:, dim:] = I
    A[:, :dim, dim:] = T[:, None, None] * I
    B[:, :dim, :] = (0.5 * T * T)[:, None, None] * I
    B[:, dim:, :] = T[:, None, None] * I
    d = np.tile(0.01 * rng.standard_normal(nx), (b, 1))
    x0 = np.hstack([0.3 * rng.standard_normal((b, dim)), rng.uniform(-0.25, 0.25, (b, dim))])
    goal = np.concatenate([rng.uniform(0.5, 1.0, dim), np.zeros(dim)])
    costs = [dict(kind="trajectory", M=np.eye(nx), p=goal, weights=[10.0] * dim + [1.0] * dim),
             dict(kind="control", N=np.eye(nu), p=np.zeros(nu), weights=[1e-2] * nu),
             dict(kind="target", M=np.eye(nx), p=goal, weights=[50.0] * nx)]
    inf = np.inf
    cstrs = [dict(kind="trajectory_bound", lower=[-inf] * nx, upper=[inf] * dim + [0.9] * dim),
             dict(kind="control_bound", lower=[-6.0] * nu, upper=[6.0] * nu)]
    wl = dict(A=A, B=B, d=d, x0=x0, N=N, costs=costs, cstrs=cstrs)
    re, ro = _axis_case(emu, oracle, wl)
    assert re["lane_pass_finished"] >= b // 4 and ro["iter"][:, 0].max() >= 3


def test_axis_solver_is_a_choice_of_kernels_not_of_results(emu, oracle, monkeypatch):
    """copra_options_t::no_axis_solver: the same controller on the one-instance-per-lane pass + first tier -- the same statuses and counters,
    results equal to rounding"""
    from copra_amd import workloads
    wl = workloads.com_preview(64, v_max=0.4, u_max=2.0, seed=17)
    args = (wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
    ra = emu.lmpc_solve(*args)
    monkeypatch.setitem(OPTIONS, "no_axis_solver", 1)
    rb = emu.lmpc_solve(*args)
    assert (ra["status"] == rb["status"]).all() and (ra["iter"] == rb["iter"]).all()
    assert _rel(ra["control"], rb["control"]) <= 1e-9 and _rel(ra["trajectory"], rb["trajectory"]) <= 1e-9
