"""Edge cases of the reference path that its own tests never exercise (SURVEY.md 7 "hard parts", 8a quirks):
shared by tests/test_oracle.py (oracle vs an independent formulation), tests/test_emu_kernels.py (kernel bodies on
the CPU emulator) and tests/test_gpu_parity.py (HIP through the C ABI)."""
import numpy as np

import fixtures as F

INF = np.inf


def finite_lower_trajectory_bound(N=12):
    """Quirk Q1 (src/constraints.cpp:289-296): a FINITE lower trajectory bound is stacked as  x <= lower  (same
    orientation as the upper rows, :303-310): all lower rows first (step-major), then all upper rows.  The reference
    never tests it (systems.h:81 uses -inf; TestLMPC_InitialState.cpp:68-69 comments the finite variant out).
    Returns the problem and the SAME constraint written with two explicit TrajectoryConstraint objects."""
    pb = F.bounded_system("trajectory", N=N)
    pb["costs"] = F._costs("trajectory", [0.0, 0.5])  # wants v -> +0.5; the 'lower' row keeps v <= -4 (v0 = -5: active)
    lower, upper = [1.0, -4.0], [INF, 0.0]
    quirk = [dict(kind="trajectory_bound", lower=lower, upper=upper), pb["cstrs"][1]]
    explicit = [dict(kind="trajectory", E=np.eye(2), f=lower),  # x <= lower, both components finite
                dict(kind="trajectory", E=[[0.0, 1.0]], f=[0.0]),  # v <= 0 (the only finite upper component)
                pb["cstrs"][1]]
    return pb, quirk, explicit


def duplicate_and_opposite_rows(N=12):
    """Duplicate rows (the same control limit three times: two rows of a ControlConstraint + the bound): the solver must
    neither cycle nor report a singular working set.  And an opposite pair (u <= 150 and -u <= -150, i.e. u = 150 written
    as two inequalities): once one of the pair is active the other is linearly dependent with a slack of +-1 ulp -- the
    Goldfarb-Idnani code of qpgen2 ends with "no solution" (status 1) on it although u = 150 is feasible; what is checked
    is that every implementation reports the SAME status.  Returns (problem with duplicates, problem with the pair)."""
    pb = F.bounded_system("target", N=N)
    dup = dict(pb)
    dup["cstrs"] = [dict(kind="control", G=[[1.0], [1.0]], f=[200.0, 200.0]),
                    dict(kind="control_bound", lower=[-INF], upper=[200.0])]
    opp = dict(pb)
    opp["cstrs"] = [dict(kind="control", G=[[1.0], [-1.0]], f=[150.0, -150.0])]
    return dup, opp


def opposite_state_rows_infeasible(N=12):
    """v <= 0 and -v <= 0 on EVERY step including step 0 (Q5: src/constraints.cpp:52,76) with v0 = -5: the rows of step
    0 read  0 . U <= -5  -> SI_fail() == 1"""
    pb = F.bounded_system("target", N=N)
    pb["cstrs"] = [dict(kind="trajectory", E=[[0.0, 1.0], [0.0, -1.0]], f=[0.0, 0.0])]
    return pb


R_QUADPROG_EXAMPLE = dict(
    # R package quadprog, help page of solve.QP (the published example of the qpgen2 code eigen-quadprog wraps):
    #   Dmat = I3, dvec = (0,5,0), Amat = matrix(c(-4,-3,0, 2,1,0, 0,-2,1), 3, 3), bvec = (-8,2,0),  t(Amat) x >= bvec
    #   $solution 0.4761905 1.0476190 2.0952381   $value -2.380952   $iterations 3 0   $Lagrangian 0 0.2380952 2.0952381
    # here in SolverInterface form (include/SolverInterface.h:54-80):  Aineq x <= bineq  with  Aineq = -t(Amat)
    Q=np.eye(3), c=-np.array([0.0, 5.0, 0.0]),
    Aineq=-np.array([[-4.0, -3.0, 0.0], [2.0, 1.0, 0.0], [0.0, -2.0, 1.0]]), bineq=-np.array([-8.0, 2.0, 0.0]),
    XL=np.full(3, -np.finfo(float).max), XU=np.full(3, np.finfo(float).max),
    x_star=np.array([10.0 / 21.0, 22.0 / 21.0, 44.0 / 21.0]), f_star=-50.0 / 21.0, iterations=(3, 0))
