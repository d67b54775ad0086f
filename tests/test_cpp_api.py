"""The C++ API mirror (copra_amd/cpp/include/copra/copra.h) compiled with g++ against libcopra_hip.so and driven by
tests written like the reference's doctest cases (tests/cpp/test_api.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_api")


def _build():
    from copra_amd import _capi
    _capi.build_library()
    src = os.path.join(ROOT, "tests", "cpp", "test_api.cpp")
    hdr = os.path.join(ROOT, "copra_amd", "cpp", "include", "copra", "copra.h")
    newest = max(os.path.getmtime(p) for p in (src, hdr, _capi.LIB_PATH))
    if os.path.exists(EXE) and os.path.getmtime(EXE) >= newest:
        return
    libdir = os.path.dirname(_capi.LIB_PATH)
    import pyoracle  # the user SolverInterface of the plug-in test runs the oracle's QuadProgDense restatement
    oracle_so = pyoracle.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "copra_amd", "cpp", "include"),
                           src, "-o", EXE, "-L", libdir, "-lcopra_hip", oracle_so, "-Wl,-rpath," + libdir,
                           "-Wl,-rpath," + os.path.dirname(oracle_so), "-Wl,-rpath,/opt/rocm/lib"])


DROPIN = os.path.join(ROOT, "tests", "cpp", "test_dropin")


def _build_dropin():
    """tests/cpp/test_dropin.cpp: ONLY the reference's header names (LMPC.h, PreviewSystem.h, QuadProgSolver.h, constraints.h,
    costFunctions.h, <Eigen/Core>, ...) and class names (copra::QuadProgDenseSolver)"""
    from copra_amd import _capi
    _capi.build_library()
    src = os.path.join(ROOT, "tests", "cpp", "test_dropin.cpp")
    inc = os.path.join(ROOT, "copra_amd", "cpp", "include")
    hdrs = [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")] + [os.path.join(inc, "copra", "copra.h")]
    newest = max(os.path.getmtime(p) for p in [src, _capi.LIB_PATH] + hdrs)
    if os.path.exists(DROPIN) and os.path.getmtime(DROPIN) >= newest:
        return
    libdir = os.path.dirname(_capi.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", inc, "-I", os.path.join(inc, "copra", "eigen_shim"),
                           src, "-o", DROPIN, "-L", libdir, "-lcopra_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])


def test_reference_header_names_and_class_names_compile_unchanged():
    """round-3 verdict, missing #4: code written against copra includes "LMPC.h", "PreviewSystem.h", "QuadProgSolver.h",
    "constraints.h", "costFunctions.h" (tests/TestLMPC.cpp:5-9) and instantiates copra::QuadProgDenseSolver
    (tests/TestSolvers.cpp:27).  The TU compiles with -Wall -Werror against the forwarding headers and its host-only mode runs
    (controller assembly, AutoSpan, debugUtils.h's exception macro, typedefs.h's trait) without a GPU."""
    _build_dropin()
    r = subprocess.run([DROPIN], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    inc = os.path.join(ROOT, "copra_amd", "cpp", "include")
    ref_names = {"AutoSpan.h", "InitialStateLMPC.h", "LMPC.h", "PreviewSystem.h", "QuadProgSolver.h", "SolverInterface.h", "api.h",
                 "constraints.h", "costFunctions.h", "debugUtils.h", "solverUtils.h", "typedefs.h"}  # include/ of the reference, minus
    assert ref_names <= set(os.listdir(inc))                                     # the optional proprietary solvers


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["solvers", "lmpc"])
def test_reference_style_cases_through_the_reference_names(mode):
    """TestSolvers.cpp:25-33 (QuadProgTest, + the Scilab known answer) and the first case of TestLMPC.cpp (300 steps, both bound
    constraints, solver by flag and by useSolver(QuadProgDenseSolver)) on the device, through the reference's names only"""
    _build_dropin()
    r = subprocess.run([DROPIN, mode], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_error_handlers_like_TestLMPC():
    """TestLMPC.cpp:949-1087: std::domain_error / std::runtime_error from system/addCost/addConstraint/weights"""
    _build()
    r = subprocess.run([EXE, "errors"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("steps", [12, 300])
def test_solve_cases_like_TestLMPC(steps):
    """TestLMPC.cpp:36-771: the twelve {cost} x {constraint} cases with the reference's acceptance checks, through the
    C++ mirror of copra's classes; 12 steps = one-wave kernel, 300 steps (the reference's nbStep) = workgroup kernel"""
    _build()
    r = subprocess.run([EXE, "solve", str(steps)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_initial_state_cases_like_TestLMPC_InitialState():
    """TestLMPC_InitialState.cpp: INITIAL-STATE-OPTIMIZATION and LMPC_AND_INITIAL-STATE-LMPC_COMPARISON, per-step and
    full-size entries, through copra::InitialStateLMPC of the C++ mirror"""
    _build()
    r = subprocess.run([EXE, "initial_state"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("steps", [12, 150])
def test_plugin_surface_user_subclasses_and_user_solver(steps):
    """north_star "constraint/cost plugin surface as a drop-in": a user-defined EqIneqConstraint subclass and a
    user-defined CostFunction subclass (their update() runs on the host against ps.Phi / Psi / xi, the results join the
    fused device solve as COPRA_CSTR_DENSE / COPRA_COST_DENSE), a user SolverInterface installed with LMPC::useSolver
    (device-condensed QP -> the CPU QuadProgDense restatement), LMPC::checkDeleteCostsAndConstraints, and the
    Q() c() E() f() / A() b() Y() z() accessors of the built-in classes; LMPC and InitialStateLMPC, 12 steps (one-wave
    kernels) and 150 steps (workgroup kernel)"""
    _build()
    r = subprocess.run([EXE, "plugins", str(steps)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_tracking_tick_reuses_the_device_controller():
    """a tracking controller through the reference's API: every tick REPLACES the TrajectoryCost by a new one with the moved reference
    trajectory (the reference has no setter for p, costFunctions.h:103-131).  The mirror sees that the cost list differs from the handle's
    in p alone and sends p (copra_batch_set_cost_reference): 300 ticks on ONE handle, each equal to a controller built from scratch
    with the same costs; a weights() call on a cost inside the controller is seen by the next solve (LMPC.cpp:233-247 evaluates every
    cost anew)"""
    _build()
    r = subprocess.run([EXE, "tracking", "300"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_reference_accumulation_across_solves_is_available_on_request():
    """reference quirk Q2 (src/costFunctions.cpp:73-80, 205-213: per-step TrajectoryCost / MixedCost accumulate Q, E, f -- MixedCost also
    c -- across the solves of one controller) as an opt-in of the mirror, copra::LMPC::referenceAccumulation(true): the k-th solve equals
    a fresh controller with k x the TrajectoryCost weights / the dense QP with k Q1 and sum_j j c1(x0_j); switched off, every solve is a
    fresh controller's first one (the batched engine's semantics)"""
    _build()
    r = subprocess.run([EXE, "accumulation"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
