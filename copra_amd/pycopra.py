"""pycopra -- the reference's Python surface (binding/python/CopraBindings.cpp, module `pyCopra`) over the HIP engine.

    import copra_amd.pycopra as copra          # instead of:  import pyCopra as copra
    ps = copra.PreviewSystem(); ps.system(A, B, c, x0, nb_steps)
    controller = copra.LMPC(ps)
    controller.add_cost(copra.TargetCost(M, -xd)); controller.add_constraint(copra.ControlConstraint(G, h))
    controller.solve(); controller.control(); controller.trajectory()

Same names, argument order and behaviour as the boost.python binding (which is stale upstream: it no longer builds):
snake_case members, numpy in / numpy out, C++ std::domain_error / std::runtime_error surfacing as RuntimeError
(boost.python's default translation; binding/python/tests/pyTests.py:311-339), and the reference's ownership rule
that a cost or constraint the caller has dropped is removed from the controller after the next solve
(src/LMPC.cpp:288-307, pyTests.py:233-277).  One controller = a batch of one instance on the GPU; use
copra_amd.BatchLMPC for batches.  Every solve() is a fresh controller's first solve (reference quirk Q2 is not
reproduced).
"""
import ctypes as C
import sys
import time

import numpy as np

from . import _capi
from .autospan import autospan_cost, autospan_cstr, span_matrix, span_vector


def _guard(fn):
    """C++ exceptions reach Python as RuntimeError through boost.python"""
    def wrapped(*a, **k):
        try:
            return fn(*a, **k)
        except (_capi.CopraDomainError, _capi.CopraRuntimeError) as e:
            raise RuntimeError(str(e)) from None
    return wrapped


class SolverFlag:
    # include/solverUtils.h:34-50 -> copra_batch_select_solver
    DEFAULT = 0  # what it is in the reference (src/solverUtils.cpp:9-34: DEFAULT -> QuadProgDense): Goldfarb-Idnani at every size
    QuadProgDense = 1  # always the hand-written Goldfarb-Idnani kernels (the reference's QuadProgDense arithmetic)
    HipRiccati = 2  # (no reference counterpart) the engine picks: the stage-wise Riccati interior-point kernel for long stage-wise horizons,
    #                 Goldfarb-Idnani elsewhere; iteration counts are Newton steps there

    @staticmethod
    def engine_solver(flag):
        """name of the engine's solver (BatchLMPC.select_solver) behind a flag"""
        return "default" if flag == SolverFlag.HipRiccati else "quadprog_dense"


class AutoSpan:
    """include/AutoSpan.h (static helpers); CopraBindings.cpp:123-127"""
    span_matrix = staticmethod(_guard(lambda mat, new_dim, add_cols=0: span_matrix(mat, new_dim, add_cols)))
    span_vector = staticmethod(_guard(lambda vec, new_dim: span_vector(vec, new_dim)))


class PreviewSystem:
    """include/PreviewSystem.h:22-68"""

    def __init__(self, A=None, B=None, c=None, x_init=None, number_of_steps=None):
        self.is_updated = False
        self.nr_u_step = self.nr_x_step = self.x_dim = self.u_dim = self.full_x_dim = self.full_u_dim = 0
        if A is not None:
            self.system(A, B, c, x_init, number_of_steps)

    @_guard
    def system(self, A, B, c, x_init, number_of_steps):  # src/PreviewSystem.cpp:16-55
        A, B = np.atleast_2d(np.asarray(A, dtype=np.float64)), np.atleast_2d(np.asarray(B, dtype=np.float64))
        c, x0 = np.atleast_1d(np.asarray(c, dtype=np.float64)), np.atleast_1d(np.asarray(x_init, dtype=np.float64))
        if number_of_steps <= 0:
            raise _capi.CopraDomainError("The number of step sould be a positive number! ")
        if A.shape[0] != A.shape[1]:
            raise _capi.CopraDomainError("A should be a square matrix")
        if B.shape[0] != A.shape[0]:
            raise _capi.CopraDomainError("A and B should have the same number of rows")
        if c.shape[0] != A.shape[0] or x0.shape[0] != A.shape[0]:
            raise _capi.CopraDomainError("A, the bias and the initial state should have the same number of rows")
        self.A, self.B, self.d, self.x0 = A.copy(), B.copy(), c.copy(), x0.copy()
        self.nr_u_step, self.nr_x_step = int(number_of_steps), int(number_of_steps) + 1
        self.x_dim, self.u_dim = A.shape[0], B.shape[1]
        self.full_x_dim, self.full_u_dim = self.x_dim * self.nr_x_step, self.u_dim * self.nr_u_step
        self.is_updated = False

    def x_init(self, x):  # PreviewSystem::xInit (PreviewSystem.h:52): receding horizon, nothing is rebuilt
        self.x0 = np.atleast_1d(np.asarray(x, dtype=np.float64)).copy()

    def update_system(self):  # done on the device at every solve
        self.is_updated = True


class _Piece:
    def _dict(self):
        raise NotImplementedError


class CostFunction(_Piece):
    kind = None

    def __init__(self, M, N, p):
        self._M = None if M is None else np.atleast_2d(np.asarray(M, dtype=np.float64))
        self._N = None if N is None else np.atleast_2d(np.asarray(N, dtype=np.float64))
        self._p = np.atleast_1d(np.asarray(p, dtype=np.float64))
        self._w = np.ones(self._p.shape[0])  # costFunctions.h:117

    @_guard
    def weights(self, w):  # costFunctions.h:54-67
        w = np.atleast_1d(np.asarray(w, dtype=np.float64))
        rows = self._p.shape[0]
        if w.shape[0] == rows:
            self._w = w.copy()
        elif w.shape[0] > 0 and rows % w.shape[0] == 0:
            self._w = np.tile(w, rows // w.shape[0])
        else:
            raise _capi.CopraDomainError("weights: the vector size must divide the number of rows of the cost")

    def weight(self, w):  # costFunctions.h:72-76
        self._w = np.full(self._p.shape[0], float(w))

    def auto_span(self):
        d = autospan_cost(self._dict())
        self._M, self._N = d.get("M"), d.get("N")
        self._p, self._w = np.atleast_1d(d["p"]), np.atleast_1d(d["weights"])

    def name(self):
        return type(self).__name__

    def _dict(self):
        return dict(kind=self.kind, M=self._M, N=self._N, p=self._p, weights=self._w)


class TrajectoryCost(CostFunction):
    kind = "trajectory"

    def __init__(self, M, p):
        super().__init__(M, None, p)


class TargetCost(CostFunction):
    kind = "target"

    def __init__(self, M, p):
        super().__init__(M, None, p)

    def auto_span(self):
        pass


class ControlCost(CostFunction):
    kind = "control"

    def __init__(self, N, p):
        super().__init__(None, N, p)


class MixedCost(CostFunction):
    kind = "mixed"

    def __init__(self, M, N, p):
        super().__init__(M, N, p)


class Constraint(_Piece):
    kind = None

    def name(self):
        return type(self).__name__

    def auto_span(self):
        d = autospan_cstr(self._dict())
        for k, v in d.items():
            if k not in ("kind", "ineq"):
                setattr(self, "_" + k, v)


class EqIneqConstraint(Constraint):
    def __init__(self, E, G, f, is_inequality_constraint=True):
        self._E = None if E is None else np.atleast_2d(np.asarray(E, dtype=np.float64))
        self._G = None if G is None else np.atleast_2d(np.asarray(G, dtype=np.float64))
        self._f = np.atleast_1d(np.asarray(f, dtype=np.float64))
        self._ineq = bool(is_inequality_constraint)

    def _dict(self):
        return dict(kind=self.kind, E=self._E, G=self._G, f=self._f, ineq=self._ineq)


class TrajectoryConstraint(EqIneqConstraint):
    kind = "trajectory"

    def __init__(self, E, f, is_inequality_constraint=True):
        super().__init__(E, None, f, is_inequality_constraint)


class ControlConstraint(EqIneqConstraint):
    kind = "control"

    def __init__(self, G, f, is_inequality_constraint=True):
        super().__init__(None, G, f, is_inequality_constraint)


class MixedConstraint(EqIneqConstraint):
    kind = "mixed"

    def __init__(self, E, G, f, is_inequality_constraint=True):
        super().__init__(E, G, f, is_inequality_constraint)


class _BoundConstraint(Constraint):
    def __init__(self, lower, upper):
        self._lower = np.atleast_1d(np.asarray(lower, dtype=np.float64))
        self._upper = np.atleast_1d(np.asarray(upper, dtype=np.float64))

    def lower(self):
        return self._lower

    def upper(self):
        return self._upper

    def _dict(self):
        return dict(kind=self.kind, lower=self._lower, upper=self._upper)


class TrajectoryBoundConstraint(_BoundConstraint):
    kind = "trajectory_bound"


class ControlBoundConstraint(_BoundConstraint):
    kind = "control_bound"


class LMPC:
    """include/LMPC.h:36-191; binding names of CopraBindings.cpp:275-286"""
    _initial_state = False

    def __init__(self, ps_or_flag=None, flag=SolverFlag.DEFAULT):
        self._ps = ps_or_flag if isinstance(ps_or_flag, PreviewSystem) else None
        self._costs, self._cstrs = [], []
        self._eng = None
        self._flag = ps_or_flag if (ps_or_flag is not None and not isinstance(ps_or_flag, PreviewSystem)) else flag
        self._dirty = True
        self._costs_dirty = False
        self._built_costs = None
        self.handle_builds = 0  # (not in the reference: how often the device-side controller was built)
        self._control = np.zeros(0)
        self._trajectory = np.zeros(0)
        self._solve_time = self._solve_and_build_time = 0.0
        self._fail = 0
        self._ref_accumulation = False
        self._acc = {}  # id(cost) -> [Q, c, E, f] accumulated over the solves (reference_accumulation)

    def reference_accumulation(self, on=True):
        """Opt-in compatibility with reference quirk Q2 (no counterpart in the reference's API: it is how its cost classes behave).  The
        per-step entries of TrajectoryCost and MixedCost ADD each solve's Q, E, f to members that are only zeroed in initializeCost
        (src/costFunctions.cpp:52-55, 73-80 / 184-187, 205-213), and MixedCost also adds to c: the k-th solve() on one controller sees
        k x the trajectory Hessian.  Off (default): every solve is a fresh controller's first one, the batched engine's semantics.
        On: each solve evaluates such a cost on the device (the condense code behind copra_batch_dump_qp), accumulates it here in the
        reference's own order and hands the sums to the fused solve as a dense cost (COPRA_COST_DENSE, the plug-in route)."""
        self._ref_accumulation = bool(on)
        self._dirty = True

    def _accumulating(self, c):
        per_step = c._M is None or c._M.shape[1] == self._ps.x_dim
        return self._ref_accumulation and c.kind in ("trajectory", "mixed") and per_step and hasattr(c, "_M")

    def _accumulated_dict(self, c):
        """one more update() of cost c as the reference performs it: Q_ += .., E_ += .., f_ += .., c_ = (mixed: +=) E_' x0 + f_"""
        from .batch import BatchLMPC
        ps = self._ps
        nx, n = ps.x_dim, ps.u_dim * ps.nr_u_step
        e1 = BatchLMPC(nx, ps.u_dim, ps.nr_u_step, 1, [c._dict()], [])
        e1.set_system(ps.A[None], ps.B[None], ps.d[None], ps.x0[None])
        Q1 = e1.dump_qp(0)["Q"] - 1e-6 * np.eye(n)  # (minus LMPC::updateSystem's 1e-6 I, src/LMPC.cpp:228-230)
        e1.close()
        E1, f1 = np.zeros((nx, n)), np.zeros(n)
        try:  # E and f from the InitialStateLMPC form of the same cost (src/InitialStateLMPC.cpp:80-84)
            e2 = BatchLMPC(nx, ps.u_dim, ps.nr_u_step, 1, [c._dict()], [], initial_state=dict(R=np.eye(nx), r=np.zeros(nx)))
            e2.set_system(ps.A[None], ps.B[None], ps.d[None], ps.x0[None])
            qp = e2.dump_qp(0)
            E1, f1 = qp["Q"][:nx, nx:], qp["c"][nx:]
            e2.close()
        except _capi.CopraUnsupported:
            pass
        acc = self._acc.setdefault(id(c), [np.zeros((n, n)), np.zeros(n), np.zeros((nx, n)), np.zeros(n)])
        acc[0] += Q1
        acc[2] += E1
        acc[3] += f1
        cnow = acc[2].T @ ps.x0 + acc[3]
        acc[1] = acc[1] + cnow if c.kind == "mixed" else cnow
        return dict(kind="dense", Q=acc[0], c=acc[1], E=acc[2], f=acc[3])

    def select_qp_Solver(self, flag):
        pass

    def initialize_controller(self, ps):
        self._ps = ps
        self._dirty = True

    # ---- LMPC::addCost / addConstraint: initializeCost / initializeConstraint checks happen here ----
    def _check(self, costs, cstrs):
        keep = []
        cc = _capi.pack_costs([c._dict() for c in costs], keep)
        kk = _capi.pack_cstrs([c._dict() for c in cstrs], keep)
        dims = _capi.Dims(self._ps.x_dim, self._ps.u_dim, self._ps.nr_u_step, 1)
        rc = _capi.lib().copra_plan_check(C.byref(dims), len(costs), cc, len(cstrs), kk, None)
        if rc in (_capi.COPRA_ERR_DOMAIN, _capi.COPRA_ERR_RUNTIME):
            _capi.check(rc)

    @_guard
    def add_cost(self, cost):
        self._check([cost], [])
        self._costs.append(cost)
        self._costs_dirty = True  # (_engine() looks at what has changed in the list)

    @_guard
    def add_constraint(self, constr):
        self._check([], [constr])
        if isinstance(constr, ControlBoundConstraint) and any(isinstance(c, ControlBoundConstraint) for c in self._cstrs):
            pass  # (the reference silently keeps writing into lb/ub: only one fits, LMPC.cpp:274-279)
        self._cstrs.append(constr)
        self._dirty = True

    def remove_cost(self, cost):
        self._costs = [c for c in self._costs if c is not cost]
        self._costs_dirty = True

    def remove_constraint(self, constr):
        self._cstrs = [c for c in self._cstrs if c is not constr]
        self._dirty = True

    def clear_costs(self):
        self._costs, self._costs_dirty = [], True

    def clear_constraints(self):
        self._cstrs, self._dirty = [], True

    reset_constraints = clear_constraints  # (the binding's stale name, CopraBindings.cpp:286)

    # ---- LMPC::solve (src/LMPC.cpp:79-101) ----
    def _costs_against_engine(self):
        """The reference evaluates every cost anew in every solve (LMPC.cpp:233-247): weights changed on a cost that is inside the
        controller, or a cost replaced by a new one, take effect at the next solve.  0: the costs are what the engine was built from;
        1: they are up to the references p -- the only way the reference's API has to move a reference trajectory is a NEW cost object
        (M, N, p are constructor arguments) --, which go to the engine that exists (copra_batch_set_cost_reference); 2: anything else"""
        built = self._built_costs
        if built is None:  # (costs that are not the built-in classes: the list alone decides, as before)
            return 2 if self._costs_dirty else 0
        if len(built) != len(self._costs):
            return 2
        same = lambda a, b: (a is None and b is None) or (a is not None and b is not None and a.shape == b.shape and np.array_equal(a, b))
        for c, (kind, M, N, w, _) in zip(self._costs, built):
            if not all(hasattr(c, a) for a in ("_M", "_N", "_p", "_w")):  # (not one of the built-in classes)
                return 2
            if c.kind != kind or not same(c._w, w) or not same(c._M, M) or not same(c._N, N):
                return 2
        rc = 0
        for t, c in enumerate(self._costs):
            if not same(c._p, built[t][4]):
                # (a controller past the one-wave kernels would leave its fast kernels in per-instance-reference mode -- the LDS-resident
                #  interior-point kernel above all, include/copra_hip.h --: there a new engine is worth more than the engine build)
                if self._eng.lanes_per_instance() > 64 or self._eng.solver() == "riccati_ipm":
                    return 2
                try:
                    self._eng.set_cost_reference(t, c._p[None])
                except Exception:
                    return 2  # (a kernel that cannot: a new engine can)
                built[t] = built[t][:4] + (c._p.copy(),)
                rc = 1
        return rc

    def _engine(self):
        from .batch import BatchLMPC
        if self._ref_accumulation and any(self._accumulating(c) for c in self._costs):
            # (the accumulated members are part of the plan: a new engine for every solve, as for user-defined costs)
            if self._eng is not None:
                self._eng.close()
            live = {id(c) for c in self._costs}
            self._acc = {k: v for k, v in self._acc.items() if k in live}
            cd = [self._accumulated_dict(c) if self._accumulating(c) else c._dict() for c in self._costs]
            ist = self._initial_state_desc() if self._initial_state else None
            self._eng = BatchLMPC(self._ps.x_dim, self._ps.u_dim, self._ps.nr_u_step, 1, cd, [c._dict() for c in self._cstrs],
                                  initial_state=ist)
            self._eng.select_solver(SolverFlag.engine_solver(self._flag))
            self._dirty, self._costs_dirty, self._built_costs = True, False, None
            self.handle_builds += 1
            return self._eng
        if self._eng is not None and not self._dirty and self._costs_against_engine() < 2:
            self._costs_dirty = False
            return self._eng
        if self._eng is not None:
            self._eng.close()
        ist = self._initial_state_desc() if self._initial_state else None
        self._eng = BatchLMPC(self._ps.x_dim, self._ps.u_dim, self._ps.nr_u_step, 1, [c._dict() for c in self._costs],
                              [c._dict() for c in self._cstrs], initial_state=ist)
        self._eng.select_solver(SolverFlag.engine_solver(self._flag))
        self._dirty = self._costs_dirty = False
        self.handle_builds += 1
        cp = lambda a: None if a is None else np.array(a, dtype=np.float64, copy=True)
        builtin = all(hasattr(c, a) for c in self._costs for a in ("_M", "_N", "_p", "_w"))
        self._built_costs = [(c.kind, cp(c._M), cp(c._N), cp(c._w), cp(c._p)) for c in self._costs] if builtin else None
        return self._eng

    def select_qp_solver(self, flag):
        """LMPC::selectQPSolver (src/LMPC.cpp:62-65)"""
        self._flag = flag
        if self._eng is not None:
            self._eng.select_solver(SolverFlag.engine_solver(flag))

    @_guard
    def solve(self):
        t0 = time.perf_counter()
        ps = self._ps
        eng = self._engine()
        eng.set_system(ps.A[None], ps.B[None], ps.d[None], ps.x0[None])
        self._before_solve(eng)
        ps.is_updated = True
        eng.solve()
        res = eng.results()
        self._fail = int(res["status"][0])
        self._solve_time = eng.last_solve_seconds()
        if self._fail == 0:  # outputs only on success (LMPC.cpp:95-97)
            self._control = res["control"][0].copy()
            self._trajectory = res["trajectory"][0].copy()
            self._after_solve(eng)
        self._check_delete()
        self._solve_and_build_time = time.perf_counter() - t0
        return self._fail == 0

    def _before_solve(self, eng):
        pass

    def _after_solve(self, eng):
        pass

    def _check_delete(self):
        """LMPC::checkDeleteCostsAndConstraints (src/LMPC.cpp:288-307): pieces only the controller still refers to are
        dropped AFTER the solve.  sys.getrefcount sees the list slot and its own argument: 2 = nobody else."""
        for lst in (self._costs, self._cstrs):
            for i in range(len(lst) - 1, -1, -1):
                if sys.getrefcount(lst[i]) <= 2:
                    del lst[i]
                    self._dirty = True

    def inform(self):
        print({0: "No problems", 1: "The minimization problem has no solution"}.get(
            self._fail, "Problems with the decomposition of Q (Is it symmetric?)"))

    def solver_kind(self):
        """(not in the reference) the algorithm the next solve() runs on the device: 'quadprog_dense' -- Goldfarb-Idnani, what every flag of
        the reference means -- or 'riccati_ipm', which only SolverFlag.HipRiccati can select"""
        return self._engine().solver()

    def solve_time(self):
        return self._solve_time  # device time of the launch (SI_solve only, LMPC.cpp:88-91)

    def solve_and_build_time(self):
        return self._solve_and_build_time

    def control(self):
        return self._control

    def trajectory(self):
        return self._trajectory


class InitialStateLMPC(LMPC):
    """include/InitialStateLMPC.h:18-42"""
    _initial_state = True

    def __init__(self, ps_or_flag=None, flag=SolverFlag.DEFAULT):
        super().__init__(ps_or_flag, flag)
        nx = self._ps.x_dim if self._ps is not None else 0
        self._R, self._r = np.zeros((nx, nx)), np.zeros(nx)  # InitialStateLMPC.cpp:20-28
        # x0lb_ = x0ub_ = ps->x0 captured ONCE, at construction (InitialStateLMPC.cpp:20-28): a later x_init() does not
        # move the default bounds
        self._x0lb = self._x0ub = (np.array(self._ps.x0, dtype=np.float64) if self._ps is not None else None)
        self._x0_opt = np.zeros(0)

    def _initial_state_desc(self):
        return dict(R=self._R, r=self._r)

    def reset_initial_state_cost(self, R, r):
        self._R, self._r = np.atleast_2d(np.asarray(R, dtype=np.float64)), np.atleast_1d(np.asarray(r, dtype=np.float64))
        self._dirty = True

    def reset_initial_state_bounds(self, lower, upper):
        self._x0lb = np.atleast_1d(np.asarray(lower, dtype=np.float64))
        self._x0ub = np.atleast_1d(np.asarray(upper, dtype=np.float64))

    def initial_state(self):
        return self._x0_opt

    def _before_solve(self, eng):
        lo = self._ps.x0 if self._x0lb is None else self._x0lb
        up = self._ps.x0 if self._x0ub is None else self._x0ub
        eng.set_initial_state_bounds(lo[None], up[None])

    def _after_solve(self, eng):
        self._x0_opt = eng.initial_state()[0].copy()


def pythonSolverFactory(flag=SolverFlag.DEFAULT):
    """CopraBindings.cpp:100: a QP back-end object (plug-in point 1) -- here the batched dense-QP entry point"""
    from .batch import qp_solve_dense_batch
    return qp_solve_dense_batch
