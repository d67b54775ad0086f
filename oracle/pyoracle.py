"""ctypes front-end of the CPU ORACLE (oracle/copra_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under copra_amd/ may import this module (the product path must fail loudly without the HIP library).

Problems are described with plain dicts so that the same description can be handed to the oracle and to
the C-ABI of the HIP engine:

    cost  = {"kind": "trajectory"|"target"|"control"|"mixed", "M": ..., "N": ..., "p": ..., "weights": ...}
    cstr  = {"kind": "trajectory"|"control"|"mixed"|"trajectory_bound"|"control_bound",
             "E": ..., "G": ..., "f": ..., "lower": ..., "upper": ..., "ineq": True}
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

COST_KINDS = {"trajectory": 0, "target": 1, "control": 2, "mixed": 3}
CSTR_KINDS = {"trajectory": 0, "control": 1, "mixed": 2, "trajectory_bound": 3, "control_bound": 4}

OR_ERR_DOMAIN = -1
OR_ERR_RUNTIME = -2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class OrCost(C.Structure):
    _fields_ = [("kind", C.c_int), ("rows", C.c_int), ("m_cols", C.c_int), ("n_cols", C.c_int),
                ("M", _dp), ("N", _dp), ("p", _dp), ("w", _dp)]


class OrCstr(C.Structure):
    _fields_ = [("kind", C.c_int), ("rows", C.c_int), ("e_cols", C.c_int), ("g_cols", C.c_int),
                ("is_ineq", C.c_int), ("E", _dp), ("G", _dp), ("f", _dp), ("lower", _dp), ("upper", _dp)]


class OrQp(C.Structure):
    _fields_ = [("nvar", C.c_int), ("neq", C.c_int), ("nineq", C.c_int),
                ("nx", C.c_int), ("nu", C.c_int), ("N", C.c_int), ("fullX", C.c_int), ("fullU", C.c_int),
                ("Q", _dp), ("c", _dp), ("Aeq", _dp), ("beq", _dp), ("Aineq", _dp), ("bineq", _dp),
                ("lb", _dp), ("ub", _dp), ("Phi", _dp), ("Psi", _dp), ("xi", _dp)]


_lib = None


def _cpu_tag():
    """-march=native objects must not travel between hosts: tag them with a hash of this host's CPU flags"""
    import hashlib
    try:
        with open("/proc/cpuinfo") as fh:
            txt = "".join(l for l in fh if l.startswith(("model name", "flags")))[:20000]
    except OSError:
        txt = "unknown"
    return hashlib.sha1(txt.encode()).hexdigest()[:10]


def build(native=False):
    src = os.path.join(_HERE, "copra_oracle.c")
    if native:
        path = os.path.join(_HERE, "libcopra_oracle_native_%s.so" % _cpu_tag())
        if (not os.path.exists(path)) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O3", "-march=native", "-std=c99", "-fPIC", "-ffp-contract=off",
                                   "-fno-fast-math", "-pthread", "-shared", "-o", path, src, "-lm", "-lpthread"])
        return path
    path = os.path.join(_HERE, "libcopra_oracle.so")
    if (not os.path.exists(path)) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libcopra_oracle.so"], stdout=subprocess.DEVNULL)
    return path


def lib(native=False):
    global _lib
    key = bool(native)
    if _lib is None:
        _lib = {}
    if key not in _lib:
        L = C.CDLL(build(native))
        L.or_lmpc_build.restype = C.c_int
        L.or_islmpc_build.restype = C.c_int
        L.or_quadprog_dense.restype = C.c_int
        L.or_lmpc_solve.restype = C.c_int
        L.or_islmpc_solve.restype = C.c_int
        L.or_lmpc_solve_batch.restype = C.c_int
        L.or_qp_free.restype = None
        L.or_preview_update.restype = None
        _lib[key] = L
    return _lib[key]


def _f(a):
    """column-major (Fortran-order) contiguous float64 copy"""
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _ptr(a):
    return a.ctypes.data_as(_dp) if a is not None else _dp()


class _Keep:
    """keeps numpy buffers referenced by ctypes structs alive"""

    def __init__(self):
        self.bufs = []

    def mat(self, a):
        if a is None:
            return None
        a = _f(a)
        self.bufs.append(a)
        return a


def _pack_costs(costs, keep):
    arr = (OrCost * max(1, len(costs)))()
    for i, c in enumerate(costs):
        M = keep.mat(np.atleast_2d(c["M"])) if c.get("M") is not None else None
        Nm = keep.mat(np.atleast_2d(c["N"])) if c.get("N") is not None else None
        p = keep.mat(np.atleast_1d(c["p"]))
        rows = p.shape[0]
        w = c.get("weights")
        w = np.ones(rows) if w is None else np.atleast_1d(np.asarray(w, dtype=np.float64))
        if w.shape[0] != rows:  # CostFunction::weights tiling (costFunctions.h:54-67)
            if rows % w.shape[0] != 0:
                raise ValueError("weights: bad dimension")
            w = np.tile(w, rows // w.shape[0])
        w = keep.mat(w)
        arr[i].kind = COST_KINDS[c["kind"]]
        arr[i].rows = rows
        arr[i].m_cols = M.shape[1] if M is not None else 0
        arr[i].n_cols = Nm.shape[1] if Nm is not None else 0
        if M is not None and M.shape[0] != rows:
            raise ValueError("M/p rows mismatch")  # costFunctions.cpp:47-49
        if Nm is not None and Nm.shape[0] != rows:
            raise ValueError("N/p rows mismatch")
        arr[i].M, arr[i].N, arr[i].p, arr[i].w = _ptr(M), _ptr(Nm), _ptr(p), _ptr(w)
    return arr


def _pack_cstrs(cstrs, keep):
    arr = (OrCstr * max(1, len(cstrs)))()
    for i, c in enumerate(cstrs):
        kind = CSTR_KINDS[c["kind"]]
        arr[i].kind = kind
        arr[i].is_ineq = 1 if c.get("ineq", True) else 0
        if kind in (3, 4):
            lo = keep.mat(np.atleast_1d(c["lower"]))
            up = keep.mat(np.atleast_1d(c["upper"]))
            if lo.shape[0] != up.shape[0]:
                raise ValueError("lower/upper rows mismatch")  # constraints.cpp:265-267
            arr[i].rows = lo.shape[0]
            arr[i].lower, arr[i].upper = _ptr(lo), _ptr(up)
        else:
            E = keep.mat(np.atleast_2d(c["E"])) if c.get("E") is not None else None
            G = keep.mat(np.atleast_2d(c["G"])) if c.get("G") is not None else None
            f = keep.mat(np.atleast_1d(c["f"]))
            arr[i].rows = f.shape[0]
            for m in (E, G):
                if m is not None and m.shape[0] != f.shape[0]:
                    raise ValueError("E/G/f rows mismatch")
            arr[i].e_cols = E.shape[1] if E is not None else 0
            arr[i].g_cols = G.shape[1] if G is not None else 0
            arr[i].E, arr[i].G, arr[i].f = _ptr(E), _ptr(G), _ptr(f)
    return arr


def _check(rc):
    if rc == OR_ERR_DOMAIN:
        raise ValueError("std::domain_error (oracle)")
    if rc == OR_ERR_RUNTIME:
        raise RuntimeError("std::runtime_error (oracle)")


def preview(A, B, d, N):
    A, B, d = _f(A), _f(B), _f(d)
    nx, nu = B.shape
    X, U = nx * (N + 1), nu * N
    Phi = np.zeros((X, nx), order="F")
    Psi = np.zeros((X, U), order="F")
    xi = np.zeros(X)
    lib().or_preview_update(nx, nu, N, _ptr(A), _ptr(B), _ptr(d), _ptr(Phi), _ptr(Psi), _ptr(xi))
    return Phi, Psi, xi


def _qp_to_dict(qp):
    def mat(p, r, c):
        if r * c == 0:
            return np.zeros((r, c))
        return np.ctypeslib.as_array(p, shape=(c, r)).T.copy()

    def vec(p, n):
        if n == 0:
            return np.zeros(0)
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    n = qp.nvar
    out = dict(nvar=n, neq=qp.neq, nineq=qp.nineq,
               Q=mat(qp.Q, n, n), c=vec(qp.c, n), Aeq=mat(qp.Aeq, qp.neq, n), beq=vec(qp.beq, qp.neq),
               Aineq=mat(qp.Aineq, qp.nineq, n), bineq=vec(qp.bineq, qp.nineq), lb=vec(qp.lb, n), ub=vec(qp.ub, n),
               Phi=mat(qp.Phi, qp.fullX, qp.nx), Psi=mat(qp.Psi, qp.fullX, qp.fullU), xi=vec(qp.xi, qp.fullX))
    return out


def lmpc_build(A, B, d, x0, N, costs, cstrs, initial_state=None):
    """LMPC::updateSystem + makeQPForm.  initial_state = dict(R, r, x0lb, x0ub) for InitialStateLMPC."""
    keep = _Keep()
    A, B, d, x0 = _f(A), _f(B), _f(d), _f(x0)
    nx, nu = B.shape
    cc, kk = _pack_costs(costs, keep), _pack_cstrs(cstrs, keep)
    qp = OrQp()
    if initial_state is None:
        rc = lib().or_lmpc_build(nx, nu, N, _ptr(A), _ptr(B), _ptr(d), _ptr(x0), len(costs), cc, len(cstrs), kk,
                                 C.byref(qp))
    else:
        R, r = _f(initial_state["R"]), _f(initial_state["r"])
        lo, up = _f(initial_state["x0lb"]), _f(initial_state["x0ub"])
        rc = lib().or_islmpc_build(nx, nu, N, _ptr(A), _ptr(B), _ptr(d), _ptr(x0), len(costs), cc, len(cstrs), kk,
                                   _ptr(R), _ptr(r), _ptr(lo), _ptr(up), C.byref(qp))
    _check(rc)
    out = _qp_to_dict(qp)
    lib().or_qp_free(C.byref(qp))
    return out


def quadprog_dense(Q, c, Aeq, beq, Aineq, bineq, XL, XU):
    """QuadProgDenseSolver::SI_solve.  Returns (x, fail, iter[2])."""
    Q, c = _f(Q), _f(c)
    n = c.shape[0]
    Aeq = _f(np.zeros((0, n)) if Aeq is None else np.atleast_2d(Aeq)).reshape(-1, n, order="F")
    Aineq = _f(np.zeros((0, n)) if Aineq is None else np.atleast_2d(Aineq)).reshape(-1, n, order="F")
    Aeq, Aineq = _f(Aeq), _f(Aineq)
    beq = _f(np.zeros(0) if beq is None else beq)
    bineq = _f(np.zeros(0) if bineq is None else bineq)
    XL, XU = _f(XL), _f(XU)
    x = np.zeros(n)
    it = (C.c_int * 2)()
    fail = lib().or_quadprog_dense(n, Aeq.shape[0], Aineq.shape[0], _ptr(Q), _ptr(c), _ptr(Aeq), _ptr(beq),
                                   _ptr(Aineq), _ptr(bineq), _ptr(XL), _ptr(XU), _ptr(x), it)
    return x, fail, (it[0], it[1])


def lmpc_solve(A, B, d, x0, N, costs, cstrs, initial_state=None):
    """LMPC::solve (fresh controller).  Returns dict(control, trajectory, status, iter[, x0_opt])."""
    keep = _Keep()
    A, B, d, x0 = _f(A), _f(B), _f(d), _f(x0)
    nx, nu = B.shape
    X, U = nx * (N + 1), nu * N
    cc, kk = _pack_costs(costs, keep), _pack_cstrs(cstrs, keep)
    u = np.full(U, np.nan)
    tr = np.full(X, np.nan)
    it = (C.c_int * 2)()
    out = {}
    if initial_state is None:
        rc = lib().or_lmpc_solve(nx, nu, N, _ptr(A), _ptr(B), _ptr(d), _ptr(x0), len(costs), cc, len(cstrs), kk,
                                 _ptr(u), _ptr(tr), it)
    else:
        R, r = _f(initial_state["R"]), _f(initial_state["r"])
        lo, up = _f(initial_state["x0lb"]), _f(initial_state["x0ub"])
        x0o = np.full(nx, np.nan)
        rc = lib().or_islmpc_solve(nx, nu, N, _ptr(A), _ptr(B), _ptr(d), _ptr(x0), len(costs), cc, len(cstrs), kk,
                                   _ptr(R), _ptr(r), _ptr(lo), _ptr(up), _ptr(u), _ptr(tr), _ptr(x0o), it)
        out["x0_opt"] = x0o
    _check(rc)
    out.update(control=u, trajectory=tr, status=rc, iter=(it[0], it[1]))
    return out


_qlib = None


def quad_lib():
    """libcopra_oracle_quad.so: the same source compiled with `double` meaning __float128 (oracle/copra_oracle_quad.c)"""
    global _qlib
    if _qlib is None:
        path = os.path.join(_HERE, "libcopra_oracle_quad.so")
        srcs = [os.path.join(_HERE, f) for f in ("copra_oracle_quad.c", "copra_oracle.c", "copra_oracle.h")]
        if (not os.path.exists(path)) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in srcs):
            subprocess.check_call(["make", "-C", _HERE, "libcopra_oracle_quad.so"], stdout=subprocess.DEVNULL)
        _qlib = C.CDLL(path)
        _qlib.orq_solve.restype = C.c_int
    return _qlib


def lmpc_solve_quad(A, B, d, x0, N, costs, cstrs, initial_state=None):
    """LMPC::solve / InitialStateLMPC::solve by the oracle's own statements in IEEE binary128 arithmetic; doubles in, doubles out.
    Returns dict(control, trajectory, status, iter[, x0_opt])."""
    keep = _Keep()
    A, B, d, x0 = _f(A), _f(B), _f(d), _f(x0)
    nx, nu = B.shape
    X, U = nx * (N + 1), nu * N
    cc, kk = _pack_costs(costs, keep), _pack_cstrs(cstrs, keep)
    u = np.full(U, np.nan)
    tr = np.full(X, np.nan)
    x0o = np.full(nx, np.nan)
    it = (C.c_int * 2)()
    null = C.c_void_p()
    if initial_state is None:
        rc = quad_lib().orq_solve(nx, nu, N, _ptr(A), _ptr(B), _ptr(d), _ptr(x0), len(costs), cc, len(cstrs), kk, null, null, null, null,
                                  _ptr(u), _ptr(tr), _ptr(x0o), it)
    else:
        R, r = _f(initial_state["R"]), _f(initial_state["r"])
        lo, up = _f(initial_state["x0lb"]), _f(initial_state["x0ub"])
        rc = quad_lib().orq_solve(nx, nu, N, _ptr(A), _ptr(B), _ptr(d), _ptr(x0), len(costs), cc, len(cstrs), kk, _ptr(R), _ptr(r),
                                  _ptr(lo), _ptr(up), _ptr(u), _ptr(tr), _ptr(x0o), it)
    _check(rc)
    out = dict(control=u, trajectory=tr, status=rc, iter=(it[0], it[1]))
    if initial_state is not None:
        out["x0_opt"] = x0o
    return out


def lmpc_solve_batch(A, B, d, x0, N, costs, cstrs, nthreads=1, native=False):
    """Batched CPU driver: A (b,nx,nx), B (b,nx,nu), d (b,nx), x0 (b,nx) in natural numpy (row-major) indexing."""
    keep = _Keep()
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    batch, nx, nu = B.shape
    # per instance column-major
    Ab = np.ascontiguousarray(np.transpose(A, (0, 2, 1)))
    Bb = np.ascontiguousarray(np.transpose(B, (0, 2, 1)))
    db = np.ascontiguousarray(np.asarray(d, dtype=np.float64))
    xb = np.ascontiguousarray(np.asarray(x0, dtype=np.float64))
    X, U = nx * (N + 1), nu * N
    cc, kk = _pack_costs(costs, keep), _pack_cstrs(cstrs, keep)
    u = np.full((batch, U), np.nan)
    tr = np.full((batch, X), np.nan)
    st = np.zeros(batch, dtype=np.int32)
    it = np.zeros((batch, 2), dtype=np.int32)
    rc = lib(native).or_lmpc_solve_batch(batch, nthreads, nx, nu, N, _ptr(Ab), _ptr(Bb), _ptr(db), _ptr(xb),
                                         len(costs), cc, len(cstrs), kk, _ptr(u), _ptr(tr),
                                         st.ctypes.data_as(_ip), it.ctypes.data_as(_ip))
    _check(rc)
    return dict(control=u, trajectory=tr, status=st, iter=it)
