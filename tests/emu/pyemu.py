"""ctypes front-end of the CPU wave emulator (tests/emu/emu_harness.cpp) -- TEST INFRASTRUCTURE ONLY.

Runs the HIP kernel BODIES (copra_amd/csrc/*.hpp) lane-by-lane on the CPU so that `pytest -m "not gpu"` can compare
the kernel logic with the oracle without a GPU.  Never used by the product path.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, "..", ".."))
from copra_amd import _capi  # noqa: E402  (struct definitions only; does not load the HIP library)

_lib = None


def lib():
    global _lib
    if _lib is None:
        import fcntl
        with open(os.path.join(_HERE, ".build.lock"), "w") as lock:  # (pytest -n: one worker builds, the others wait for it)
            fcntl.flock(lock, fcntl.LOCK_EX)
            subprocess.check_call(["make", "-C", _HERE, "libcopra_emu.so"], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(os.path.join(_HERE, "libcopra_emu.so"))
        _lib.emu_lmpc_solve.restype = C.c_int
        _lib.emu_qp_dense.restype = C.c_int
        _lib.emu_lmpc_solve_shared.restype = C.c_int
        _lib.emu_lmpc_solve_riccati.restype = C.c_int
    return _lib


def _sync_options():
    """hand _capi.OPTIONS (what BatchLMPC would pass to copra_batch_create_with_options) to the harness"""
    o = _capi.make_options()
    lib().emu_set_options(C.byref(o))


def last_lane_hist():
    """histogram (32 bins, the last one open) of the violated-row counts of the instances the last lane pass left to the first tier"""
    out = (C.c_int * 32)()
    lib().emu_last_lane_hist(out)
    return np.array(list(out))


def _batchify(A, B, d, x0):
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    if A.ndim == 2:
        A, B = A[None], B[None]
        d, x0 = np.asarray(d, dtype=np.float64)[None], np.asarray(x0, dtype=np.float64)[None]
    # per-instance column-major
    Ab = np.ascontiguousarray(np.transpose(A, (0, 2, 1)))
    Bb = np.ascontiguousarray(np.transpose(B, (0, 2, 1)))
    return Ab, Bb, np.ascontiguousarray(d, dtype=np.float64), np.ascontiguousarray(x0, dtype=np.float64)


def lmpc_solve(A, B, d, x0, N, costs, cstrs, dump_instance=-1, specialised=True, initial_state=None, cost_refs=None,
               row_rhs=None, bounds=None):
    """cost_refs: {cost_index: array (batch, rows)} per-instance references (copra_batch_set_cost_reference);
    row_rhs: (batch, mgen) per-instance right-hand sides in stacked row order; bounds: (lower, upper) each (batch, n)"""
    _sync_options()
    rr = None if row_rhs is None else np.ascontiguousarray(row_rhs, dtype=np.float64)
    lo = None if bounds is None else np.ascontiguousarray(bounds[0], dtype=np.float64)
    up = None if bounds is None else np.ascontiguousarray(bounds[1], dtype=np.float64)
    lib().emu_set_instance_rows(_capi.dptr(rr) if rr is not None else C.c_void_p(),
                                _capi.dptr(lo) if lo is not None else C.c_void_p(),
                                _capi.dptr(up) if up is not None else C.c_void_p())
    Ab, Bb, db, xb = _batchify(A, B, d, x0)
    refs = {int(k): np.ascontiguousarray(v, dtype=np.float64) for k, v in (cost_refs or {}).items()}
    for k in range(8):
        lib().emu_set_cost_reference(k, _capi.dptr(refs[k]) if k in refs else C.c_void_p())
    batch, nu, nx = Bb.shape[0], Bb.shape[1], Bb.shape[2]
    keep = []
    cc = _capi.pack_costs(costs, keep)
    kk = _capi.pack_cstrs(cstrs, keep)
    dims = _capi.Dims(nx, nu, N, batch)
    sizes = (C.c_int * 9)()
    vp = C.c_void_p
    isd, x0lb, x0ub, x0o = vp(), None, None, None
    if initial_state is not None:
        Rm = _capi.fcol(initial_state["R"])
        rv = _capi.fcol(initial_state["r"])
        keep.extend([Rm, rv])
        isd = _capi.InitialStateDesc(_capi.dptr(Rm), _capi.dptr(rv))
        if initial_state.get("x0lb") is not None:
            x0lb = np.ascontiguousarray(np.broadcast_to(initial_state["x0lb"], (batch, nx)), dtype=np.float64)
            x0ub = np.ascontiguousarray(np.broadcast_to(initial_state["x0ub"], (batch, nx)), dtype=np.float64)
        x0o = np.full((batch, nx), np.nan)
        isd = C.byref(isd)
    rc = lib().emu_lmpc_solve(C.byref(dims), len(costs), cc, len(cstrs), kk, vp(), vp(), vp(), vp(), vp(), vp(), vp(),
                              vp(), -1, vp(), vp(), vp(), vp(), sizes, 0, isd, vp(), vp(), vp())
    if rc == _capi.COPRA_ERR_DOMAIN:
        raise _capi.CopraDomainError("emu")
    if rc == _capi.COPRA_ERR_RUNTIME:
        raise _capi.CopraRuntimeError("emu")
    if rc == _capi.COPRA_ERR_UNSUPPORTED:
        raise _capi.CopraUnsupported("emu")
    n, neq, nineq = sizes[0], sizes[1], sizes[2]  # n = number of decision variables of the QP
    mgen = neq + nineq
    X = nx * (N + 1)
    u = np.full((batch, nu * N), np.nan)
    tr = np.full((batch, X), np.nan)
    st = np.full(batch, -1, dtype=np.int32)
    it = np.zeros((batch, 2), dtype=np.int32)
    dQ = np.zeros((n, n), order="F")
    dc = np.zeros(n)
    dA = np.zeros((max(mgen, 1), n), order="F")
    db_ = np.zeros(max(mgen, 1))
    p = _capi.dptr
    rc = lib().emu_lmpc_solve(C.byref(dims), len(costs), cc, len(cstrs), kk, p(Ab), p(Bb), p(db), p(xb), p(u), p(tr),
                              st.ctypes.data_as(C.POINTER(C.c_int)), it.ctypes.data_as(C.POINTER(C.c_int)),
                              dump_instance, p(dQ), p(dc), p(dA), p(db_), sizes, 1 if specialised else 0, isd,
                              p(x0lb) if x0lb is not None else vp(), p(x0ub) if x0ub is not None else vp(),
                              p(x0o) if x0o is not None else vp())
    if rc != 0:
        raise RuntimeError("emulator failed rc=%d" % rc)
    out = dict(control=u, trajectory=tr, status=st, iter=it, lds_bytes=sizes[3], overflowed=sizes[4], rcap=sizes[5], factor_only=bool(sizes[6]),
               riccati_factor=bool(sizes[7]), lane_pass_finished=sizes[8])
    if x0o is not None:
        out["x0_opt"] = x0o
    if dump_instance >= 0:
        out.update(Q=np.array(dQ), c=dc, Aeq=np.array(dA[:neq]), beq=db_[:neq], Aineq=np.array(dA[neq:mgen]),
                   bineq=db_[neq:mgen])
    return out


def lmpc_solve_riccati(A, B, d, x0, N, costs, cstrs, initial_state=None, cost_refs=None, row_rhs=None, bounds=None):
    """lmpc_riccati.hpp (stage-wise interior-point body) for every instance; returns None when the controller is not
    stage-wise.  Arguments as lmpc_solve."""
    _sync_options()
    rr = None if row_rhs is None else np.ascontiguousarray(row_rhs, dtype=np.float64)
    lo = None if bounds is None else np.ascontiguousarray(bounds[0], dtype=np.float64)
    up = None if bounds is None else np.ascontiguousarray(bounds[1], dtype=np.float64)
    vp = C.c_void_p
    p = _capi.dptr
    lib().emu_set_instance_rows(p(rr) if rr is not None else vp(), p(lo) if lo is not None else vp(),
                                p(up) if up is not None else vp())
    Ab, Bb, db, xb = _batchify(A, B, d, x0)
    refs = {int(k): np.ascontiguousarray(v, dtype=np.float64) for k, v in (cost_refs or {}).items()}
    for k in range(8):
        lib().emu_set_cost_reference(k, p(refs[k]) if k in refs else vp())
    batch, nu, nx = Bb.shape[0], Bb.shape[1], Bb.shape[2]
    keep = []
    cc = _capi.pack_costs(costs, keep)
    kk = _capi.pack_cstrs(cstrs, keep)
    dims = _capi.Dims(nx, nu, N, batch)
    isd, x0lb, x0ub, x0o = vp(), None, None, None
    if initial_state is not None:
        Rm = _capi.fcol(initial_state["R"])
        rv = _capi.fcol(initial_state["r"])
        keep.extend([Rm, rv])
        isd = C.byref(_capi.InitialStateDesc(p(Rm), p(rv)))
        if initial_state.get("x0lb") is not None:
            x0lb = np.ascontiguousarray(np.broadcast_to(initial_state["x0lb"], (batch, nx)), dtype=np.float64)
            x0ub = np.ascontiguousarray(np.broadcast_to(initial_state["x0ub"], (batch, nx)), dtype=np.float64)
        x0o = np.full((batch, nx), np.nan)
    u = np.full((batch, nu * N), np.nan)
    tr = np.full((batch, nx * (N + 1)), np.nan)
    st = np.full(batch, -1, dtype=np.int32)
    it = np.zeros((batch, 2), dtype=np.int32)
    nc = (C.c_int * 2)()
    rc = lib().emu_lmpc_solve_riccati(C.byref(dims), len(costs), cc, len(cstrs), kk, p(Ab), p(Bb), p(db), p(xb), p(u),
                                      p(tr), st.ctypes.data_as(C.POINTER(C.c_int)), it.ctypes.data_as(C.POINTER(C.c_int)),
                                      isd, p(x0lb) if x0lb is not None else vp(), p(x0ub) if x0ub is not None else vp(),
                                      p(x0o) if x0o is not None else vp(), nc)
    if rc == -200:
        return None
    if rc != 0:
        raise RuntimeError("emulator failed rc=%d" % rc)
    out = dict(control=u, trajectory=tr, status=st, iter=it, not_converged=nc[0], lds_resident=bool(nc[1]))
    if x0o is not None:
        out["x0_opt"] = x0o
    return out


WARM_CAP = 32  # plan.hpp::kWarmCap


def lmpc_solve_shared(A, B, d, x0, N, costs, cstrs, warm=None, cost_refs=None):
    """shared-model fast path: ONE system (A, B, d), x0 of shape (batch, nx).  warm: int32 array (batch, WARM_CAP) kept by
    the caller across receding-horizon ticks (initialised to -1): the active set of the previous tick, shifted by one step.
    cost_refs: {cost_index: array (batch, rows)} per-instance references -- the records form with the pass in front only"""
    _sync_options()
    refs = {int(k): np.ascontiguousarray(v, dtype=np.float64) for k, v in (cost_refs or {}).items()}
    for k in range(8):
        lib().emu_set_cost_reference(k, _capi.dptr(refs[k]) if k in refs else C.c_void_p())
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    x0 = np.ascontiguousarray(np.atleast_2d(x0), dtype=np.float64)
    batch, nx, nu = x0.shape[0], A.shape[0], B.shape[1]
    keep = []
    cc = _capi.pack_costs(costs, keep)
    kk = _capi.pack_cstrs(cstrs, keep)
    dims = _capi.Dims(nx, nu, N, batch)
    Ac, Bc = np.ascontiguousarray(A.T), np.ascontiguousarray(B.T)  # column-major
    dc = np.ascontiguousarray(d, dtype=np.float64)
    u = np.full((batch, nu * N), np.nan)
    tr = np.full((batch, nx * (N + 1)), np.nan)
    st = np.full(batch, -1, dtype=np.int32)
    it = np.zeros((batch, 2), dtype=np.int32)
    sizes = (C.c_int * 3)()
    p = _capi.dptr
    rc = lib().emu_lmpc_solve_shared(C.byref(dims), len(costs), cc, len(cstrs), kk, p(Ac), p(Bc), p(dc), p(x0), p(u),
                                     p(tr), st.ctypes.data_as(C.POINTER(C.c_int)),
                                     it.ctypes.data_as(C.POINTER(C.c_int)), sizes,
                                     warm.ctypes.data_as(C.POINTER(C.c_int)) if warm is not None else C.c_void_p())
    for k in range(8):
        lib().emu_set_cost_reference(k, C.c_void_p())
    if rc != 0:
        raise RuntimeError("emulator failed rc=%d" % rc)
    return dict(control=u, trajectory=tr, status=st, iter=it, overflowed=sizes[0], riccati_factor=bool(sizes[1]), lane_pass_finished=sizes[2])


def qp_dense(Q, c, Aeq, beq, Aineq, bineq, XL, XU):
    """batched: Q (b,n,n), c (b,n), Aeq (b,meq,n) ... natural numpy indexing"""
    _sync_options()
    Q = np.asarray(Q, dtype=np.float64)
    if Q.ndim == 2:
        Q, c, XL, XU = Q[None], np.asarray(c)[None], np.asarray(XL)[None], np.asarray(XU)[None]
        Aeq = None if Aeq is None else np.asarray(Aeq)[None]
        beq = None if beq is None else np.asarray(beq)[None]
        Aineq = None if Aineq is None else np.asarray(Aineq)[None]
        bineq = None if bineq is None else np.asarray(bineq)[None]
    b, n = Q.shape[0], Q.shape[1]
    Aeq = np.zeros((b, 0, n)) if Aeq is None else np.asarray(Aeq, dtype=np.float64).reshape(b, -1, n)
    Aineq = np.zeros((b, 0, n)) if Aineq is None else np.asarray(Aineq, dtype=np.float64).reshape(b, -1, n)
    beq = np.zeros((b, 0)) if beq is None else np.asarray(beq, dtype=np.float64).reshape(b, -1)
    bineq = np.zeros((b, 0)) if bineq is None else np.asarray(bineq, dtype=np.float64).reshape(b, -1)
    neq, nineq = Aeq.shape[1], Aineq.shape[1]
    cm = lambda a: np.ascontiguousarray(np.transpose(a, (0, 2, 1)))  # per-instance column-major
    Qb, Aeqb, Aineqb = cm(Q), cm(Aeq), cm(Aineq)
    cb = np.ascontiguousarray(c, dtype=np.float64)
    beqb, bineqb = np.ascontiguousarray(beq), np.ascontiguousarray(bineq)
    XLb, XUb = np.ascontiguousarray(XL, dtype=np.float64), np.ascontiguousarray(XU, dtype=np.float64)
    x = np.full((b, n), np.nan)
    fail = np.full(b, -1, dtype=np.int32)
    it = np.zeros((b, 2), dtype=np.int32)
    p = _capi.dptr
    rc = lib().emu_qp_dense(b, n, neq, nineq, p(Qb), p(cb), p(Aeqb), p(beqb), p(Aineqb), p(bineqb), p(XLb), p(XUb),
                            p(x), fail.ctypes.data_as(C.POINTER(C.c_int)), it.ctypes.data_as(C.POINTER(C.c_int)))
    if rc != 0:
        raise RuntimeError("emulator failed rc=%d" % rc)
    return x, fail, it
