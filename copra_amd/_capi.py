"""ctypes binding of the C ABI declared in include/copra_hip.h (libcopra_hip.so).

The library is hand-written HIP for gfx950; there is NO CPU fallback: if the shared object is missing or no HIP
device is usable, every entry point raises.
"""
import ctypes as C
import os

import numpy as np

# copra_code_object_check_t (include/copra_hip.h): the gate copra_batch_specialise_checked calls before it loads a code object
CODE_OBJECT_CHECK = C.CFUNCTYPE(C.c_int, C.c_char_p, C.c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcopra_hip.so")

COPRA_OK, COPRA_ERR_DOMAIN, COPRA_ERR_RUNTIME, COPRA_ERR_HIP, COPRA_ERR_UNSUPPORTED, COPRA_ERR_ARG = range(6)

COST_KINDS = {"trajectory": 0, "target": 1, "control": 2, "mixed": 3, "dense": 4}
CSTR_KINDS = {"trajectory": 0, "control": 1, "mixed": 2, "trajectory_bound": 3, "control_bound": 4, "dense": 5}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class CostDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("rows", C.c_int), ("m_cols", C.c_int), ("n_cols", C.c_int),
                ("M", _dp), ("N", _dp), ("p", _dp), ("weights", _dp),
                ("Q", _dp), ("c", _dp), ("E", _dp), ("f", _dp)]  # COPRA_COST_DENSE


class CstrDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("rows", C.c_int), ("e_cols", C.c_int), ("g_cols", C.c_int),
                ("is_inequality", C.c_int), ("E", _dp), ("G", _dp), ("f", _dp), ("lower", _dp), ("upper", _dp),
                ("A", _dp), ("b", _dp), ("Y", _dp), ("z", _dp)]  # COPRA_CSTR_DENSE


class Dims(C.Structure):
    _fields_ = [("nx", C.c_int), ("nu", C.c_int), ("N", C.c_int), ("batch", C.c_int)]


class InitialStateDesc(C.Structure):
    _fields_ = [("R", _dp), ("r", _dp)]


class Options(C.Structure):
    """copra_options_t (include/copra_hip.h): every switch the engine consults, fixed when the controller is created.  The library reads
    no environment variable; experiments steer it through the ONE variable this module parses, COPRA_OPTIONS="name=value,name=value"
    (e.g. COPRA_OPTIONS=no_lane_pass=1,lane_min_batch=-1), which seeds OPTIONS below."""
    _fields_ = ([("struct_size", C.c_int)]
                + [(n, C.c_int) for n in ("no_stage_refs", "no_step_rows", "no_selection_rows",
                                           "no_ric", "no_tri", "ric_general", "no_dense_layout", "no_q1regs", "no_ladder", "no_packed", "ric_k",
                                           "no_lane_pass", "no_lane_handover", "no_lane_spec", "no_lane_axes", "lane_min_batch", "no_ric_shared",
                                           "no_riccati", "no_ric_fast")]
                + [(n, C.c_double) for n in ("ric_step_tol", "ric_mu_tol")]
                + [("debug", C.c_int), ("no_axis_solver", C.c_int)])


OPTION_NAMES = tuple(n for n, _ in Options._fields_ if n != "struct_size")


def _options_from_env():
    out = {}
    for item in os.environ.get("COPRA_OPTIONS", "").replace(";", ",").split(","):
        item = item.strip()
        if not item:
            continue
        k, _, v = item.partition("=")
        k = k.strip()
        if k not in OPTION_NAMES:
            raise ValueError("COPRA_OPTIONS: unknown option %r (known: %s)" % (k, ", ".join(OPTION_NAMES)))
        out[k] = float(v) if k in ("ric_step_tol", "ric_mu_tol") else int(v or "1")
    return out


# defaults every controller created through this module starts from (tests: monkeypatch.setitem(_capi.OPTIONS, name, value));
# BatchLMPC(..., options={...}) overrides them per controller
OPTIONS = _options_from_env()


def make_options(overrides=None):
    """an Options struct: the library's built-in defaults, then OPTIONS, then `overrides`"""
    o = Options()
    o.struct_size = C.sizeof(Options)
    for src in (OPTIONS, overrides or {}):
        for k, v in src.items():
            if k not in OPTION_NAMES:
                raise ValueError("unknown engine option %r" % k)
            setattr(o, k, v)
    return o


def apply_default_options():
    """hand OPTIONS to the library as its process-wide defaults (the entry points without an options argument -- the dense-QP
    plug-in point, copra_plan_check, the C++ mirror's handles -- use those)"""
    o = make_options()
    check(lib().copra_set_default_options(C.byref(o)))


class CopraDomainError(ValueError):
    """std::domain_error of the reference (include/debugUtils.h:32-36)"""


class CopraRuntimeError(RuntimeError):
    """std::runtime_error of the reference (include/debugUtils.h:38-42)"""


class CopraUnsupported(NotImplementedError):
    pass


def fcol(a):
    """column-major contiguous float64"""
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def dptr(a):
    return a.ctypes.data_as(_dp) if a is not None else _dp()


def pack_costs(costs, keep):
    """costs: list of dicts {kind, M, N, p, weights}.  `keep` collects the numpy buffers the structs point into."""
    arr = (CostDesc * max(1, len(costs)))()
    for i, c in enumerate(costs):
        if c["kind"] == "dense":  # a host-evaluated user cost function: Q, c (LMPC) / Q, E, f (InitialStateLMPC)
            bufs = {k: (fcol(c[k]) if c.get(k) is not None else None) for k in ("Q", "c", "E", "f")}
            keep.extend(bufs.values())
            arr[i].kind = COST_KINDS["dense"]
            arr[i].Q, arr[i].c, arr[i].E, arr[i].f = (dptr(bufs[k]) for k in ("Q", "c", "E", "f"))
            continue
        M = fcol(np.atleast_2d(c["M"])) if c.get("M") is not None else None
        Nm = fcol(np.atleast_2d(c["N"])) if c.get("N") is not None else None
        p = fcol(np.atleast_1d(c["p"]))
        rows = p.shape[0]
        w = c.get("weights")
        w = np.ones(rows) if w is None else np.atleast_1d(np.asarray(w, dtype=np.float64))
        if w.shape[0] != rows:  # CostFunction::weights (include/costFunctions.h:54-67)
            if w.shape[0] == 0 or rows % w.shape[0] != 0:
                raise CopraDomainError("weights: bad dimension")
            w = np.tile(w, rows // w.shape[0])
        w = fcol(w)
        for m, nm in ((M, "M"), (Nm, "N")):
            if m is not None and m.shape[0] != rows:  # costFunctions.cpp:47-49
                raise CopraDomainError("%s and p must have the same number of rows (try autoSpan)" % nm)
        keep.extend([M, Nm, p, w])
        arr[i].kind = COST_KINDS[c["kind"]]
        arr[i].rows = rows
        arr[i].m_cols = M.shape[1] if M is not None else 0
        arr[i].n_cols = Nm.shape[1] if Nm is not None else 0
        arr[i].M, arr[i].N, arr[i].p, arr[i].weights = dptr(M), dptr(Nm), dptr(p), dptr(w)
    return arr


def pack_cstrs(cstrs, keep):
    arr = (CstrDesc * max(1, len(cstrs)))()
    for i, c in enumerate(cstrs):
        kind = CSTR_KINDS[c["kind"]]
        arr[i].kind = kind
        arr[i].is_inequality = 1 if c.get("ineq", True) else 0
        if kind == 5:  # a host-evaluated user constraint: A, b (LMPC) / Y, A, z (InitialStateLMPC)
            bufs = {k: (fcol(np.atleast_2d(c[k]) if k in ("A", "Y") else np.atleast_1d(c[k])) if c.get(k) is not None else None)
                    for k in ("A", "b", "Y", "z")}
            keep.extend(bufs.values())
            arr[i].rows = bufs["A"].shape[0]
            arr[i].A, arr[i].b, arr[i].Y, arr[i].z = (dptr(bufs[k]) for k in ("A", "b", "Y", "z"))
            continue
        if kind in (3, 4):
            lo = fcol(np.atleast_1d(c["lower"]))
            up = fcol(np.atleast_1d(c["upper"]))
            if lo.shape[0] != up.shape[0]:  # constraints.cpp:265-267, 339-341
                raise CopraDomainError("lower and upper must have the same number of rows (try autoSpan)")
            keep.extend([lo, up])
            arr[i].rows = lo.shape[0]
            arr[i].lower, arr[i].upper = dptr(lo), dptr(up)
        else:
            E = fcol(np.atleast_2d(c["E"])) if c.get("E") is not None else None
            G = fcol(np.atleast_2d(c["G"])) if c.get("G") is not None else None
            f = fcol(np.atleast_1d(c["f"]))
            for m, nm in ((E, "E"), (G, "G")):
                if m is not None and m.shape[0] != f.shape[0]:  # constraints.cpp:47-49, 112-114, 173-178
                    raise CopraDomainError("%s and f must have the same number of rows (try autoSpan)" % nm)
            keep.extend([E, G, f])
            arr[i].rows = f.shape[0]
            arr[i].e_cols = E.shape[1] if E is not None else 0
            arr[i].g_cols = G.shape[1] if G is not None else 0
            arr[i].E, arr[i].G, arr[i].f = dptr(E), dptr(G), dptr(f)
    return arr


_lib = None
_SRC_EXT = (".hip", ".hpp", ".h", ".inc")


def source_hash():
    """sha1 over the HIP sources + the ABI header: recorded next to the .so at build time"""
    import hashlib
    h = hashlib.sha1()
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(f for f in os.listdir(csrc) if f.endswith(_SRC_EXT) or f == "Makefile")
    for f in files:
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    with open(os.path.join(_HERE, "..", "include", "copra_hip.h"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()


def build_library(force=False):
    """(Re)build libcopra_hip.so with hipcc when the sources changed since the last build.  Never falls back to
    anything else: without hipcc a stale or missing library is an error the caller sees."""
    import fcntl
    import subprocess
    stamp = LIB_PATH + ".srchash"
    want = source_hash()

    def current():
        have = open(stamp).read().strip() if os.path.exists(stamp) else ""
        return os.path.exists(LIB_PATH) and have == want

    if current() and not force:
        return False
    if os.environ.get("COPRA_NO_BUILD"):
        # profilers preload a library that initialises the GPU in every child: never spawn make -> hipcc from there
        raise ImportError("copra_amd: libcopra_hip.so is missing or older than its sources and COPRA_NO_BUILD is set "
                          "(build first: python -c 'import __graft_entry__ as g; g.build()')")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        if os.path.exists(LIB_PATH) and not os.path.exists(stamp):
            return False  # a prebuilt library without its stamp on a box that cannot rebuild it: used as it is
        if os.path.exists(LIB_PATH):
            raise ImportError("copra_amd: libcopra_hip.so is older than its sources and hipcc is not available")
        raise ImportError("copra_amd: libcopra_hip.so has not been built and hipcc is not available")
    # several ranks of one job may get here at once: build under an exclusive lock, re-check after acquiring it
    with open(os.path.join(_HERE, "csrc", ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if current() and not force:
                return False
            subprocess.check_call(["make", "-B", "-j4", "-C", os.path.join(_HERE, "csrc"), "libcopra_hip.so"],
                                  stdout=subprocess.DEVNULL)  # (the Makefile writes the stamp, same hash formula)
            have = open(stamp).read().strip() if os.path.exists(stamp) else ""
            if have != want:
                raise ImportError("copra_amd: Makefile and _capi.source_hash() disagree on the source hash")
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return True


def lib():
    """Load libcopra_hip.so (rebuilding it first if its sources changed); raises -- no fallback -- otherwise."""
    global _lib
    if _lib is None:
        build_library()
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "copra_amd: %s is missing -- build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()' or make -C copra_amd/csrc). There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.copra_batch_create.restype = C.c_int
        L.copra_batch_create.argtypes = [C.POINTER(vp), C.POINTER(Dims), C.c_int, C.POINTER(CostDesc), C.c_int,
                                         C.POINTER(CstrDesc)]
        L.copra_batch_create_with_options.restype = C.c_int
        L.copra_batch_create_with_options.argtypes = [C.POINTER(vp), C.POINTER(Dims), C.c_int, C.POINTER(CostDesc), C.c_int,
                                                      C.POINTER(CstrDesc), C.POINTER(InitialStateDesc), C.POINTER(Options)]
        L.copra_set_default_options.restype = C.c_int
        L.copra_set_default_options.argtypes = [C.POINTER(Options)]
        L.copra_options_init.restype = None
        L.copra_options_init.argtypes = [C.POINTER(Options)]
        L.copra_batch_create_initial_state.restype = C.c_int
        L.copra_batch_create_initial_state.argtypes = [C.POINTER(vp), C.POINTER(Dims), C.c_int, C.POINTER(CostDesc),
                                                       C.c_int, C.POINTER(CstrDesc), C.POINTER(InitialStateDesc)]
        L.copra_batch_set_initial_state_bounds.restype = C.c_int
        L.copra_batch_set_initial_state_bounds.argtypes = [vp, vp, vp, C.c_int]
        L.copra_batch_get_initial_state.restype = C.c_int
        L.copra_batch_get_initial_state.argtypes = [vp, vp]
        L.copra_batch_destroy.restype = None
        L.copra_batch_destroy.argtypes = [vp]
        L.copra_batch_set_system.restype = C.c_int
        L.copra_batch_set_system.argtypes = [vp, vp, vp, vp, vp, C.c_int]
        L.copra_batch_set_x0.restype = C.c_int
        L.copra_batch_set_x0.argtypes = [vp, vp, C.c_int]
        L.copra_batch_lanes_per_instance.restype = C.c_int
        L.copra_batch_lanes_per_instance.argtypes = [vp]
        L.copra_batch_specialise.restype = C.c_int
        L.copra_batch_specialise.argtypes = [vp, C.c_char_p]
        L.copra_batch_specialise_checked.restype = C.c_int
        L.copra_batch_specialise_checked.argtypes = [vp, C.c_char_p, CODE_OBJECT_CHECK, vp]
        L.copra_batch_layout_info.restype = C.c_int
        L.copra_batch_layout_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
        L.copra_qp_dense_specialise.restype = C.c_int
        L.copra_qp_dense_specialise.argtypes = [C.c_int, C.c_char_p]
        L.copra_plan_check.restype = C.c_int
        L.copra_plan_check.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp]
        L.copra_batch_set_constraint_rhs.restype = C.c_int
        L.copra_batch_set_constraint_rhs.argtypes = [vp, C.c_int, vp, C.c_int]
        L.copra_batch_set_control_bounds.restype = C.c_int
        L.copra_batch_set_control_bounds.argtypes = [vp, vp, vp, C.c_int]
        L.copra_batch_set_cost_reference.restype = C.c_int
        L.copra_batch_set_cost_reference.argtypes = [vp, C.c_int, vp, C.c_int]
        L.copra_batch_set_cost_reference_all.restype = C.c_int
        L.copra_batch_set_cost_reference_all.argtypes = [vp, C.c_int, vp, C.c_int]
        L.copra_batch_set_shared_system.restype = C.c_int
        L.copra_batch_set_shared_system.argtypes = [vp, vp, vp, vp, C.c_int]
        L.copra_batch_set_outputs.restype = C.c_int
        L.copra_batch_set_outputs.argtypes = [vp, vp, vp, vp, vp]
        L.copra_batch_solve.restype = C.c_int
        L.copra_batch_solve.argtypes = [vp, vp]
        L.copra_batch_synchronize.restype = C.c_int
        L.copra_batch_synchronize.argtypes = [vp]
        for nm in ("copra_batch_control_device", "copra_batch_trajectory_device", "copra_batch_status_device",
                   "copra_batch_iter_device"):
            getattr(L, nm).restype = vp
            getattr(L, nm).argtypes = [vp]
        L.copra_batch_get_results.restype = C.c_int
        L.copra_batch_get_results.argtypes = [vp, vp, vp, vp, vp]
        L.copra_batch_qp_sizes.restype = C.c_int
        L.copra_batch_qp_sizes.argtypes = [vp, _ip, _ip, _ip]
        L.copra_batch_dump_qp.restype = C.c_int
        L.copra_batch_dump_qp.argtypes = [vp, C.c_int] + [vp] * 8
        L.copra_batch_set_system_rowmajor_async.restype = C.c_int
        L.copra_batch_set_system_rowmajor_async.argtypes = [vp, vp, vp, vp, vp, vp]
        L.copra_batch_last_solve_seconds.restype = C.c_int
        L.copra_batch_last_solve_seconds.argtypes = [vp, _dp]
        L.copra_batch_last_first_tier_seconds.restype = C.c_int
        L.copra_batch_last_first_tier_seconds.argtypes = [vp, _dp]
        L.copra_batch_lane_pass_info.restype = C.c_int
        L.copra_batch_lane_pass_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.copra_batch_phase_profile.restype = C.c_int
        L.copra_batch_phase_profile.argtypes = [vp, C.c_int, vp]
        L.copra_qp_solve_dense_batch.restype = C.c_int
        L.copra_qp_solve_dense_batch.argtypes = [C.c_int] * 4 + [vp] * 11 + [C.c_int, vp]
        L.copra_last_error.restype = C.c_char_p
        L.copra_device_info.restype = C.c_int
        L.copra_device_info.argtypes = [_ip, _ip, C.c_char_p, C.c_int]
        L.copra_abi_version.restype = C.c_int
        L.copra_source_hash.restype = C.c_char_p
        L.copra_preview_update.restype = C.c_int
        L.copra_preview_update.argtypes = [C.c_int] * 3 + [vp] * 6
        L.copra_batch_set_warm_start.restype = C.c_int
        L.copra_batch_set_warm_start.argtypes = [vp, C.c_int]
        L.copra_batch_select_solver.restype = C.c_int
        L.copra_batch_select_solver.argtypes = [vp, C.c_int]
        L.copra_batch_solver_info.restype = C.c_int
        L.copra_batch_solver_info.argtypes = [vp]
        _lib = L
    return _lib


def library_source_hash():
    """the source hash compiled into the LOADED library (copra_source_hash)"""
    return lib().copra_source_hash().decode()


def check(rc):
    if rc == COPRA_OK:
        return
    msg = lib().copra_last_error().decode("utf-8", "replace")
    if rc == COPRA_ERR_DOMAIN:
        raise CopraDomainError(msg)
    if rc == COPRA_ERR_RUNTIME:
        raise CopraRuntimeError(msg)
    if rc == COPRA_ERR_UNSUPPORTED:
        raise CopraUnsupported(msg)
    if rc == COPRA_ERR_HIP:
        raise RuntimeError("HIP: " + msg)
    raise ValueError("copra_hip: bad argument: " + msg)
