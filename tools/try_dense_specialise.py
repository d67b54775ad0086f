"""copra_qp_solve_dense_batch before and after copra_qp_dense_specialise, device-resident data."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import fixtures as F  # noqa: E402
from copra_amd import _capi, qp_dense_specialise  # noqa: E402

L = _capi.lib()
dev = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr())


def few_active(rng, n, meq, mi, k_active):
    """MPC-like: the unconstrained minimiser violates only k_active of the inequalities, the bounds are wide"""
    P = F.random_dense_qp(rng, n, meq, mi, tight=1.0)
    xs = np.linalg.lstsq(P["Aeq"], P["beq"], rcond=None)[0] if meq else np.zeros(n)
    xs = xs + 0.0  # a point on the equalities
    P["c"] = -P["Q"] @ xs
    P["bineq"] = P["Aineq"] @ xs + 1.0
    cut = rng.choice(mi, size=k_active, replace=False)
    P["bineq"][cut] = P["Aineq"][cut] @ xs - 0.05
    P["XL"], P["XU"] = xs - 10.0, xs + 10.0
    return P


def run(n, meq, mi, b, k_active=None):
    rng = np.random.default_rng(n)
    if k_active is not None:
        base = [few_active(rng, n, meq, mi, k_active) for _ in range(64)]
    else:
        base = [F.random_dense_qp(rng, n, meq, mi, tight=0.05 + 0.5 * rng.random()) for _ in range(64)]
    idx = rng.integers(0, 64, b)
    cm = lambda k: np.ascontiguousarray(np.stack([np.asarray(P[k]).T if np.asarray(P[k]).ndim == 2 else P[k] for P in base])[idx])
    T = {k: torch.from_numpy(cm(k)).to(dev) for k in ("Q", "c", "Aeq", "beq", "Aineq", "bineq", "XL", "XU")}
    x = torch.empty((b, n), dtype=torch.float64, device=dev)
    fail = torch.empty(b, dtype=torch.int32, device=dev)
    it = torch.empty((b, 2), dtype=torch.int32, device=dev)

    def rate():
        ts = []
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _capi.check(L.copra_qp_solve_dense_batch(b, n, meq, mi, p(T["Q"]), p(T["c"]), p(T["Aeq"]), p(T["beq"]), p(T["Aineq"]),
                                                     p(T["bineq"]), p(T["XL"]), p(T["XU"]), p(x), p(fail), p(it), 1, None))
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return b / float(np.mean(ts[2:])) / 1e6, x.clone(), int((fail != 0).sum())

    r0, x0, f0 = rate()
    t0 = time.perf_counter()
    qp_dense_specialise(n)
    tc = time.perf_counter() - t0
    r1, x1, f1 = rate()
    same = bool(torch.equal(torch.nan_to_num(x0), torch.nan_to_num(x1)))
    print("dense QP n=%2d meq=%d mineq=%2d batch %6d%s: %.2f -> %.2f M QPs/s (compile %.1f s), fails %d/%d, identical %s, mean iterations %.1f"
          % (n, meq, mi, b, "" if k_active is None else " (%d cut)" % k_active, r0, r1, tc, f0, f1, same,
             float(it[:, 0].double().mean())), flush=True)


run(6, 3, 2, 262144)
run(12, 2, 9, 262144)
run(24, 3, 16, 131072)
run(40, 3, 25, 65536)
run(60, 4, 40, 32768)
# MPC-like problems (few active constraints)
run(40, 2, 26, 65536, k_active=3)
run(60, 3, 41, 32768, k_active=3)
run(60, 2, 42, 32768, k_active=8)
