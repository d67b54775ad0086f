"""Packed small-problem kernels vs one wave per instance (COPRA_OPTIONS=no_packed=1): kernel rates for a few small shapes."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import fixtures as F  # noqa: E402
from copra_amd import BatchLMPC, qp_solve_dense_batch, workloads  # noqa: E402


def rate(eng, batch):
    ts = []
    for _ in range(6):
        eng.solve()
        eng.synchronize()
        ts.append(eng.last_solve_seconds())
    return batch / float(np.mean(ts[2:])) / 1e6


b = 262144
wl = workloads.double_integrator(b)
eng = BatchLMPC(2, 1, wl["N"], b, wl["costs"], wl["cstrs"])
eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
print("config 2 (n=10):            %.1f M solves/s" % rate(eng, b))
eng = BatchLMPC(2, 1, wl["N"], b, wl["costs"], wl["cstrs"])
eng.set_shared_system(wl["A"][0], wl["B"][0], wl["d"][0])
eng.set_x0(wl["x0"])
print("config 2 shared model:      %.1f M solves/s" % rate(eng, b))
b = 65536
pb = F.com_walk_problem()
eng = BatchLMPC(6, 3, pb["N"], b, pb["costs"], pb["cstrs"])
x0 = np.tile(pb["x0"], (b, 1)) + 0.01 * np.random.default_rng(0).standard_normal((b, 6))
eng.set_system(np.tile(pb["A"], (b, 1, 1)), np.tile(pb["B"], (b, 1, 1)), np.tile(pb["d"], (b, 1)), x0)
print("CoM walk (n=30, 66 rows):   %.2f M solves/s" % rate(eng, b))
pb = F.bounded_system("trajectory", N=12)
eng = BatchLMPC(2, 1, 12, b, pb["costs"], pb["cstrs"], initial_state=dict(R=10.0 * np.eye(2), r=np.zeros(2)))
x0 = np.tile(pb["x0"], (b, 1))
eng.set_system(np.tile(pb["A"], (b, 1, 1)), np.tile(pb["B"], (b, 1, 1)), np.tile(pb["d"], (b, 1)), x0)
eng.set_initial_state_bounds(x0 - 0.05, x0 + 0.05)
print("InitialStateLMPC (nvar=14): %.2f M solves/s" % rate(eng, b))
wl = workloads.com_preview(b, N=10)
eng = BatchLMPC(6, 3, 10, b, wl["costs"], wl["cstrs"])
eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
print("CoM preview N=10 (n=30):    %.2f M solves/s" % rate(eng, b))
P = F.scilab_problem()
dev = torch.device("cuda:0")
cm = lambda a: np.ascontiguousarray(np.tile(a.T if a.ndim == 2 else a, (b,) + (1,) * a.ndim))
T = {k: torch.from_numpy(cm(np.asarray(P[k], dtype=np.float64))).to(dev) for k in ("Q", "c", "Aeq", "beq", "Aineq", "bineq", "XL", "XU")}
x = torch.empty((b, 6), dtype=torch.float64, device=dev)
fail = torch.empty(b, dtype=torch.int32, device=dev)
it = torch.empty((b, 2), dtype=torch.int32, device=dev)
from copra_amd import _capi  # noqa: E402
L = _capi.lib()
p = lambda t: C.c_void_p(t.data_ptr())
ts = []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _capi.check(L.copra_qp_solve_dense_batch(b, 6, 3, 2, p(T["Q"]), p(T["c"]), p(T["Aeq"]), p(T["beq"]), p(T["Aineq"]),
                                             p(T["bineq"]), p(T["XL"]), p(T["XU"]), p(x), p(fail), p(it), 1, None))
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print("dense QP n=6 (device data): %.1f M QPs/s; fails %d; x[0] %s" % (b / float(np.mean(ts[2:])) / 1e6, int((fail != 0).sum()), x[0].cpu().numpy().round(4)))
