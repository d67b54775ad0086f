"""Device-resident timing of copra_qp_solve_dense_batch for n > 64 (dev tool; GPU box only).
usage: try_dense_large.py N BATCH   env: COPRA_OPTIONS=debug=1"""
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
from copra_amd import _capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250
b = int(sys.argv[2]) if len(sys.argv) > 2 else 512
meq, mi = n // 8, n
rng = np.random.default_rng(0)
base = [F.random_dense_qp(rng, n, meq, mi) for _ in range(4)]
cm = lambda k: np.ascontiguousarray(np.stack([base[i % 4][k].T if base[i % 4][k].ndim == 2 else base[i % 4][k]
                                              for i in range(b)]))
dev = torch.device("cuda:0")
t_ = {k: torch.from_numpy(cm(k)).to(dev) for k in ("Q", "c", "Aeq", "beq", "Aineq", "bineq", "XL", "XU")}
x = torch.empty((b, n), dtype=torch.float64, device=dev)
fail = torch.empty(b, dtype=torch.int32, device=dev)
it = torch.empty((b, 2), dtype=torch.int32, device=dev)
L = _capi.lib()
p = lambda t: C.c_void_p(t.data_ptr())
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.time()
    _capi.check(L.copra_qp_solve_dense_batch(b, n, meq, mi, p(t_["Q"]), p(t_["c"]), p(t_["Aeq"]), p(t_["beq"]),
                                             p(t_["Aineq"]), p(t_["bineq"]), p(t_["XL"]), p(t_["XU"]), p(x), p(fail),
                                             p(it), 1, None))
    torch.cuda.synchronize()
    dt = time.time() - t0
print("n=%d batch=%d: %.2f ms -> %.0f QPs/s; iterations mean %.1f drops %.1f; fails %d" % (
    n, b, dt * 1e3, b / dt, it[:, 0].double().mean().item(), it[:, 1].double().mean().item(), int((fail != 0).sum())))
