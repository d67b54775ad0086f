"""Where the one-instance-per-lane pass starts to pay for other shapes (GPU box): step time with the pass forced on / off over batch sizes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from copra_amd import BatchLMPC, workloads  # noqa: E402
from copra_amd import _capi  # engine options (copra_options_t) instead of the COPRA_* environment variables of earlier rounds


def t(wl, b, on):
    _capi.OPTIONS["lane_min_batch"] = int("1") if on else "100000000"
    nx, nu = wl["B"].shape[1], wl["B"].shape[2]
    eng = BatchLMPC(nx, nu, wl["N"], b, wl["costs"], wl["cstrs"])
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    ts = []
    for _ in range(10):
        eng.solve()
        eng.synchronize()
        ts.append(eng.last_solve_seconds())
    ran = eng.lane_pass_info()
    eng.close()
    return float(np.mean(ts[5:])) * 1e3, ran


for name, make in (("falling mass N=32", lambda b: workloads.double_integrator(b, N=32)), ("falling mass N=64", lambda b: workloads.double_integrator(b, N=64)),
                   ("CoM N=10", lambda b: workloads.com_preview(b, N=10)), ("CoM N=15", lambda b: workloads.com_preview(b, N=15))):
    for b in (1024, 2048, 4096, 8192, 16384):
        wl = make(b)
        off, _ = t(wl, b, False)
        on, ran = t(wl, b, True)
        print("%-18s batch %6d: %.4f ms without, %.4f ms with the pass %s" % (name, b, off, on, ran))
