"""Shared-model tick on shapes / horizons other than the headline's compile-time builds: the Riccati-factor tier's shared-model mode on its
run-time-horizon builds against lmpc_shared.hpp (option no_ric_shared), kernel time per solve at batch 65536 (GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import random_controllers as RC  # noqa: E402
from copra_amd import BatchLMPC, workloads  # noqa: E402

b = 65536
cases = [("CoM preview N = %2d" % N, workloads.com_preview(b, N=N)) for N in (8, 12, 16, 20, 21)]
for seed in (6, 46, 55, 51):  # planar point masses of the random controllers (bounds; a reference trajectory; a target cost)
    c = RC.make_integrator(seed, b)
    cases.append(("random controller %d %s %s" % (seed, (c["nx"], c["nu"], c["N"]), c["forms"]), c))
for name, wl in cases:
    A, B, d = wl["A"][3], wl["B"][3], wl["d"][3]
    N = wl["N"]
    line, res = [], []
    for opts in (None, dict(no_ric_shared=1)):
        eng = BatchLMPC(A.shape[0], B.shape[1], N, b, wl["costs"], wl["cstrs"], options=opts)
        eng.set_shared_system(A, B, d)
        eng.set_x0(wl["x0"])
        ts = []
        for _ in range(8):
            eng.solve()
            eng.synchronize()
            ts.append(eng.last_solve_seconds())
        res.append(eng.results())
        line.append("%s %.3f ms (%.1f M solves/s)" % ("records" if opts is None else "lmpc_shared.hpp", min(ts) * 1e3, b / min(ts) / 1e6))
        eng.close()
    ok = (res[0]["status"] == 0) & (res[1]["status"] == 0)
    print("%s | iterations %.2f |" % (name, res[0]["iter"][:, 0].mean()), " | ".join(line), "| status equal", bool((res[0]["status"] == res[1]["status"]).all()),
          "iter equal", bool((res[0]["iter"][ok] == res[1]["iter"][ok]).all()), "max |dU| %.1e" % np.abs(res[0]["control"][ok] - res[1]["control"][ok]).max(), flush=True)
