"""One model, every instance its own goal, at batches below the shared lane pass's default threshold (20480): lmpc_shared.hpp (what such a
controller runs there) against the records tier behind the pass forced on (option lane_min_batch = -1); kernel ms per solve (GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from copra_amd import BatchLMPC, workloads  # noqa: E402

for b in (1024, 2048, 4096, 8192, 16384, 32768):
    rng = np.random.default_rng(5)
    wl = workloads.com_preview(b, seed=3)
    A, B, d, N = wl["A"][5], wl["B"][5], wl["d"][5], wl["N"]
    goals = workloads.COM_X_GOAL[None, :] + 0.05 * rng.standard_normal((b, 6))
    line = []
    for name, opts in (("default", None), ("pass forced on", dict(lane_min_batch=-1)), ("lmpc_shared.hpp", dict(no_ric_shared=1))):
        eng = BatchLMPC(6, 3, N, b, wl["costs"], wl["cstrs"], options=opts)
        eng.set_shared_system(A, B, d)
        eng.set_x0(wl["x0"])
        eng.set_cost_reference(0, goals)
        t = []
        for _ in range(10):
            eng.solve()
            eng.synchronize()
            t.append(eng.last_solve_seconds())
        line.append("%s %.3f ms (%.1f M/s, pass %s)" % (name, min(t) * 1e3, b / min(t) / 1e6, eng.lane_pass_info()[0]))
        eng.close()
    print("batch %6d | " % b + " | ".join(line), flush=True)
