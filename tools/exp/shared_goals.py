"""One model for the batch, every instance its own GOAL (copra_batch_set_cost_reference on the per-step TrajectoryCost of the headline
workload): shared-model mode (lmpc_shared.hpp: the records tier needs controller-wide references) against per-instance systems with the
one model repeated; kernel time per solve at batch 65536 (GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from copra_amd import BatchLMPC, workloads  # noqa: E402

b = 65536
rng = np.random.default_rng(5)
wl = workloads.com_preview(b, seed=3)
A, B, d, N = wl["A"][5], wl["B"][5], wl["d"][5], wl["N"]
goals = workloads.COM_X_GOAL[None, :] + 0.05 * rng.standard_normal((b, 6))
out = []
for mode in ("shared model", "per-instance systems"):
    eng = BatchLMPC(6, 3, N, b, wl["costs"], wl["cstrs"])
    if mode == "shared model":
        eng.set_shared_system(A, B, d)
        eng.set_x0(wl["x0"])
    else:
        eng.set_system(np.tile(A, (b, 1, 1)), np.tile(B, (b, 1, 1)), np.tile(d, (b, 1)), wl["x0"])
    eng.set_cost_reference(0, goals)
    t = []
    for _ in range(10):
        eng.solve()
        eng.synchronize()
        t.append(eng.last_solve_seconds())
    out.append(eng.results())
    print("%-22s %.3f ms per solve, %.1f M solves/s, mean iterations %.2f" % (mode, min(t) * 1e3, b / min(t) / 1e6, out[-1]["iter"][:, 0].mean()), flush=True)
    eng.close()
ok = (out[0]["status"] == 0) & (out[1]["status"] == 0)
print("status equal", bool((out[0]["status"] == out[1]["status"]).all()), "iterations equal", bool((out[0]["iter"][ok] == out[1]["iter"][ok]).all()),
      "max |dU| %.1e" % np.abs(out[0]["control"][ok] - out[1]["control"][ok]).max())
