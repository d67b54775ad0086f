"""Tracking with one model for the batch and every instance its own reference trajectory (copra_batch_set_shared_system +
copra_batch_set_cost_reference on a TrajectoryCost whose reference changes along the horizon), against the per-instance-system mode with the
same references: kernel time per solve at batch 65536 (GPU box)."""
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
ts = np.linspace(0.0, 1.0, N + 1)
xref = workloads.COM_X_INIT[None, :] + ts[:, None] * (workloads.COM_X_GOAL - workloads.COM_X_INIT)[None, :]
costs = [dict(kind="trajectory", M=np.kron(np.eye(N + 1), np.eye(6)), p=xref.reshape(-1), weights=np.tile([10.0, 10, 10, 1, 1, 1], N + 1)), wl["costs"][1]]
refs = np.tile(xref.reshape(-1), (b, 1)) + 0.02 * rng.standard_normal((b, 6 * (N + 1)))
out = []
for mode in ("shared model", "per-instance systems"):
    eng = BatchLMPC(6, 3, N, b, costs, wl["cstrs"])
    if mode == "shared model":
        eng.set_shared_system(A, B, d)
        eng.set_x0(wl["x0"])
    else:
        eng.set_system(np.tile(A, (b, 1, 1)), np.tile(B, (b, 1, 1)), np.tile(d, (b, 1)), wl["x0"])
    eng.set_cost_reference(0, refs)
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
