"""Shared-model receding-horizon tick rate next to the headline path on the same workload shape:
CoM preview (nx=6, nu=3, N=20), ONE (A, B, d) for the whole batch, per-instance x0.  GPU box only (dev / report tool;
bench.py keeps the BASELINE metric, where A and B differ per instance)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from copra_amd import BatchLMPC, workloads  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
wl = workloads.com_preview(batch)
A, B, d = wl["A"][0], wl["B"][0], wl["d"][0]
out = {}
for mode in ("per-instance system (fused kernel)", "shared model (factorised once)"):
    eng = BatchLMPC(6, 3, wl["N"], batch, wl["costs"], wl["cstrs"])
    if mode.startswith("shared"):
        eng.set_shared_system(A, B, d)
        eng.set_x0(wl["x0"])
    else:
        eng.set_system(np.tile(A, (batch, 1, 1)), np.tile(B, (batch, 1, 1)), np.tile(d, (batch, 1)), wl["x0"])
    ts = []
    for _ in range(steps + 3):
        eng.solve()
        eng.synchronize()
        ts.append(eng.last_solve_seconds())
    res = eng.results()
    t = float(np.mean(ts[3:]))
    out[mode] = {"kernel_ms": t * 1e3, "solves_per_s": batch / t, "solved_ok": int((res["status"] == 0).sum()),
                 "mean_iters": float(res["iter"][:, 0].mean())}
    if mode.startswith("shared"):
        u_shared = res["control"]
    else:
        u_ref = res["control"]
out["max_abs_u_diff"] = float(np.nanmax(np.abs(u_shared - u_ref)))
out["batch"] = batch
print(json.dumps(out))
