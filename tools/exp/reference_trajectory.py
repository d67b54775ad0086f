"""Tracking a reference TRAJECTORY (GPU box): the headline workload with its state cost written as the reference's API demands for a
reference that changes along the horizon -- a full-size TrajectoryCost, M = blkdiag(I .. I), stacked p.  Default: recognised as a per-step
entry with the step's reference; COPRA_OPTIONS=no_stage_refs=1: the dense contraction of a full-size entry (the previous path)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from copra_amd import BatchLMPC, workloads  # noqa: E402

b = 65536
wl = workloads.com_preview(b)
N = wl["N"]
ts = np.linspace(0.0, 1.0, N + 1)
xref = workloads.COM_X_INIT[None, :] + ts[:, None] * (workloads.COM_X_GOAL - workloads.COM_X_INIT)[None, :]
costs = [dict(kind="trajectory", M=np.kron(np.eye(N + 1), np.eye(6)), p=xref.reshape(-1), weights=np.tile([10.0, 10, 10, 1, 1, 1], N + 1)), wl["costs"][1]]
if len(sys.argv) > 1 and sys.argv[1] == "profile":  # rocprofv3 --kernel-trace --stats -- python3 tools/exp/reference_trajectory.py profile
    eng = BatchLMPC(6, 3, N, b, costs, wl["cstrs"])
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    for _ in range(25):
        eng.solve()
    eng.synchronize()
    print("lane pass", eng.lane_pass_info())
    eng.close()
    sys.exit(0)
out = {}
MODES = {"dense contraction": {"no_stage_refs": 1}, "per-step, factor-only tier": {"no_ric": 1},
         "per-step, tier's own sweep": {"no_lane_pass": 1}, "per-step with p_k": {}, "own reference per instance": {}}
rng = np.random.default_rng(4)
for mode, env in MODES.items():
    eng = BatchLMPC(6, 3, N, b, costs, wl["cstrs"], options=env)
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    if mode == "own reference per instance":
        eng.set_cost_reference(0, np.tile(xref.reshape(-1), (b, 1)) + 0.002 * rng.standard_normal((b, 6 * (N + 1))))
    ts_ = []
    for _ in range(10):
        eng.solve()
        eng.synchronize()
        ts_.append(eng.last_solve_seconds())
    out[mode] = (eng.results(), float(np.mean(ts_[5:])), dict(eng.layout_info(), lane_pass=eng.lane_pass_info()))
    eng.close()
r0, r1 = out["dense contraction"][0], out["per-step with p_k"][0]
ok = r0["status"] == 0
for mode in out:
    print("%-28s %.4f ms (%.1f M solves/s), layout %s" % (mode, out[mode][1] * 1e3, b / out[mode][1] / 1e6, out[mode][2]))
print("status equal", (r0["status"] == r1["status"]).all(), "iter equal", (r0["iter"][ok] == r1["iter"][ok]).all(), "max rel |dU|",
      np.abs(r0["control"][ok] - r1["control"][ok]).max() / np.abs(r0["control"][ok]).max(), "mean iterations", r1["iter"][:, 0].mean())
