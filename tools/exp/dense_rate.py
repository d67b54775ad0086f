"""The dense Psi' W Psi path (full-size cost entries on v_mfma_f64_16x16x4): rate at batch 65 536 and a parity sample"""
import sys
sys.path.insert(0, "oracle")
import numpy as np
import pyoracle
from copra_amd import BatchLMPC, workloads
from copra_amd.autospan import autospan_cost
b = 65536
wl = workloads.com_preview(b)
c0 = wl["costs"][0]
costs = [autospan_cost(dict(c0, p=np.tile(c0["p"], wl["N"] + 1))), wl["costs"][1]]
eng = BatchLMPC(6, 3, wl["N"], b, costs, wl["cstrs"], options=dict(no_stage_refs=1))
eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
for _ in range(4): eng.solve()
eng.synchronize()
ts = []
for _ in range(8):
    eng.solve(); eng.synchronize(); ts.append(eng.last_solve_seconds())
r = eng.results()
bad = 0
for k in range(0, b, 4099):
    ro = pyoracle.lmpc_solve(wl["A"][k], wl["B"][k], wl["d"][k], wl["x0"][k], wl["N"], wl["costs"], wl["cstrs"])
    bad += int(r["status"][k] != ro["status"] or tuple(r["iter"][k]) != tuple(ro["iter"]) or np.abs(r["control"][k] - ro["control"]).max() > 1e-6 * max(1e-3, np.abs(ro["control"]).max()))
print("dense path: %.2f M solves/s (%.3f ms), layout %s, sample mismatches %d" % (b / np.median(ts) / 1e6, np.median(ts) * 1e3, eng.layout_info(), bad))
