"""Which instances of a TIGHT jerk-model workload does the (instance, axis)-per-lane solver list for the tiers, and why (emulator, COPRA_EMU_AXIS_REPORT);\ncounters and U against the oracle.   python tools/exp/jerk_emu_probe.py"""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/emu"); sys.path.insert(0, "oracle")
os.environ["COPRA_EMU_AXIS_REPORT"] = "1"
import numpy as np
import pyemu, pyoracle
from copra_amd import workloads
b = 420
wl = workloads.jerk_preview(b, nu=3, N=20, seed=77, v_max=0.3, j_max=6.0)
re = pyemu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"])
print("finished", re["lane_pass_finished"], "of", b, "mean iters", re["iter"][:, 0].mean(), "max", re["iter"][:, 0].max())
ro = pyoracle.lmpc_solve_batch(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"], nthreads=8)
ok = ro["status"] == 0
print("status equal", (re["status"] == ro["status"]).all(), "iters differ on", int((re["iter"][ok] != ro["iter"][ok]).any(axis=1).sum()), "of", int(ok.sum()))
dif = np.where((re["iter"] != ro["iter"]).any(axis=1))[0]
print(dif, re["iter"][dif].tolist(), ro["iter"][dif].tolist())
rel = np.abs(re["control"][ok] - ro["control"][ok]).max(axis=1) / np.maximum(np.abs(ro["control"][ok]).max(axis=1), 1e-3)
print("max rel U", rel.max(), "for listed-before", rel[[22, 39, 106, 156, 172, 184, 187, 373]])
