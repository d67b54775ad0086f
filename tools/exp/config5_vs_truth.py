"""HIP path vs the 60-digit truth vectors of config 5 at R = 1e-6 (tests/golden/config5_truth.npz)"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from copra_amd import BatchLMPC, workloads
T = np.load(os.path.join(ROOT, "tests", "golden", "config5_truth.npz"))
b = int(T["batch"])
wl = workloads.long_horizon_initial_state(b, R_diag=float(T["r_diag"]))
ist = wl["initial_state"]
eng = BatchLMPC(12, 6, wl["N"], b, wl["costs"], wl["cstrs"], initial_state=dict(R=ist["R"], r=ist["r"]))
eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
eng.set_initial_state_bounds(ist["x0lb"], ist["x0ub"])
eng.solve()
res = eng.results()
x0o = eng.initial_state()
for k in T["instances"]:
    k = int(k)
    ut, xt = T["control_%d" % k], T["trajectory_%d" % k]
    eu = np.abs(res["control"][k] - ut) / (1 + np.abs(ut))
    ex = np.abs(res["trajectory"][k] - xt) / (1 + np.abs(xt))
    print("instance %d: status %d iter %s  HIP vs truth: U %.3e  X %.3e  x0 %.3e   (oracle vs truth U %.3e)"
          % (k, res["status"][k], tuple(res["iter"][k]), eu.max(), ex.max(), np.abs(x0o[k] - T["x0_opt_%d" % k]).max(),
             float(T["oracle_err_control_%d" % k])))
