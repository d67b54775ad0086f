import os, sys
sys.path.insert(0, os.getcwd())
from copra_amd import BatchLMPC, workloads
b = 65536
wt = workloads.com_preview(b, v_max=0.25, u_max=1.2, seed=5)
eng = BatchLMPC(6, 3, wt["N"], b, wt["costs"], wt["cstrs"])
eng.set_system(wt["A"], wt["B"], wt["d"], wt["x0"])
for k in range(6):
    eng.solve(); s = eng.last_solve_seconds()
    print("solve", k, "first-tier ms %.3f" % (s * 1e3), eng.layout_info(), flush=True)
