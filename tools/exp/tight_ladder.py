import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from copra_amd import BatchLMPC, workloads
b = 65536
wl = workloads.com_preview(b, v_max=0.25, u_max=1.2)
eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"])
eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
for i in range(6):
    eng.solve(); eng.synchronize() if hasattr(eng, "synchronize") else None
    r = eng.results()
    print(i, "ms", eng.last_solve_seconds() * 1e3, "layout", eng.layout_info() if hasattr(eng, "layout_info") else "", "ok", int((r["status"] == 0).sum()))
