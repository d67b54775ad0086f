import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from copra_amd import BatchLMPC, workloads
from copra_amd.batch import to_abi_layout
dev = torch.device("cuda", 0)
b = 65536
for vmax, umax in ((0.25, 1.2), (0.3, 1.5), (0.35, 1.75), (0.4, 2.0), (0.6, 3.0)):
    wl = workloads.com_preview(b, v_max=vmax, u_max=umax)
    Ab, Bb, db, xb = to_abi_layout(wl["A"], wl["B"], wl["d"], wl["x0"])
    t = [torch.from_numpy(a).to(dev) for a in (Ab, Bb, db, xb)]
    eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"])
    eng.set_system(*t)
    hist = []
    for k in range(8):
        eng.solve(); eng.synchronize()
        hist.append((eng.layout_info()["active_capacity"], eng.layout_info()["lds_bytes"], round(eng.last_solve_seconds() * 1e3, 3)))
    it = eng.results()["iter"][:, 0]
    best = min(h[2] for h in hist[4:])
    print("v_max %.2f u_max %.1f: mean iters %.2f  %.2f M solves/s  ladder (capacity, LDS bytes, ms): %s" % (vmax, umax, it.mean(), b / best / 1e3, hist))
    eng.close()
