import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from copra_amd import BatchLMPC, workloads
for b in (1, 4, 64, 1024):
    wl = workloads.com_preview(b)
    eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"])
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    for _ in range(20): eng.solve(); eng.results()
    t1, t2, tw = [], [], []
    for _ in range(200):
        t0 = time.perf_counter(); eng.solve(); eng.results(); tw.append(time.perf_counter() - t0)
        t1.append(eng.last_first_tier_seconds()); t2.append(eng.last_solve_seconds())
    print("batch %5d: first tier %.1f us, whole solve %.1f us (device), wall solve+results %.1f us" % (b, np.median(t1)*1e6, np.median(t2)*1e6, np.median(tw)*1e6))
    eng.close()
