"""Kernel time of the (instance, axis)-per-lane solver against the batch (how many waves run side by side), sustained: GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from copra_amd import BatchLMPC, workloads  # noqa: E402
opts = {}
for waves in (192, 768, 1023, 2046, 3069):
    b = waves * 64 // 3 // 64 * 64
    wl = workloads.com_preview(b, v_max=10.0, u_max=100.0)  # nothing is violated: sweep + roll-out + results only, no tier behind it
    eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"], options=opts)
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    ms = []
    for i in range(400):
        eng.solve()
        if i >= 300:
            eng.synchronize(); ms.append(eng.last_solve_seconds() * 1e3)
    eng.enable_phase_profile(True)
    eng.solve(); eng.solve()
    pr = eng.phase_profile()
    nw = (b * 3 + 63) // 64
    print("waves %5d batch %6d: median %.4f ms min %.4f | ticks per wave %d" % (nw, b, np.median(ms), min(ms), pr[:nw, 7].mean()), "phases (load, sweep, roll-out, iteration, results):", pr[:nw, :5].mean(axis=0).astype(int).tolist(), flush=True)
    eng.close()
