"""Headline shape through the RUN-TIME-shape kernel (a zero-weight seventh cost row keeps it off the compile-time
instantiation): what the specialisation is worth (dev tool; GPU box only)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from copra_amd import BatchLMPC, workloads  # noqa: E402

b = 65536
wl = workloads.com_preview(b)
c0 = wl["costs"][0]
generic = [dict(kind="trajectory", M=np.vstack([c0["M"], np.zeros((1, 6))]), p=np.append(c0["p"], 0.0),
                weights=np.append(c0["weights"], 0.0)), wl["costs"][1]]
import time  # noqa: E402
for name, costs, jit in (("compile-time shape <6,3,20,6>", wl["costs"], False), ("run-time shape (7 cost rows)", generic, False),
                         ("7 cost rows, copra_batch_specialise", generic, True)):
    eng = BatchLMPC(6, 3, wl["N"], b, costs, wl["cstrs"])
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    if jit:
        t0 = time.time()
        eng.specialise()
        print("  specialise(): %.1f s" % (time.time() - t0))
        t0 = time.time()
        eng.specialise()
        e2 = BatchLMPC(6, 3, wl["N"], 8, costs, wl["cstrs"])
        e2.specialise()
        print("  again (cached): %.2f s" % (time.time() - t0))
    ts = []
    for _ in range(8):
        eng.solve()
        eng.synchronize()
        ts.append(eng.last_solve_seconds())
    res = eng.results()
    print("%-36s %.3f ms  %.2f M solves/s  ok %d" % (name, np.mean(ts[4:]) * 1e3, b / np.mean(ts[4:]) / 1e6,
                                                     int((res["status"] == 0).sum())))
    if name.startswith("run-time"):
        u_generic = res["control"]
    if jit:
        print("  max |u_specialised - u_generic| = %.2e" % np.nanmax(np.abs(res["control"] - u_generic)))
