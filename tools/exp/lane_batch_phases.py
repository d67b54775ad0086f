"""Per-wave phase cycles of the one-instance-per-lane pass at several batch sizes (GPU box): does a wave's time depend on how many others run?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from copra_amd import BatchLMPC, workloads  # noqa: E402

for b in (4096, 16384, 32768, 65536, 131072):
    wl = workloads.com_preview(b)
    eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"])
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    for _ in range(3):
        eng.solve()
    eng.enable_phase_profile(True)
    eng.solve()
    eng.solve()
    eng.synchronize()
    pr = eng.phase_profile()
    nw = (b + 63) // 64
    pw = pr[:nw]
    print("batch %6d: solve %.4f ms | per wave: staging %6.0f sweep %6.0f roll-out %6.0f verdict %5.0f total %6.0f cycles"
          % (b, eng.last_solve_seconds() * 1e3, pw[:, 0].mean(), pw[:, 1].mean(), pw[:, 2].mean(), pw[:, 3].mean(), pw[:, 7].mean()))
    del eng
