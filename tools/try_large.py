"""Kernel time of the workgroup-per-instance path on the 300-step reference fixture (dev tool; GPU box only).
usage: try_large.py BATCH [system] [xcost]   env: COPRA_OPTIONS=debug=1"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import fixtures as F  # noqa: E402
from copra_amd import BatchLMPC  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 512
system = sys.argv[2] if len(sys.argv) > 2 else "bounded"
xcost = sys.argv[3] if len(sys.argv) > 3 else "trajectory"
pb = getattr(F, system + "_system")(xcost, N=300)
rng = np.random.default_rng(0)
x0 = np.tile(pb["x0"], (b, 1))
x0[:, 1] += rng.uniform(-0.5, 0.5, b)
eng = BatchLMPC(2, 1, 300, b, pb["costs"], pb["cstrs"])
eng.set_system(np.tile(pb["A"], (b, 1, 1)), np.tile(pb["B"], (b, 1, 1)), np.tile(pb["d"], (b, 1)), x0)
for _ in range(3):
    eng.solve()
    eng.results()
    t = eng.last_solve_seconds()
res = eng.results()
print("batch %d: kernel %.2f ms -> %.0f solves/s; iterations mean %.1f drops %.1f; status %s" % (
    b, t * 1e3, b / t, res["iter"][:, 0].mean(), res["iter"][:, 1].mean(), np.bincount(res["status"], minlength=4)))
