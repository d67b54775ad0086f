"""one controller of tests/random_controllers.py on the device under several engine options, against the oracle"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as oracle  # noqa: E402
import random_controllers as RC  # noqa: E402
from copra_amd import BatchLMPC  # noqa: E402

seed = int(sys.argv[1])
mv = int(sys.argv[2]) if len(sys.argv) > 2 else (72 if seed % 5 == 0 else 64)
c = RC.make(seed, batch=48, max_vars=mv)
r = oracle.lmpc_solve_batch(c["A"], c["B"], c["d"], c["x0"], c["N"], c["costs"], c["cstrs"], nthreads=8)
print(seed, (c["nx"], c["nu"], c["N"]), c["forms"], "oracle status", np.bincount(r["status"]))
for opts in (None, dict(no_riccati=1), dict(no_ric_fast=1), dict(debug=1)):
    eng = BatchLMPC(c["nx"], c["nu"], c["N"], 48, c["costs"], c["cstrs"], options=opts)
    eng.set_system(c["A"], c["B"], c["d"], c["x0"])
    eng.solve()
    e = eng.results()
    info = eng.layout_info()
    eng.close()
    ru = np.max(np.abs(e["control"] - r["control"]) / np.maximum(np.abs(r["control"]), 1e-3), axis=1)
    rx = np.max(np.abs(e["trajectory"] - r["trajectory"]) / np.maximum(np.abs(r["trajectory"]), 1e-3), axis=1)
    worst = np.argsort(-ru)[:5]
    print(opts, info, "status", np.bincount(e["status"]), "relU max %.2e relX max %.2e" % (ru.max(), rx.max()))
    for k in worst:
        print("    instance %d relU %.2e relX %.2e device iter %s oracle iter %s" % (k, ru[k], rx[k], e["iter"][k].tolist(), r["iter"][k].tolist()))
