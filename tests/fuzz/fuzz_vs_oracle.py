"""Random controllers (tests/random_controllers.py) on the device against the oracle: statuses, U, X (entry-wise, floor 1e-3), iteration
counters.  python tests/fuzz/fuzz_vs_oracle.py [first_seed] [count] [batch]"""
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


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3))) if a.size else 0.0


first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 48
bad = itdiff = inst = 0
for seed in range(first, first + count):
    c = RC.make(seed, batch=batch, max_vars=72 if seed % 5 == 0 else 64)
    r = oracle.lmpc_solve_batch(c["A"], c["B"], c["d"], c["x0"], c["N"], c["costs"], c["cstrs"], nthreads=8)
    try:
        eng = BatchLMPC(c["nx"], c["nu"], c["N"], batch, c["costs"], c["cstrs"])
        eng.set_system(c["A"], c["B"], c["d"], c["x0"])
        eng.solve()
        e = eng.results()
        info = eng.layout_info()
        eng.close()
    except Exception as ex:  # noqa: BLE001
        print(seed, (c["nx"], c["nu"], c["N"]), c["forms"], "ENGINE ERROR", ex, flush=True)
        bad += 1
        continue
    ok = r["status"] == 0
    same_st = bool((e["status"] == r["status"]).all())
    nd = int((e["iter"][ok] != r["iter"][ok]).any(axis=1).sum()) if c["nu"] * c["N"] <= 64 else 0  # (solved instances on the Goldfarb-Idnani kernels)
    ru, rx = rel(e["control"][ok], r["control"][ok]), rel(e["trajectory"][ok], r["trajectory"][ok])
    itdiff += nd
    inst += int(ok.sum()) if c["nu"] * c["N"] <= 64 else 0
    if not same_st or ru > 1e-6 or rx > 1e-6:
        bad += 1
        print(seed, (c["nx"], c["nu"], c["N"]), c["forms"], "status equal", same_st, "relU %.2e relX %.2e" % (ru, rx), "iter differ", nd,
              "oracle iters %.1f" % r["iter"][:, 0].mean(), info, "  <<<<<<", flush=True)
    elif nd:
        print(seed, (c["nx"], c["nu"], c["N"]), c["forms"], "iteration counters differ on", nd, "relU %.1e" % ru, flush=True)
print("seeds %d..%d: %d mismatching controllers, iteration counters differ on %d of %d instances" % (first, first + count - 1, bad, itdiff, inst))
