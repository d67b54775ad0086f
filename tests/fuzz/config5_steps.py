"""BASELINE config 5 on the device: histogram of the Newton steps, and the instances the interior-point kernel handed to the Goldfarb-Idnani
kernel (iteration counter far above the others)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from copra_amd import BatchLMPC, workloads  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
wl = workloads.long_horizon_initial_state(b)
ist = wl["initial_state"]
eng = BatchLMPC(12, 6, wl["N"], b, wl["costs"], wl["cstrs"], initial_state=dict(R=ist["R"], r=ist["r"]))
eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
eng.set_initial_state_bounds(ist["x0lb"], ist["x0ub"])
for _ in range(2):
    eng.solve()
res = eng.results()
print("device time %.1f ms" % (eng.last_solve_seconds() * 1e3), "status", np.bincount(res["status"], minlength=4))
it = res["iter"][:, 0]
print("Newton steps: mean %.2f histogram" % it[it <= 40].mean(), np.bincount(np.minimum(it, 41))[:42].tolist())
far = np.where(it > 40)[0]
print("finished by the Goldfarb-Idnani kernel:", far.tolist()[:50], "count", len(far))
eng.close()
