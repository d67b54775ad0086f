"""PCIe-inclusive rate of the headline workload: inputs handed over as HOST buffers, results fetched to the host
(copra_batch_set_system(on_device = 0) + solve + get_results), vs the device-resident figure bench.py reports."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from copra_amd import BatchLMPC, workloads  # noqa: E402

batch = 65536
wl = workloads.com_preview(batch)
eng = BatchLMPC(6, 3, wl["N"], batch, wl["costs"], wl["cstrs"])
ts = []
for _ in range(6):
    t0 = time.perf_counter()
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])  # numpy -> layout conversion + H2D
    eng.solve()
    res = eng.results()  # D2H
    ts.append(time.perf_counter() - t0)
t = float(np.mean(ts[2:]))
print("host-inclusive: %.2f ms per %d solves -> %.2f M solves/s (kernel alone %.2f ms)" % (
    t * 1e3, batch, batch / t / 1e6, eng.last_solve_seconds() * 1e3))
