"""Wall-clock step time of the headline workload on a side build of the library (tools/exp/lane_variants.sh), profiling off: python tools/exp/ab_bench.py TAG"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
from copra_amd import _capi  # noqa: E402
_capi.LIB_PATH = os.path.abspath("copra_amd/csrc/variants/libcopra_hip_exp%s.so" % sys.argv[1])
_capi.build_library = lambda force=False: False
from copra_amd import BatchLMPC, workloads  # noqa: E402

b = 65536
wl = workloads.com_preview(b)
eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"])
eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
for _ in range(30):
    eng.solve()
eng.synchronize()
out = []
for rep in range(5):
    t = time.perf_counter()
    for _ in range(50):
        eng.solve()
    eng.synchronize()
    out.append((time.perf_counter() - t) / 50)
print("%s: ms per step %s -> best %.4f" % (sys.argv[1], " ".join("%.4f" % (1e3 * x) for x in out), 1e3 * min(out)))
