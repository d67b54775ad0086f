"""One seed of tests/fuzz/fuzz_integrators.py with and without the pass's speculative steps: which sampled instances differ from the oracle in
their iteration counters, and how.   python tests/fuzz/debug_spec_seed.py SEED [SEED ...]"""
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

for seed in [int(a) for a in sys.argv[1:]]:
    b = (24576, 4096, 6144)[seed % 3]
    c = RC.make_integrator(seed, b)
    pick = np.linspace(0, b - 1, 160).astype(int)
    ref = oracle.lmpc_solve_batch(c["A"][pick], c["B"][pick], c["d"][pick], c["x0"][pick], c["N"], c["costs"], c["cstrs"], nthreads=8)
    ok = ref["status"] == 0
    for spec in (1, 0):
        opts = dict(lane_min_batch=-1, no_lane_spec=0 if spec else 1)
        eng = BatchLMPC(c["nx"], c["nu"], c["N"], b, c["costs"], c["cstrs"], options=opts)
        eng.set_system(c["A"], c["B"], c["d"], c["x0"])
        eng.solve()
        res = eng.results()
        d = np.flatnonzero((res["iter"][pick] != ref["iter"]).any(axis=1) & ok)
        print("seed %d (%d, %d, %d) %s spec %d pass %s: %d of %d sampled instances differ in their counters" % (seed, c["nx"], c["nu"], c["N"], c["forms"], spec, eng.lane_pass_info(), len(d), int(ok.sum())))
        for j in d[:6]:
            k = pick[j]
            ru = float(np.max(np.abs(res["control"][k] - ref["control"][j]) / np.maximum(np.abs(ref["control"][j]), 1e-3)))
            print("     instance %d: device %s oracle %s relU %.1e" % (k, tuple(res["iter"][k]), tuple(ref["iter"][j]), ru))
        eng.close()
