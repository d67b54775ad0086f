"""tests/random_controllers.py::make_integrator (third argument "chain3" / "chain1": make_chain3 / make_chain1 -- chains of three states, of one state per control) on the device against the
oracle (sample), over many seeds: which layouts fail?   python tests/fuzz/fuzz_integrators.py first count [chain3 | chain1]"""
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

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    b = (24576, 4096, 6144)[seed % 3]
    gen = sys.argv[3] if len(sys.argv) > 3 else ""
    c = RC.make_chain3(seed, b) if gen == "chain3" else RC.make_chain1(seed, b) if gen == "chain1" else RC.make_integrator(seed, b)
    eng = BatchLMPC(c["nx"], c["nu"], c["N"], b, c["costs"], c["cstrs"], options=dict(lane_min_batch=-1) if b == 6144 else None)
    eng.set_system(c["A"], c["B"], c["d"], c["x0"])
    pick = np.linspace(0, b - 1, 160).astype(int)
    ref = oracle.lmpc_solve_batch(c["A"][pick], c["B"][pick], c["d"][pick], c["x0"][pick], c["N"], c["costs"], c["cstrs"], nthreads=8)
    ok = ref["status"] == 0
    line = []
    for rep in range(3):  # (the layout may change between the solves: prediction, then the sizes of the final active sets)
        eng.solve()
        res = eng.results()
        info, lane = eng.layout_info(), eng.lane_pass_info()
        st = int((res["status"][pick] != ref["status"]).sum())
        itd = int((res["iter"][pick][ok] != ref["iter"][ok]).any(axis=1).sum())
        both = ok & (res["status"][pick] == 0)
        ru = float(np.max(np.abs(res["control"][pick][both] - ref["control"][both]) / np.maximum(np.abs(ref["control"][both]), 1e-3))) if both.any() else 0.0
        line.append("%d B/%d cols%s pass %s: status %d iter %d relU %.0e" % (info["lds_bytes"], info["active_capacity"], "" if info["factor_only"] else " (square)", lane[0], st, itd, ru))
        if st or itd or ru > 1e-6:
            bad += 1
    eng.close()
    flag = "  <<<<<<" if any(("status 0 iter 0" not in x) or float(x.split("relU ")[1]) > 1e-6 for x in line) else ""
    print(seed, (c["nx"], c["nu"], c["N"]), b, c["forms"], " | ".join(line), flag, flush=True)
print("mismatching solves:", bad)
