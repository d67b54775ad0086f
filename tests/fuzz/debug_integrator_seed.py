"""one controller of tests/random_controllers.py::make_integrator on the device under several engine options, against the oracle"""
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

seed, b = int(sys.argv[1]), int(sys.argv[2])
c = RC.make_integrator(seed, b)
pick = np.arange(b) if b <= 4096 else np.linspace(0, b - 1, 2048).astype(int)
r = oracle.lmpc_solve_batch(c["A"][pick], c["B"][pick], c["d"][pick], c["x0"][pick], c["N"], c["costs"], c["cstrs"], nthreads=8)
print(seed, (c["nx"], c["nu"], c["N"]), c["forms"], "oracle status", np.bincount(r["status"], minlength=3), "iters max", r["iter"][:, 0].max())
for opts in (None, dict(no_ladder=1), dict(no_lane_pass=1), dict(no_lane_handover=1), dict(no_ric=1), dict(ric_general=1)):
    eng = BatchLMPC(c["nx"], c["nu"], c["N"], b, c["costs"], c["cstrs"], options=opts)
    eng.set_system(c["A"], c["B"], c["d"], c["x0"])
    for rep in range(3):
        eng.solve()
        e = eng.results()
        info, lane = eng.layout_info(), eng.lane_pass_info()
        bad = np.where(e["status"][pick] != r["status"])[0]
        ok = (r["status"] == 0) & (e["status"][pick] == 0)
        ru = np.max(np.abs(e["control"][pick][ok] - r["control"][ok]) / np.maximum(np.abs(r["control"][ok]), 1e-3)) if ok.any() else 0.0
        itd = int((e["iter"][pick][ok] != r["iter"][ok]).any(axis=1).sum())
        print("  ", opts, "solve", rep, info, "pass", lane, "device status", np.bincount(e["status"], minlength=4), "mismatching statuses", len(bad),
              "relU %.1e iter differ %d" % (ru, itd))
        if len(bad):
            k = bad[0]
            print("       first: sample %d = instance %d device status %d iter %s oracle iter %s" % (k, pick[k], e["status"][pick[k]], e["iter"][pick[k]].tolist(), r["iter"][k].tolist()))
    eng.close()
