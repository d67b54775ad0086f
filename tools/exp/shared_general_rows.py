"""Shared-model tick on the headline shape with general rows next to the bounds (tests/random_controllers.py: com_preview_with_general_rows):
the Riccati-factor tier in shared-model mode against lmpc_shared.hpp (option no_ric_shared), kernel time per solve at batch 65536 (GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import random_controllers as RC  # noqa: E402
from copra_amd import BatchLMPC  # noqa: E402

b = 65536
for seed in (0, 1, 2):
    wl, cstrs = RC.com_preview_with_general_rows(seed, b)
    A, B, d = wl["A"][3], wl["B"][3], wl["d"][3]
    line = []
    res = []
    for opts in (None, dict(no_ric_shared=1)):
        eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], cstrs, options=opts)
        eng.set_shared_system(A, B, d)
        eng.set_x0(wl["x0"])
        ts = []
        for _ in range(8):
            eng.solve()
            eng.synchronize()
            ts.append(eng.last_solve_seconds())
        res.append(eng.results())
        line.append("%s %.3f ms (%.1f M solves/s) %s" % ("records" if opts is None else "lmpc_shared.hpp", min(ts) * 1e3, b / min(ts) / 1e6, eng.layout_info()))
        eng.close()
    ok = (res[0]["status"] == 0) & (res[1]["status"] == 0)
    print(("dense state rows", "mixed row", "control row")[seed % 3], "| iterations %.2f |" % res[0]["iter"][:, 0].mean(), " | ".join(line),
          "| status equal", bool((res[0]["status"] == res[1]["status"]).all()), "iter equal", bool((res[0]["iter"][ok] == res[1]["iter"][ok]).all()),
          "max |dU| %.1e" % np.abs(res[0]["control"][ok] - res[1]["control"][ok]).max(), flush=True)
