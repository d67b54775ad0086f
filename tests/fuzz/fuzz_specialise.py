"""Random controllers with copra_batch_specialise (run-time compiled kernels for the controller's shape: hipcc on this box) against the
oracle -- shapes the library has no compile-time instantiation for, incl. the Riccati-factor tier where take_ric_layout grants it.
python tests/fuzz/fuzz_specialise.py first count"""
import os
import sys
import tempfile
import time

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


first, count = int(sys.argv[1]), int(sys.argv[2])
cache = tempfile.mkdtemp(prefix="copra_jit_")
bad = 0
for seed in range(first, first + count):
    b = 4096
    c = RC.make(seed, batch=b) if seed % 2 else RC.make_integrator(seed, b)
    pick = np.linspace(0, b - 1, 96).astype(int)
    ref = oracle.lmpc_solve_batch(c["A"][pick], c["B"][pick], c["d"][pick], c["x0"][pick], c["N"], c["costs"], c["cstrs"], nthreads=8)
    ok = ref["status"] == 0
    out = []
    for spec in (False, True):
        eng = BatchLMPC(c["nx"], c["nu"], c["N"], b, c["costs"], c["cstrs"])
        t0 = time.time()
        if spec:
            try:
                eng.specialise(cache)
            except Exception as ex:  # noqa: BLE001
                out.append("specialise: %s" % str(ex)[:80])
                eng.close()
                continue
        dt = time.time() - t0
        eng.set_system(c["A"], c["B"], c["d"], c["x0"])
        for rep in range(3):
            eng.solve()
        res = eng.results()
        sec = eng.last_solve_seconds()
        info = eng.layout_info()
        eng.close()
        st = int((res["status"][pick] != ref["status"]).sum())
        both = ok & (res["status"][pick] == 0)
        itd = int((res["iter"][pick][both] != ref["iter"][both]).any(axis=1).sum())
        ru = rel(res["control"][pick][both], ref["control"][both])
        flag = "OK " if (st == 0 and ru <= 1e-6) else ("~  " if st == 0 and ru <= 1e-4 else "BAD")
        bad += flag == "BAD"
        out.append("%s %s%dB/%d %s st %d it %d relU %.0e %.2f ms%s" % ("jit" if spec else "lib", "ric " if info["factor_only"] else "", info["lds_bytes"], info["active_capacity"], flag, st, itd, ru, sec * 1e3,
                                                                    " (compiled in %.0f s)" % dt if spec else ""))
    print(seed, (c["nx"], c["nu"], c["N"]), c["forms"], " | ".join(out), flush=True)
print("mismatching:", bad)
