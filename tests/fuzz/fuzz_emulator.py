"""tests/random_controllers.py through the CPU wave emulator (tests/emu) against the oracle, no GPU needed: logic of the kernel bodies on
thousands of random controllers.   python tests/fuzz/fuzz_emulator.py first count [batch] [option=value,...]   (COPRA_EMU_LADDER_STEPS=k in the environment: k steps down the layout ladder first)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyemu  # noqa: E402
import pyoracle as oracle  # noqa: E402
import random_controllers as RC  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3))) if a.size else 0.0


first, count = int(sys.argv[1]), int(sys.argv[2])
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if len(sys.argv) > 4:  # engine options, name=value,...  (which kernel variants run: copra_options_t)
    from copra_amd import _capi
    _capi.OPTIONS.update({k: int(v) for k, v in (kv.split("=") for kv in sys.argv[4].split(","))})
bad = itd = inst = 0
for seed in range(first, first + count):
    integ = seed % 4 == 3
    c = RC.make_integrator(seed, 66) if integ else RC.make(seed, batch=batch)
    ro = oracle.lmpc_solve_batch(c["A"], c["B"], c["d"], c["x0"], c["N"], c["costs"], c["cstrs"], nthreads=1)
    try:
        re = pyemu.lmpc_solve(c["A"], c["B"], c["d"], c["x0"], c["N"], c["costs"], c["cstrs"])
    except Exception as ex:  # noqa: BLE001
        print(seed, (c["nx"], c["nu"], c["N"]), c["forms"], "EMULATOR ERROR", str(ex)[:120], flush=True)
        bad += 1
        continue
    ok = ro["status"] == 0
    same = bool((re["status"] == ro["status"]).all())
    nd = int((re["iter"][ok] != ro["iter"][ok]).any(axis=1).sum())
    ru, rx = rel(re["control"][ok], ro["control"][ok]), rel(re["trajectory"][ok], ro["trajectory"][ok])
    itd += nd
    inst += int(ok.sum())
    if not same or ru > 1e-6 or rx > 1e-6:
        bad += 1
        print(seed, (c["nx"], c["nu"], c["N"]), c["forms"], "status equal", same, "relU %.1e relX %.1e" % (ru, rx), "iter differ", nd, "rcap", re.get("rcap"), "  <<<<<<", flush=True)
    ist = c.get("initial_state")
    if ist is not None and not integ:
        k = 0
        rs = oracle.lmpc_solve(c["A"][k], c["B"][k], c["d"][k], c["x0"][k], c["N"], c["costs"], c["cstrs"],
                               initial_state=dict(R=ist["R"], r=ist["r"], x0lb=ist["x0lb"][k], x0ub=ist["x0ub"][k]))
        es = pyemu.lmpc_solve(c["A"][[k]], c["B"][[k]], c["d"][[k]], c["x0"][[k]], c["N"], c["costs"], c["cstrs"],
                              initial_state=dict(R=ist["R"], r=ist["r"], x0lb=ist["x0lb"][[k]], x0ub=ist["x0ub"][[k]]))
        if es["status"][0] != rs["status"] or (rs["status"] == 0 and (rel(es["control"][0], rs["control"]) > 1e-6 or rel(es["trajectory"][0], rs["trajectory"]) > 1e-6)):
            bad += 1
            print(seed, "InitialStateLMPC", (c["nx"], c["nu"], c["N"]), c["forms"], "status", es["status"][0], rs["status"],
                  "relU %.1e" % (rel(es["control"][0], rs["control"]) if rs["status"] == 0 else 0.0), "  <<<<<<", flush=True)
print("seeds %d..%d: %d mismatching, iteration counters differ on %d of %d solved instances" % (first, first + count - 1, bad, itd, inst))
