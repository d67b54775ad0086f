"""Random controllers at (nx, nu) = (12, 6) -- the shape of the LDS-resident interior-point kernel on the matrix cores (BASELINE config 5) --
with random horizons 11..24, on the device against the oracle (and as InitialStateLMPC where the generator draws one)."""
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


first, count, batch = int(sys.argv[1]), int(sys.argv[2]), 24
NX, NU = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (12, 6)  # (other shapes: the streaming interior-point kernel)
NMIN = (64 // NU) + 1
bad = 0
for seed in range(first, first + count):
    N = NMIN + seed % 14
    c = RC.make(seed, batch=batch, shape=(NX, NU, N))
    ist = c["initial_state"]
    for variant in ("lmpc", "initial-state") if ist is not None else ("lmpc",):
        if variant == "lmpc":
            r = oracle.lmpc_solve_batch(c["A"], c["B"], c["d"], c["x0"], N, c["costs"], c["cstrs"], nthreads=8)
            eng = BatchLMPC(NX, NU, N, batch, c["costs"], c["cstrs"])
            eng.set_system(c["A"], c["B"], c["d"], c["x0"])
            ks = range(batch)
        else:
            ks = range(0, batch, 6)
            rs = [oracle.lmpc_solve(c["A"][k], c["B"][k], c["d"][k], c["x0"][k], N, c["costs"], c["cstrs"],
                                    initial_state=dict(R=ist["R"], r=ist["r"], x0lb=ist["x0lb"][k], x0ub=ist["x0ub"][k])) for k in ks]
            r = dict(status=np.array([x["status"] for x in rs]), control=np.array([x["control"] for x in rs]), trajectory=np.array([x["trajectory"] for x in rs]))
            eng = BatchLMPC(NX, NU, N, batch, c["costs"], c["cstrs"], initial_state=dict(R=ist["R"], r=ist["r"]))
            eng.set_system(c["A"], c["B"], c["d"], c["x0"])
            eng.set_initial_state_bounds(ist["x0lb"], ist["x0ub"])
        eng.solve()
        e = eng.results()
        eng.close()
        ks = np.array(list(ks))
        ok = r["status"] == 0
        same = bool((e["status"][ks] == r["status"]).all())
        ru, rx = rel(e["control"][ks][ok], r["control"][ok]), rel(e["trajectory"][ks][ok], r["trajectory"][ok])
        newton = e["iter"][ks][ok][:, 0]
        flag = "" if (same and ru <= 1e-6 and rx <= 1e-6) else "   <<<<<<"
        if flag and same:  # further than 1e-6 from the CPU path: the certified optimum decides, on the instance where they are furthest apart
            import truth
            d = np.array([max(rel(e["control"][k], r["control"][j]), rel(e["trajectory"][k], r["trajectory"][j])) if ok[j] else 0.0 for j, k in enumerate(ks)])
            j = int(np.argmax(d))
            k = int(ks[j])
            io = None if variant == "lmpc" else dict(R=ist["R"], r=ist["r"], x0lb=ist["x0lb"][k], x0ub=ist["x0ub"][k])
            zg = r["control"][j] if io is None else np.concatenate([rs[j]["x0_opt"], r["control"][j]])
            t = truth.solve(c["A"][k], c["B"][k], c["d"][k], c["x0"][k], N, c["costs"], c["cstrs"], zg, initial_state=io)
            dev = max(rel(e["control"][k], t["control"]), rel(e["trajectory"][k], t["trajectory"]))
            ora = max(rel(r["control"][j], t["control"]), rel(r["trajectory"][j], t["trajectory"]))
            flag = "   (instance %d against the certified optimum: device %.1e, CPU path %.1e)%s" % (k, dev, ora, "" if dev <= 1e-6 else "   <<<<<<")
        bad += flag.endswith("<<<<<<")
        print(seed, variant, N, c["forms"], "status", np.bincount(r["status"], minlength=3).tolist(), "equal", same, "relU %.1e relX %.1e" % (ru, rx),
              "device iterations %.0f..%.0f" % ((newton.min(), newton.max()) if newton.size else (0, 0)), flag, flush=True)
print("mismatching:", bad)
