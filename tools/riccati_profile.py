"""Per-phase shader-clock profile of the Riccati interior-point kernel on BASELINE config 5 (run on the GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from copra_amd import BatchLMPC, workloads  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
wl = workloads.long_horizon_initial_state(batch)
ist = wl["initial_state"]
eng = BatchLMPC(12, 6, wl["N"], batch, wl["costs"], wl["cstrs"], initial_state=dict(R=ist["R"], r=ist["r"]))
eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
eng.set_initial_state_bounds(ist["x0lb"], ist["x0ub"])
eng.solve()
print("no profile: kernel %.4f s -> %.0f solves/s" % (eng.last_solve_seconds(), batch / eng.last_solve_seconds()))
eng.enable_phase_profile(True)
eng.solve()
pr = eng.phase_profile()
res = eng.results()
print("solver", eng.solver(), "kernel ms", eng.last_solve_seconds() * 1e3, "batch", batch, "iters mean", res["iter"][:, 0].mean())
names = ("setup", "s1 rows", "s1 gradient", "s1 factor", "forward x2", "s3 backward", "update", "total")
for k, name in enumerate(names):
    print("%-12s mean %12.0f cycles  (%5.1f %%)" % (name, pr[:, k].mean(), 100.0 * pr[:, k].mean() / pr[:, 7].mean()))
it = res["iter"][:, 0].mean()
print("per iteration and stage visit (51 stages): s1 rows %.0f, gradient %.0f, factor %.0f, forward (each of two) %.0f, s3 %.0f cycles"
      % tuple(pr[:, k].mean() / it / 51 / d for k, d in ((1, 1), (2, 1), (3, 1), (4, 2), (5, 1))))
