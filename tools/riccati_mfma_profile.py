"""Per-phase shader-clock profile of the LDS-resident Riccati interior-point kernel (lmpc_riccati_mfma.hpp) on BASELINE config 5
(run on the GPU box).  Stamp slots of that kernel: set-up | rows (bulk, before the sweep) | sweep 1 | the two x0 steps | the two forward sweeps |
sweep 3 | the other bulk phases (predictor rows, corrector coefficients, final rows, update) | total."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fine = "--lib" in sys.argv  # a build with -DCOPRA_RF_FINE: the stamps sit inside the stage of sweep 1
if fine:
    from copra_amd import _capi
    _capi.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
    _capi.build_library = lambda force=False: False
    del sys.argv[sys.argv.index("--lib"):sys.argv.index("--lib") + 2]
from copra_amd import BatchLMPC, workloads  # noqa: E402

fine2 = "--fine2" in sys.argv
if fine2:
    sys.argv.remove("--fine2")
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
it = res["iter"][:, 0].mean()
print("solver", eng.solver(), "kernel ms", eng.last_solve_seconds() * 1e3, "batch", batch, "iters mean", it)
names = ("setup", "rows A", "sweep 1", "x0 step x2", "forward x2", "sweep 3", "rows B+C+upd", "total")
if fine:
    names = ("-> stage top", "H reads, P hand-over", "T (18 MFMA)", "prepare next", "M (21 MFMA)", "eliminate u_b", "eliminate u_a", "total")
    if fine2:
        names = ("stage up to T issued", "prepare: class check", "prepare: touched entries", "prepare: gradient", "M .. end of stage", "-", "-", "total")
    print("FINE build: cycles per stage of sweep 1 (stamps drain the pipelines: sums exceed the undisturbed stage)")
    for k, name in enumerate(names[:7]):
        print("%-22s %8.0f cycles per stage" % (name, pr[:, k].mean() / it / 51))
    sys.exit(0)
for k, name in enumerate(names):
    print("%-12s mean %12.0f cycles  (%5.1f %%)  per Newton step %9.0f" % (name, pr[:, k].mean(), 100.0 * pr[:, k].mean() / pr[:, 7].mean(),
                                                                          pr[:, k].mean() / it))
print("per stage visit (51 stages): sweep 1 %.0f, forward (each of two) %.0f, sweep 3 %.0f cycles; x0 step (each of two) %.0f"
      % (pr[:, 2].mean() / it / 51, pr[:, 4].mean() / it / 102, pr[:, 5].mean() / it / 51, pr[:, 3].mean() / it / 2))
