"""Phase split of the dense Psi' W Psi path (what BASELINE configs[2] names: the TrajectoryCost handed over as a full-size entry, the
Hessian built by the v_mfma_f64_16x16x4 contraction of lmpc_fused.hpp::full_size_cost_term): per-instance shader-clock cycles of
preview | costs (= the contraction) | norms | Cholesky | x0 | active set | results, and the kernel time.  Run on the GPU box."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from copra_amd import BatchLMPC, workloads  # noqa: E402
from copra_amd import _capi as _capi_opts  # noqa: E402

_capi_opts.OPTIONS["no_lane_pass"] = 1  # (the phases of the tier ALONE: under the phase profile the pass otherwise runs as in production)
from copra_amd.autospan import autospan_cost  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = workloads.com_preview(batch)
c0 = wl["costs"][0]
costs = [autospan_cost(dict(c0, p=np.tile(c0["p"], wl["N"] + 1))), wl["costs"][1]]
eng = BatchLMPC(6, 3, wl["N"], batch, costs, wl["cstrs"], options=dict(no_stage_refs=1))
eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
for _ in range(3):
    eng.solve()
eng.synchronize()
best = min((eng.solve(), eng.last_solve_seconds())[1] for _ in range(3))
print("dense path: %.3f ms per %d = %.2f M solves/s, layout %s" % (best * 1e3, batch, batch / best / 1e6, eng.layout_info()))
eng.enable_phase_profile(True)
eng.solve()
eng.solve()
pr = eng.phase_profile()
print("with stamps: %.3f ms" % (eng.last_solve_seconds() * 1e3))
tot = pr[:, 7].mean()
for k, name in enumerate(BatchLMPC.PHASES):
    print("%-12s mean %10.0f cycles  %5.1f %%" % (name, pr[:, k].mean(), 100.0 * pr[:, k].mean() / tot))
mf = 976 * 64
print("MFMA issue floor of the contraction: 976 x 64 = %d cycles = %.1f %% of its phase" % (mf, 100.0 * mf / pr[:, 1].mean()))
