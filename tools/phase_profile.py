"""Per-phase shader-clock profile of the fused kernel on the headline workload (run on the GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:  # an experimental build of the library (copra_amd/csrc/variants/*.so)
    from copra_amd import _capi  # noqa: E402

    _capi.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
    _capi.build_library = lambda force=False: False
from copra_amd import BatchLMPC, workloads  # noqa: E402
from copra_amd import _capi as _capi_opts  # noqa: E402

_capi_opts.OPTIONS["no_lane_pass"] = 1  # (the phases of the tier ALONE: under the phase profile the pass otherwise runs as in production)

batch = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 65536
wl = workloads.com_preview(batch, v_max=float(os.environ.get("VMAX", "0.6")), u_max=float(os.environ.get("UMAX", "3.0")))  # (VMAX=0.25 UMAX=1.2: the tight variant)
if "--generic" in sys.argv:  # a zero-weight seventh cost row keeps the shape off the compile-time instantiation
    c0 = wl["costs"][0]
    wl["costs"] = [dict(kind="trajectory", M=np.vstack([c0["M"], np.zeros((1, 6))]), p=np.append(c0["p"], 0.0),
                        weights=np.append(c0["weights"], 0.0)), wl["costs"][1]]
eng = BatchLMPC(6, 3, wl["N"], batch, wl["costs"], wl["cstrs"])
eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
for _ in range(int(os.environ.get("WARM_SOLVES", "0"))):  # (lets the layout ladder settle before the profiled solve)
    eng.solve()
eng.enable_phase_profile(True)
eng.solve()
eng.solve()
pr = eng.phase_profile()
res = eng.results()
print("kernel ms", eng.last_solve_seconds() * 1e3, "batch", batch)
for k, name in enumerate(BatchLMPC.PHASES):
    print("%-12s mean %10.0f  p50 %10.0f  max %10.0f cycles" % (name, pr[:, k].mean(), np.median(pr[:, k]), pr[:, k].max()))
it = res["iter"][:, 0]
print("layout:", eng.layout_info())
for v in range(1, int(it.max()) + 1):
    sel = it == v
    if sel.any():
        print("iters=%d: %6d instances, active_set mean %8.0f cycles, drops mean %.2f" % (v, sel.sum(), pr[sel, 5].mean(), res["iter"][sel, 1].mean()))
