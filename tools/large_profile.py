"""Phase + active-set sub-phase cycle counts of the workgroup-per-instance kernel on the BASELINE config-5 workload
(profiling build libcopra_hip_prof.so, -DCOPRA_FINE_PROFILE).  GPU box only."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
subprocess.check_call(["make", "-C", os.path.join(ROOT, "copra_amd", "csrc"), "libcopra_hip_prof.so"],
                      stdout=subprocess.DEVNULL)
import torch  # noqa: F401,E402  (loads the HIP runtime torch ships before ours)
from copra_amd import _capi  # noqa: E402

_capi.LIB_PATH = os.path.join(ROOT, "copra_amd", "csrc", "libcopra_hip_prof.so")
_capi.build_library = lambda force=False: False
from copra_amd import BatchLMPC, workloads  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
wl = workloads.long_horizon_initial_state(batch)
ist = wl["initial_state"]
eng = BatchLMPC(12, 6, wl["N"], batch, wl["costs"], wl["cstrs"], initial_state=dict(R=ist["R"], r=ist["r"]))
L = _capi.lib()
L.copra_batch_fine_profile.restype = C.c_int
L.copra_batch_fine_profile.argtypes = [C.c_void_p, C.c_void_p]
eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
eng.set_initial_state_bounds(ist["x0lb"], ist["x0ub"])
eng.enable_phase_profile()
_capi.check(L.copra_batch_fine_profile(eng._h, None))
eng.solve()
eng.solve()
fine = np.zeros((batch, 32), dtype=np.int64)
_capi.check(L.copra_batch_fine_profile(eng._h, fine.ctypes.data))
ph = eng.phase_profile()
res = eng.results()
it = res["iter"]
print("kernel %.1f ms for %d instances; iterations mean %.0f, drops mean %.0f" % (
    eng.last_solve_seconds() * 1e3, batch, it[:, 0].mean(), it[:, 1].mean()))
print("phases (mean cycles per instance):")
for name, v in zip(eng.PHASES, ph.mean(axis=0)):
    print("  %-12s %12.0f" % (name, v))
names = ["scan", "normal", "d=J'n", "z=J2 d2", "r=R^-1 d1", "step", "add (Givens)", "drop", "partial"]
m = fine[:, :len(names)].mean(axis=0)
print("active-set sub-phases (mean cycles per instance, per iteration):")
for name, v in zip(names, m):
    print("  %-14s %12.0f %10.0f" % (name, v, v / it[:, 0].mean()))
