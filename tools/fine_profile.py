"""Fine-grained in-kernel stamps (profiling build libcopra_hip_prof.so, -DCOPRA_FINE_PROFILE).  GPU box only."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
subprocess.check_call(["make", "-C", os.path.join(ROOT, "copra_amd", "csrc"), "libcopra_hip_prof.so"],
                      stdout=subprocess.DEVNULL)
from copra_amd import _capi  # noqa: E402

_capi.LIB_PATH = os.path.join(ROOT, "copra_amd", "csrc", "libcopra_hip_prof.so")
_capi.build_library = lambda force=False: False
from copra_amd import BatchLMPC, workloads  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
wl = workloads.com_preview(batch)
eng = BatchLMPC(6, 3, wl["N"], batch, wl["costs"], wl["cstrs"])
L = _capi.lib()
L.copra_batch_fine_profile.restype = C.c_int
L.copra_batch_fine_profile.argtypes = [C.c_void_p, C.c_void_p]
eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
_capi.check(L.copra_batch_fine_profile(eng._h, None))
eng.solve()
eng.solve()
out = np.zeros((batch, 32), dtype=np.int64)
_capi.check(L.copra_batch_fine_profile(eng._h, out.ctypes.data))
it = eng.results()["iter"][:, 0]
want = int(sys.argv[2]) if len(sys.argv) > 2 else int(np.bincount(it).argmax())
sel = it == want
m = out[sel].mean(axis=0)
print("kernel ms", eng.last_solve_seconds() * 1e3, "instances with iters ==", want, sel.sum())
prev = 0.0
for k in range(32):
    if m[k] < 0:
        break
    print("stamp %2d  at %9.0f   delta %9.0f" % (k, m[k], m[k] - prev))
    prev = m[k]
