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
which = sys.argv[2] if len(sys.argv) > 2 else "config5"
if which == "config5":
    wl = workloads.long_horizon_initial_state(batch)
    ist = wl["initial_state"]
    eng = BatchLMPC(12, 6, wl["N"], batch, wl["costs"], wl["cstrs"], initial_state=dict(R=ist["R"], r=ist["r"]))
else:  # the reference's 300-step falling-mass fixture (tests/fixtures.py)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fixtures as F
    pb = F.bounded_system("trajectory", N=300)
    x0 = np.tile(pb["x0"], (batch, 1))
    x0[:, 1] += np.random.default_rng(0).uniform(-0.5, 0.5, batch)
    wl = dict(A=np.tile(pb["A"], (batch, 1, 1)), B=np.tile(pb["B"], (batch, 1, 1)), d=np.tile(pb["d"], (batch, 1)),
              x0=x0, N=300)
    ist = None
    eng = BatchLMPC(2, 1, 300, batch, pb["costs"], pb["cstrs"])
L = _capi.lib()
L.copra_batch_fine_profile.restype = C.c_int
L.copra_batch_fine_profile.argtypes = [C.c_void_p, C.c_void_p]
eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
if ist is not None:
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
# where / when each instance ran: HW_ID (CU id bits 8-11, SH 12, SE 13-15 on gfx9), wall clock at 100 MHz
hw, t0, t1 = fine[:, 28], fine[:, 29], fine[:, 30]
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (((hw >> 4) & 0xF) << 8)
print("distinct (xcc? se, sh, cu) ids: %d;  instance wall time mean %.1f ms" % (len(np.unique(cu)), (t1 - t0).mean() / 1e5))
worst = 0
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    ev = sorted([(t0[i], 1) for i in idx] + [(t1[i], -1) for i in idx])
    live = peak = 0
    for _, d in ev:
        live += d
        peak = max(peak, live)
    worst = max(worst, peak)
print("max instances in flight on one CU id at the same time: %d; span of the launch %.1f ms" % (
    worst, (t1.max() - t0.min()) / 1e5))
