"""Kernel rate over problem sizes (dev / report tool; GPU box only): double integrator (nx=2, nu=1) and CoM preview
(nx=6, nu=3) at several horizons.  Env: COPRA_OPTIONS=no_packed=1, COPRA_OPTIONS=no_dense_layout=1 switch the small / mid-size
mappings off for comparison."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from copra_amd import BatchLMPC, workloads  # noqa: E402


def rate(eng, batch):
    ts = []
    for _ in range(7):
        eng.solve()
        eng.synchronize()
        ts.append(eng.last_solve_seconds())
    return batch / float(np.mean(ts[3:])) / 1e6


JIT = "--specialise" in sys.argv  # copra_batch_specialise for every shape (about 20 s each, cached)
rows = []
for N in (5, 10, 16, 20, 32, 48, 64):
    b = 131072
    wl = workloads.double_integrator(b, N=N)
    eng = BatchLMPC(2, 1, N, b, wl["costs"], wl["cstrs"])
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    if JIT:
        eng.specialise()
    rows.append(("double integrator N=%d" % N, N, eng.lanes_per_instance(), rate(eng, b)))
for N in (5, 8, 10, 12, 15, 18, 20, 21):
    b = 65536
    wl = workloads.com_preview(b, N=N)
    eng = BatchLMPC(6, 3, N, b, wl["costs"], wl["cstrs"])
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    if JIT:
        eng.specialise()
    rows.append(("CoM preview N=%d" % N, 3 * N, eng.lanes_per_instance(), rate(eng, b)))
for name, n, lanes, r in rows:
    print("%-26s n=%3d  lanes/instance %3d  %8.2f M solves/s" % (name, n, lanes, r))
# (nx, nu) = (4, 2): the planar point mass of tests/test_gpu_parity.py
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
try:
    import test_gpu_parity as T
    for N in (16, 24, 32):
        b = 65536
        wl = T._planar_integrator(b, N)
        eng = BatchLMPC(4, 2, N, b, wl["costs"], wl["cstrs"])
        eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
        if JIT:
            eng.specialise()
        print("%-26s n=%3d  lanes/instance %3d  %8.2f M solves/s" % ("planar point mass N=%d" % N, 2 * N, eng.lanes_per_instance(), rate(eng, b)))
except Exception as e:  # (the sweep itself must not be lost)
    print("planar cases skipped:", repr(e))
# chains of three states per control (the jerk-controlled CoM model): shapes outside the double-integrator families -- the library's own builds
# (round 6: the (instance, axis)-per-lane solver) and, with --specialise, the run-time-compiled general kernels of the same shape
for nu, N in ((3, 20), (3, 12), (2, 20), (2, 10)):
    b = 65536
    wl = workloads.jerk_preview(b, nu=nu, N=N)
    for opts, what in ((None, ""), (dict(no_axis_solver=1), " without the axis solver")):
        eng = BatchLMPC(3 * nu, nu, N, b, wl["costs"], wl["cstrs"], options=opts)
        eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
        if JIT and opts:
            eng.specialise()
        print("%-46s n=%3d  lanes/instance %3d  %8.2f M solves/s" % ("jerk model (%d, %d) N=%d%s" % (3 * nu, nu, N, what), nu * N, eng.lanes_per_instance(), rate(eng, b)))
        eng.close()
