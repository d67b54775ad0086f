"""The new paths of the (instance, axis)-per-lane solver through the UBSan build of the CPU emulator (make -C tests/emu ubsan):\nUBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python tools/exp/emu_ubsan_axis.py"""
import os, sys, ctypes as C
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/emu"); sys.path.insert(0, "oracle")
import numpy as np
import pyemu, pyoracle
L = C.CDLL(os.path.abspath("tests/emu/libcopra_emu_ubsan.so"))
for f in ("emu_lmpc_solve", "emu_qp_dense", "emu_lmpc_solve_shared", "emu_lmpc_solve_riccati"):
    getattr(L, f).restype = C.c_int
pyemu._lib = L
from copra_amd import workloads
import random_controllers as RC
def check(name, wl, **kw):
    re = pyemu.lmpc_solve(wl["A"], wl["B"], wl["d"], wl["x0"], wl["N"], wl["costs"], wl["cstrs"], **kw)
    print(name, "finished", re.get("lane_pass_finished"), "status ok", int((re["status"] == 0).sum()), flush=True)
b = 45
check("com", workloads.com_preview(b, v_max=0.3, u_max=1.5, seed=3))
check("com N=21", workloads.com_preview(b, N=21, v_max=0.4, u_max=2.0, seed=4))
check("com axis-major", workloads.axis_major(workloads.com_preview(b, v_max=0.4, u_max=2.0, seed=5)))
check("jerk", workloads.jerk_preview(b, nu=3, N=20, seed=6, v_max=0.3, j_max=6.0))
check("jerk nu=2 two rows", workloads.jerk_preview(40, nu=2, N=14, seed=7, v_max=0.3, j_max=6.0, a_max=1.5))
wl = workloads.com_preview(b, v_max=0.5, u_max=2.5, seed=8)
goals = wl["costs"][0]["p"][None, :] + 0.2 * np.random.default_rng(1).standard_normal((b, 6))
check("goals", wl, cost_refs={0: goals})
N = wl["N"]
ts = np.linspace(0, 1, N + 1)
xref = workloads.COM_X_INIT[None, :] + ts[:, None] * (workloads.COM_X_GOAL - workloads.COM_X_INIT)[None, :]
track = dict(kind="trajectory", M=np.kron(np.eye(N + 1), np.eye(6)), p=xref.reshape(-1), weights=np.tile([10.0, 10, 10, 1, 1, 1], N + 1))
check("tracking", dict(wl, costs=[track, wl["costs"][1]]))
own = np.tile(xref.reshape(-1), (b, 1)) + 0.03 * np.random.default_rng(2).standard_normal((b, xref.size))
check("tracking own", dict(wl, costs=[track, wl["costs"][1]]), cost_refs={0: own})
vl = 0.5 * np.random.default_rng(3).uniform(0.6, 1.3, b)
ul = 2.5 * np.random.default_rng(4).uniform(0.6, 1.3, b)
check("limits", wl, row_rhs=np.repeat(vl[:, None], 3 * (N + 1), axis=1), bounds=(-np.repeat(ul[:, None], 3 * N, axis=1), np.repeat(ul[:, None], 3 * N, axis=1)))
for seed in (1, 4, 9, 12):
    c = RC.make_chain3(seed, 24)
    check("chain3 %d %s" % (seed, c["forms"]), c)
print("done")
