"""Where the Riccati-factor tier (stage recursion, one wave per instance) loses to the factor-only kernels (dense triangular solves): short
horizons x constraint levels, per-instance systems and shared-model ticks, kernel ms per solve at batch 65536 (GPU box).
Columns: variables | mean iterations | tier ms | factor-only ms (option no_ric / no_ric_shared) | ratio."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from copra_amd import BatchLMPC, workloads  # noqa: E402

b = 65536


def planar(N, v_max, u_max, seed=1, T=0.1):
    rng = np.random.default_rng(seed)
    A = np.tile(np.block([[np.eye(2), T * np.eye(2)], [np.zeros((2, 2)), np.eye(2)]]), (b, 1, 1))
    B = np.tile(np.vstack([0.5 * T * T * np.eye(2), T * np.eye(2)]), (b, 1, 1))
    d = np.zeros((b, 4))
    x0 = np.hstack([rng.normal(0, 0.2, (b, 2)), rng.uniform(-0.2, 0.2, (b, 2))])
    inf = np.inf
    costs = [dict(kind="trajectory", M=np.eye(4), p=np.array([0.45, 0.3, 0.0, 0.0]), weights=[10, 7, 1, 1.5]),
             dict(kind="control", N=np.eye(2), p=np.zeros(2), weights=[1e-3, 2e-3])]
    cstrs = [dict(kind="trajectory_bound", lower=[-inf] * 4, upper=[inf, inf, v_max, 0.85 * v_max]),
             dict(kind="control_bound", lower=[-u_max, -0.9 * u_max], upper=[u_max, 0.8 * u_max])]
    return dict(A=A, B=B, d=d, x0=x0, N=N, costs=costs, cstrs=cstrs)


def run(wl, shared, opts):
    nx, nu = wl["A"].shape[1], wl["B"].shape[2]
    eng = BatchLMPC(nx, nu, wl["N"], b, wl["costs"], wl["cstrs"], options=opts)
    if shared:
        eng.set_shared_system(wl["A"][3], wl["B"][3], wl["d"][3])
        eng.set_x0(wl["x0"])
    else:
        eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    ts = []
    for _ in range(8):
        eng.solve()
        eng.synchronize()
        ts.append(eng.last_solve_seconds())
    r = eng.results()
    eng.close()
    return min(ts) * 1e3, float(r["iter"][:, 0][r["status"] == 0].mean())


cases = []
for N in (5, 8, 10, 12, 16):
    for vm, um in ((0.6, 3.0), (0.35, 1.8), (0.25, 1.2)):
        cases.append(("CoM (6,3) N=%d v_max %.2f" % (N, vm), workloads.com_preview(b, N=N, v_max=vm, u_max=um)))
for N in (6, 8, 12, 16, 24):
    for vm, um in ((0.4, 1.5), (0.2, 0.8), (0.12, 0.5)):
        cases.append(("planar (4,2) N=%d v_max %.2f" % (N, vm), planar(N, vm, um)))
for shared in (False, True):
    print("== shared-model tick" if shared else "== per-instance systems", flush=True)
    for name, wl in cases:
        t1, it = run(wl, shared, None)
        t2, _ = run(wl, shared, dict(no_ric_shared=1) if shared else dict(no_ric=1))
        print("%-32s n=%2d  iterations %5.2f  tier %.3f ms  factor-only %.3f ms  ratio %.2f" % (name, wl["B"].shape[2] * wl["N"], it, t1, t2, t1 / t2), flush=True)
