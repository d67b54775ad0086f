import numpy as np, torch
from copra_amd import BatchLMPC, workloads
b = 65536
wl = workloads.com_preview(b)
goals = workloads.COM_X_GOAL[None, :] + 0.05 * np.random.default_rng(5).standard_normal((b, 6))
same = np.tile(workloads.COM_X_GOAL, (b, 1))
for name, g in (("no refs", None), ("own goals, all equal", same), ("own goals", goals)):
    eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"])
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    if g is not None: eng.set_cost_reference(0, torch.from_numpy(np.ascontiguousarray(g)).cuda())
    for _ in range(8): eng.solve()
    eng.synchronize()
    ts = []
    for _ in range(20):
        eng.solve(); eng.synchronize(); ts.append(eng.last_solve_seconds())
    r = eng.results()
    print("%-22s %.4f ms  mean iters %.3f max %d" % (name, np.median(ts) * 1e3, r["iter"][:, 0].mean(), r["iter"][:, 0].max()))
    eng.close()
