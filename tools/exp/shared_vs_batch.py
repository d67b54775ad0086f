"""One model for the batch (copra_batch_set_shared_system) against the same systems handed over instance by instance (what the
(instance, axis)-per-lane solver takes): per batch size, with the controller-wide goal and with a goal per instance."""
import sys
import numpy as np, torch
from copra_amd import BatchLMPC, workloads

def rate(eng, b):
    for _ in range(8):
        eng.solve()
    eng.synchronize()
    ts = []
    for _ in range(20):
        eng.solve(); eng.synchronize(); ts.append(eng.last_solve_seconds())
    return b / np.median(ts) / 1e6, np.median(ts) * 1e3

for b in (256, 1024, 4096, 16384, 65536):
    wl = workloads.com_preview(b)
    A, B, d = wl["A"][0], wl["B"][0], wl["d"][0]
    goals = workloads.COM_X_GOAL[None, :] + 0.05 * np.random.default_rng(5).standard_normal((b, 6))
    x0 = torch.from_numpy(np.ascontiguousarray(wl["x0"])).cuda()
    for own in (False, True):
        sh = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"], options=dict(no_axis_solver=1))
        sh.set_shared_system(A, B, d)
        sh.set_x0(x0)
        if own: sh.set_cost_reference(0, torch.from_numpy(goals).cuda())
        r_sh = rate(sh, b)
        sh.close()
        pi = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"])
        pi.set_system(np.tile(A, (b, 1, 1)), np.tile(B, (b, 1, 1)), np.tile(d, (b, 1)), wl["x0"])
        if own: pi.set_cost_reference(0, torch.from_numpy(goals).cuda())
        r_pi = rate(pi, b)
        ran = pi.axis_solver_ran()
        pi.close()
        print("batch %6d %-22s shared-model path %7.1f M solves/s (%.4f ms) | instance by instance %7.1f M (%.4f ms) axis solver %s" % (b, "goal per instance" if own else "one goal", r_sh[0], r_sh[1], r_pi[0], r_pi[1], ran))
