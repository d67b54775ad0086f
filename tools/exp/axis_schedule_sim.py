"""Would ordering the solver's groups by the PREVIOUS solve's times help?  Takes the per-wave ticks of one headline solve (phase stamps) and
plays list scheduling on 1024 SIMDs: groups in their natural order (what the dispatcher does), longest first by this solve's own times (the
bound), and longest first by the times of a solve with slightly different initial states (what a controller would know)."""
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from copra_amd import BatchLMPC, workloads  # noqa: E402

b = 65536
wl = workloads.com_preview(b)


def wave_ticks(x0):
    eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"])
    eng.set_system(wl["A"], wl["B"], wl["d"], x0)
    for _ in range(3):
        eng.solve()
    eng.synchronize()
    eng.enable_phase_profile(True)
    eng.solve()
    eng.solve()
    pr = eng.phase_profile()[:3072]
    eng.close()
    return pr[:, 7].astype(np.int64)


def makespan(t, order, slots=1024, gap=4000):
    h = [0] * slots
    heapq.heapify(h)
    for g in order:
        s = heapq.heappop(h)
        heapq.heappush(h, s + gap + int(t[g]))
    return max(h)


t = wave_ticks(wl["x0"])
rng = np.random.default_rng(1)
t_prev = wave_ticks(wl["x0"] + 0.01 * rng.standard_normal(wl["x0"].shape))  # the "previous tick": states a little different
nat = np.arange(len(t))
print("waves %d, mean %.0f min %d max %d ticks; sum / 1024 = %.0f" % (len(t), t.mean(), t.min(), t.max(), t.sum() / 1024.0))
print("natural order            : makespan %d" % makespan(t, nat))
print("longest first (own times): makespan %d" % makespan(t, np.argsort(-t)))
print("longest first (previous) : makespan %d   (correlation of the two solves' times %.2f)" % (makespan(t, np.argsort(-t_prev)), np.corrcoef(t, t_prev)[0, 1]))
q = np.argsort(-(t_prev // 4096))  # 4096-tick buckets, as a counting sort would leave them
print("longest first (previous, 4096-tick buckets): makespan %d" % makespan(t, q))
