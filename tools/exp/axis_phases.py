"""Phase cycles of the one-(instance, axis)-per-lane solver (lmpc_axis.hpp) on the GPU box, one row per wave:
load | sweep | roll-out + first scan | active-set iteration | results.   python tools/exp/axis_phases.py [v_max u_max]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from copra_amd import BatchLMPC, workloads  # noqa: E402

b = 65536
vm = float(sys.argv[1]) if len(sys.argv) > 1 else 0.6
um = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
wl = workloads.com_preview(b, v_max=vm, u_max=um)
eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"])
eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
for _ in range(3):
    eng.solve()
eng.synchronize()
plain = eng.last_solve_seconds() * 1e3
eng.enable_phase_profile(True)
eng.solve()
eng.solve()
pr = eng.phase_profile()
ran, finished = eng.lane_pass_info()
nw = 3072 if b == 65536 else (b + 20) // 21
pw = pr[:nw]
print("v_max %.2f: solve %.3f ms (with stamps %.3f), finished here %d of %d, waves %d" % (vm, plain, eng.last_solve_seconds() * 1e3, finished, b, nw))
for k, name in enumerate(("load", "sweep", "roll-out", "iteration", "results")):
    print("  %-10s mean %9.0f  min %9.0f  max %9.0f cycles per wave" % (name, pw[:, k].mean(), pw[:, k].min(), pw[:, k].max()))
print("  %-10s mean %9.0f  min %9.0f  max %9.0f" % ("total", pw[:, 7].mean(), pw[:, 7].min(), pw[:, 7].max()))

hw = pw[:, 6].astype(np.int64)
xcc = (hw >> 32) & 0xf
hwid = hw & 0xffffffff
simd = (hwid >> 4) & 3
cu = (hwid >> 8) & 0xf
sh = (hwid >> 12) & 1
se = (hwid >> 13) & 7
slot = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("  distinct XCCs %d, distinct CUs %d, distinct (CU, SIMD) %d" % (len(np.unique(xcc)), len(np.unique(slot)), len(np.unique(slot * 4 + simd))))
print("  waves per SIMD id:", np.bincount(simd, minlength=4).tolist(), "| waves per CU: min %d max %d" % (np.bincount(np.unique(slot, return_inverse=True)[1]).min(), np.bincount(np.unique(slot, return_inverse=True)[1]).max()))

# ---- how full the SIMDs are: per CU (the counters of different CUs do not share a base), the span from its first wave's start to its last
#      wave's end against the sum of its waves' own times over the four SIMDs ----
start = pw[:, 5].astype(np.int64)
total = pw[:, 7].astype(np.int64)
spans, busys, gaps = [], [], []
for cu_id in np.unique(slot):
    m = slot == cu_id
    t0 = start[m].min()
    span = (start[m] + total[m]).max() - t0
    spans.append(span)
    busys.append(total[m].sum() / float(span * 4))
    for sd in range(4):  # idle ticks between one wave's end and the next wave's start on the same SIMD
        ms = m & (simd == sd)
        o = np.argsort(start[ms])
        st, en = start[ms][o], (start[ms] + total[ms])[o]
        gaps.extend((st[1:] - en[:-1]).tolist())
spans, busys, gaps = np.array(spans), np.array(busys), np.array(gaps)
print("  per CU: span mean %.0f min %d max %d ticks | SIMDs busy mean %.2f min %.2f | gap between waves on a SIMD: mean %.0f median %.0f max %d ticks"
      % (spans.mean(), spans.min(), spans.max(), busys.mean(), busys.min(), gaps.mean(), np.median(gaps), gaps.max()))
