"""The one-instance-per-lane pass with 64 and with 32 instances per wave (copra_options_t::lane_group) at batches around a shard of
BASELINE configs[3]: device time of a solve (pair of launches) and of the pass alone (headline workload; the pass forced on below its threshold)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401,E402
from copra_amd import BatchLMPC, workloads  # noqa: E402

for b in (8192, 16384, 24576, 32768, 49152, 65536):
    wl = workloads.com_preview(b, seed=2)
    row = []
    for grp in (64, 32, 0):
        eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"], options=dict(lane_group=grp, lane_min_batch=-1))
        eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
        for _ in range(5):
            eng.solve()
        eng.synchronize()
        ts, tp = [], []
        for _ in range(8):
            eng.solve()
            ts.append(eng.last_solve_seconds())
            tp.append(eng.last_first_tier_seconds())
        row.append((grp, min(ts) * 1e3, b / min(ts) / 1e6, eng.lane_pass_info()))
        eng.close()
    print("batch %6d: " % b + " | ".join("group %2d: %.3f ms %6.1f M/s (pass ran %s, finished %d)" % (g, ms, r, i[0], i[1]) for g, ms, r, i in row), flush=True)
# the tier alone, for the threshold
for b in (8192, 16384, 24576, 32768):
    wl = workloads.com_preview(b, seed=2)
    eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"], options=dict(no_lane_pass=1))
    eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
    for _ in range(5):
        eng.solve()
    eng.synchronize()
    ts = []
    for _ in range(8):
        eng.solve()
        ts.append(eng.last_solve_seconds())
    print("batch %6d: tier alone %.3f ms %6.1f M/s" % (b, min(ts) * 1e3, b / min(ts) / 1e6), flush=True)
    eng.close()
