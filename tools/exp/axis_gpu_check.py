"""The one-(instance, axis)-per-lane solver (lmpc_axis.hpp) on the device: parity against the CPU oracle on a sample, rates along the
constraint ladder (tools/exp/tight_ladder.py's workloads), with and without it.
    python tools/exp/axis_gpu_check.py [batch]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402
from copra_amd import BatchLMPC, workloads  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3)))


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    nref = 4096
    for vm, um in ((0.6, 3.0), (0.4, 2.0), (0.35, 1.8), (0.30, 1.5), (0.25, 1.2)):
        wl = workloads.com_preview(b, v_max=vm, u_max=um)
        ref = pyoracle.lmpc_solve_batch(wl["A"][:nref], wl["B"][:nref], wl["d"][:nref], wl["x0"][:nref], wl["N"], wl["costs"], wl["cstrs"], nthreads=16, native=True)
        line = "v_max %.2f u_max %.1f: mean iters %.2f |" % (vm, um, ref["iter"][:, 0].mean())
        for opts in ({}, dict(no_axis_solver=1)):
            eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"], options=opts)
            eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
            ms = []
            for i in range(12):
                eng.solve()
                eng.synchronize()
                ms.append(eng.last_solve_seconds() * 1e3)
            r = eng.results()
            ran, fin = eng.lane_pass_info()
            ok_s = bool((r["status"][:nref] == ref["status"]).all())
            ok_i = bool((r["iter"][:nref] == ref["iter"]).all())
            e = max(rel(r["control"][:nref], ref["control"]), rel(r["trajectory"][:nref], ref["trajectory"]))
            line += " %s: %.3f ms (first %.3f) = %.1f M/s, pass %d finished %d, status %s iters %s rel %.1e |" % (
                "axis" if not opts else "r05 ", min(ms[2:]), ms[0], b / min(ms[2:]) / 1e3, ran, fin, ok_s, ok_i, e)
            eng.close()
        print(line, flush=True)


if __name__ == "__main__":
    main()
