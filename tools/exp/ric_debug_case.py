import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from copra_amd import BatchLMPC
src = open(os.path.join(ROOT, "tools/exp/ric_variants.py")).read().split("\nb = 4096")[0]
ns = {"__file__": os.path.join(ROOT, "tools/exp/ric_variants.py")}
exec(compile(src, "rv", "exec"), ns)
dim, N = int(sys.argv[1]), int(sys.argv[2])
nx, nu, A, B, d, x0, costs, cstrs = ns["integrator"](dim, N, 4096, 5, False)
for opts in (dict(ric_k=7, no_ladder=1), dict(ric_k=7, no_ladder=1, ric_general=1), dict(no_ladder=1, ric_general=1)):
    print(opts, flush=True)
    eng = BatchLMPC(nx, nu, N, 4096, costs, cstrs, options=opts)
    eng.set_system(A, B, d, x0)
    eng.solve()
    res = eng.results()
    print("   ", eng.layout_info(), "instance 0: status", res["status"][0], "iter", res["iter"][0].tolist(), "sum|U|", np.abs(res["control"][0]).sum(), flush=True)
    eng.close()
