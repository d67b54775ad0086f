import sys
sys.path.insert(0, "tests")
import random_controllers as RC
from copra_amd import BatchLMPC
for seed in range(500, 536):
    c = RC.make_chain3(seed, 2048)
    eng = BatchLMPC(c["nx"], c["nu"], c["N"], 2048, c["costs"], c["cstrs"])
    eng.set_system(c["A"], c["B"], c["d"], c["x0"])
    eng.solve()
    print(seed, (c["nx"], c["nu"], c["N"]), c["forms"], "axis" if eng.axis_solver_ran() else "----", eng.lane_pass_info(), eng.layout_info().get("lds_bytes"), eng.lanes_per_instance())
    eng.close()
