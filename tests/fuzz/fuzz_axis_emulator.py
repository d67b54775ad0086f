"""Random controllers on the shapes of the (instance, axis)-per-lane solver (tests/random_controllers.py: make_integrator, make_chain3, make_chain1; now and then
in axis-major state order, with a goal per instance) through the CPU wave emulator of the kernel bodies against the oracle: statuses, iteration
counters, U and X.   python tests/fuzz/fuzz_axis_emulator.py [first_seed [count [batch]]]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "emu"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import pyemu  # noqa: E402
import pyoracle  # noqa: E402
import random_controllers as RC  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3))) if a.size else 0.0


first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
b = int(sys.argv[3]) if len(sys.argv) > 3 else 24
nbad = ntie = naxis = 0
for seed in range(first, first + count):
    rng = np.random.default_rng([seed, 5])
    c = RC.make_chain3(seed, b) if seed % 4 == 0 else RC.make_chain1(seed, b) if seed % 4 == 2 else RC.make_integrator(seed, b)
    if c["nu"] == 1:
        continue
    what = list(c["forms"])
    refs = None
    plain_goal = c["costs"][0]["kind"] == "trajectory" and np.asarray(c["costs"][0]["p"]).size == c["nx"]
    if plain_goal and rng.random() < 0.4:
        refs = {0: np.tile(c["costs"][0]["p"], (b, 1)) + 0.1 * rng.standard_normal((b, c["nx"]))}
        what.append("own goals")
    re = pyemu.lmpc_solve(c["A"], c["B"], c["d"], c["x0"], c["N"], c["costs"], c["cstrs"], cost_refs=refs)
    bad = tie = 0
    worst = 0.0
    for k in range(b):
        costs = c["costs"] if refs is None else [dict(c["costs"][0], p=refs[0][k])] + c["costs"][1:]
        ro = pyoracle.lmpc_solve(c["A"][k], c["B"][k], c["d"][k], c["x0"][k], c["N"], costs, c["cstrs"])
        if re["status"][k] != ro["status"]:
            bad += 1
            continue
        if ro["status"] != 0:
            continue
        tie += int(tuple(re["iter"][k]) != tuple(ro["iter"]))
        worst = max(worst, rel(re["control"][k], ro["control"]), rel(re["trajectory"][k], ro["trajectory"]))
    axis = re.get("lane_pass_finished", 0)
    naxis += int(axis > 0)
    flag = "   <<<<<<" if bad or worst > 1e-6 else ("   < counters" if tie else "")
    nbad += int(bool(bad or worst > 1e-6))
    ntie += tie
    print("%4d (%d, %d, %d) %s: ended in the pass / solver %d of %d, status differ %d, counters differ %d, rel %.1e%s" % (seed, c["nx"], c["nu"], c["N"], what, axis, b, bad, tie, worst, flag), flush=True)
print("controllers with a status or value mismatch: %d; instances with other counters: %d; controllers that ran a per-lane kernel: %d" % (nbad, ntie, naxis))
