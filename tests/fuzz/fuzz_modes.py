"""Engine MODES on random controllers (tests/random_controllers.py), device against the oracle on a sample:
   shared    one (A, B, d) for the batch (copra_batch_set_shared_system), cold and with the warm start, three ticks
   refs      per-instance cost references (copra_batch_set_cost_reference) for every cost that has a per-step p
   rhs       per-instance right-hand sides of the row constraints and per-instance control bounds
   ticks     six receding-horizon ticks (x0 <- x_1 of the previous solution): layouts are re-chosen underway
python tests/fuzz/fuzz_modes.py first count"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as oracle  # noqa: E402
import random_controllers as RC  # noqa: E402
from copra_amd import BatchLMPC  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3))) if a.size else 0.0


HARD = [0]  # solves with different statuses or results more than 1e-4 apart

# Every field of copra_options_t (include/copra_hip.h) takes part: the options choose WHICH kernels run, never what they compute, so every
# set below must give the oracle's results.  tests/test_capi_symbols.py asserts that the union of these keys IS the struct's field list --
# an option that is added without a line here fails the CPU suite (round-4 verdict: the variant surface is the main correctness risk).
OPTION_SETS = [
    {},
    dict(lane_min_batch=-1),
    dict(no_lane_pass=1),
    dict(no_lane_handover=1, lane_min_batch=-1),
    dict(no_lane_spec=1, lane_min_batch=-1),
    dict(no_lane_axes=1, lane_min_batch=-1),
    dict(no_axis_solver=1, lane_min_batch=-1),
    dict(no_ric=1),
    dict(no_tri=1),
    dict(ric_general=1, lane_min_batch=-1),
    dict(no_dense_layout=1),
    dict(no_q1regs=1),
    dict(no_ladder=1),
    dict(no_packed=1),
    dict(ric_k=8, lane_min_batch=-1),
    dict(no_ric_shared=1),
    dict(no_riccati=1),
    dict(no_ric_fast=1),
    dict(ric_step_tol=1e-9, ric_mu_tol=1e-12),
    dict(no_stage_refs=1, no_step_rows=1, no_selection_rows=1),
    dict(debug=1, no_lane_spec=1),
]


def option_set(seed):
    """the engine options of a seed's controller: eight consecutive seeds (two of every mode, an integrator shape and a random one each) share a set"""
    return dict(OPTION_SETS[(seed // 8) % len(OPTION_SETS)])


def compare(tag, seed, c, res, refs, picks, info):
    """refs: list of oracle results (one per pick)"""
    st = sum(int(res["status"][k] != r["status"]) for k, r in zip(picks, refs))
    itd = sum(int(tuple(res["iter"][k]) != tuple(r["iter"])) for k, r in zip(picks, refs) if r["status"] == 0 and res["status"][k] == 0)
    ru = max([rel(res["control"][k], r["control"]) for k, r in zip(picks, refs) if r["status"] == 0 and res["status"][k] == 0] or [0.0])
    rx = max([rel(res["trajectory"][k], r["trajectory"]) for k, r in zip(picks, refs) if r["status"] == 0 and res["status"][k] == 0] or [0.0])
    bad = st > 0 or ru > 1e-6 or rx > 1e-6
    hard = st > 0 or ru > 1e-4 or rx > 1e-4  # (between the two: conditioning -- device and oracle on either side of the optimum)
    if bad or itd:
        print(seed, tag, (c["nx"], c["nu"], c["N"]), c["forms"], info, "status differ %d iter differ %d relU %.1e relX %.1e" % (st, itd, ru, rx),
              "  <<<<<<" if hard else "  <" if bad else "", flush=True)
    HARD[0] += int(hard)
    return int(bad)


def run_seed(seed):
    """-> number of mismatching solves of this seed's controller in its mode"""
    nbad = 0
    if True:
        rng = np.random.default_rng([seed, 5])
        integ = seed % 2 == 0
        b = int(rng.choice([512, 4096, 24576])) if integ else 256
        c = RC.make_integrator(seed, b) if integ else RC.make(seed, batch=b)
        nx, nu, N = c["nx"], c["nu"], c["N"]
        picks = np.linspace(0, b - 1, 24).astype(int)
        opts = option_set(seed)
        if integ and seed % 4 == 0:
            opts.setdefault("lane_min_batch", -1)
        mode = ("shared", "refs", "rhs", "ticks")[(seed // 2) % 4]
        try:
            if mode == "shared":
                A0, B0, d0 = c["A"][0], c["B"][0], c["d"][0]
                eng = BatchLMPC(nx, nu, N, b, c["costs"], c["cstrs"], options=opts)
                eng.set_system(c["A"], c["B"], c["d"], c["x0"])
                eng.set_shared_system(A0, B0, d0)
                if seed % 3 == 0:
                    eng.set_warm_start(True)
                x0 = c["x0"].copy()
                for tick in range(3):
                    eng.set_x0(x0)
                    eng.solve()
                    res = eng.results()
                    refs = [oracle.lmpc_solve(A0, B0, d0, x0[k], N, c["costs"], c["cstrs"]) for k in picks]
                    nbad += compare("shared tick %d%s" % (tick, " warm" if seed % 3 == 0 else ""), seed, c, res, refs, picks, eng.layout_info())
                    good = res["status"] == 0
                    x0 = np.where(good[:, None], res["trajectory"][:, nx:2 * nx], x0)
                    x0 = np.nan_to_num(x0)
                eng.close()
            elif mode == "refs":
                eng = BatchLMPC(nx, nu, N, b, c["costs"], c["cstrs"], options=opts)
                eng.set_system(c["A"], c["B"], c["d"], c["x0"])
                own = {}
                for t, cost in enumerate(c["costs"]):
                    p = np.asarray(cost["p"], dtype=float)
                    own[t] = p[None, :] + 0.05 * rng.standard_normal((b, p.size))
                    eng.set_cost_reference(t, own[t])
                eng.solve()
                res = eng.results()
                refs = []
                for k in picks:
                    costs_k = [dict(cost, p=own[t][k]) for t, cost in enumerate(c["costs"])]
                    refs.append(oracle.lmpc_solve(c["A"][k], c["B"][k], c["d"][k], c["x0"][k], N, costs_k, c["cstrs"]))
                nbad += compare("refs", seed, c, res, refs, picks, eng.layout_info())
                eng.close()
            elif mode == "rhs":
                eng = BatchLMPC(nx, nu, N, b, c["costs"], c["cstrs"], options=opts)
                eng.set_system(c["A"], c["B"], c["d"], c["x0"])
                own = {}
                bounds = None
                for t, cs in enumerate(c["cstrs"]):
                    if cs["kind"] in ("trajectory", "control", "mixed") and cs.get("ineq", True):
                        f = np.asarray(cs["f"], dtype=float)
                        own[t] = f[None, :] + 0.2 * rng.random((b, f.size))  # (looser than the controller's: step 0 stays feasible)
                        eng.set_constraint_rhs(t, own[t])
                    elif cs["kind"] == "control_bound":
                        lo = np.broadcast_to(np.asarray(cs["lower"], float).reshape(-1) if np.size(cs["lower"]) == nu * N else np.tile(np.asarray(cs["lower"], float), N), (b, nu * N))
                        hi = np.broadcast_to(np.asarray(cs["upper"], float).reshape(-1) if np.size(cs["upper"]) == nu * N else np.tile(np.asarray(cs["upper"], float), N), (b, nu * N))
                        bounds = (lo - 0.3 * rng.random((b, nu * N)), hi + 0.3 * rng.random((b, nu * N)), t)
                        eng.set_control_bounds(bounds[0], bounds[1])
                eng.solve()
                res = eng.results()
                refs = []
                for k in picks:
                    cs_k = []
                    for t, cs in enumerate(c["cstrs"]):
                        if t in own:
                            cs_k.append(dict(cs, f=own[t][k]))
                        elif bounds is not None and t == bounds[2]:
                            cs_k.append(dict(kind="control_bound", lower=bounds[0][k], upper=bounds[1][k]))
                        else:
                            cs_k.append(cs)
                    refs.append(oracle.lmpc_solve(c["A"][k], c["B"][k], c["d"][k], c["x0"][k], N, c["costs"], cs_k))
                nbad += compare("rhs", seed, c, res, refs, picks, eng.layout_info())
                eng.close()
            else:
                eng = BatchLMPC(nx, nu, N, b, c["costs"], c["cstrs"], options=opts)
                x0 = c["x0"].copy()
                eng.set_system(c["A"], c["B"], c["d"], x0)
                for tick in range(6):
                    eng.set_x0(x0)
                    eng.solve()
                    res = eng.results()
                    refs = [oracle.lmpc_solve(c["A"][k], c["B"][k], c["d"][k], x0[k], N, c["costs"], c["cstrs"]) for k in picks]
                    nbad += compare("tick %d" % tick, seed, c, res, refs, picks, (eng.layout_info(), eng.lane_pass_info()))
                    good = res["status"] == 0
                    x0 = np.nan_to_num(np.where(good[:, None], res["trajectory"][:, nx:2 * nx], x0))
                eng.close()
        except Exception as ex:  # noqa: BLE001
            print(seed, mode, (nx, nu, N), c["forms"], "ERROR", str(ex)[:200], flush=True)
            nbad += 1
            HARD[0] += 1
            try:
                eng.close()
            except Exception:  # noqa: BLE001
                pass
    return nbad


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    total = sum(run_seed(seed) for seed in range(first, first + count))
    print("mismatching solves:", total)
