"""Plug-in point 1 (copra_qp_solve_dense_batch == QuadProgDenseSolver::SI_solve, src/QuadProgSolver.cpp:45-72) on random dense QPs against
the oracle: per seed one (n, meq, mineq) and a batch of problems around a feasible point, with the awkward cases mixed in -- duplicated and
scaled rows (linearly dependent constraints), rows that contradict each other (infeasible), pinned variables (XL == XU), infinite bounds,
inequality rows of zero norm, a Hessian that is not positive definite.  Statuses, both iteration counters, x.
python tests/fuzz/fuzz_dense_qp.py first count [emu]        (emu: the kernel body in the CPU wave emulator, n <= 64, no GPU)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as oracle  # noqa: E402


def make(seed, emu=False):
    """-> (dict of stacked arrays, list of per-problem tags)"""
    rng = np.random.default_rng([seed, 4242])
    big = (not emu) and rng.random() < 0.15
    n = int(rng.integers(65, 200)) if big else int(rng.integers(2, 65))
    meq = int(rng.integers(0, max(1, n // 3) + 1)) if rng.random() < 0.6 else 0
    mi = int(rng.integers(0, 2 * n + 1)) if rng.random() < 0.85 else 0
    b = 6 if big else 16
    out = dict(Q=[], c=[], Aeq=[], beq=[], Aineq=[], bineq=[], XL=[], XU=[])
    tags = []
    for _ in range(b):
        M = rng.standard_normal((n, n))
        Q = M @ M.T / n + float(rng.choice([0.5, 0.05, 5.0])) * np.eye(n)
        c = rng.standard_normal(n) * float(rng.choice([1.0, 10.0]))
        xf = 0.2 * rng.standard_normal(n)
        Aeq = rng.standard_normal((meq, n))
        Ain = rng.standard_normal((mi, n))
        beq = Aeq @ xf
        bin_ = Ain @ xf + 0.3 * rng.random(mi)
        XL = xf - 0.5 * rng.random(n) - 0.05
        XU = xf + 0.5 * rng.random(n) + 0.05
        tag = []
        r = rng.random()
        if r < 0.12 and mi >= 2:  # a duplicated and a scaled row: linearly dependent constraints
            Ain[1] = Ain[0]
            bin_[1] = bin_[0]
            if mi >= 3:
                Ain[2] = 2.5 * Ain[0]
                bin_[2] = 2.5 * bin_[0] + 0.01
            tag.append("dependent")
        elif r < 0.22 and mi >= 2:  # two rows that contradict each other
            Ain[1] = -Ain[0]
            bin_[1] = -bin_[0] - 0.5
            tag.append("contradiction")
        elif r < 0.30:  # pinned variables
            k = rng.integers(0, n, max(1, n // 6))
            XU[k] = XL[k]
            tag.append("pinned")
        elif r < 0.38:  # no bounds at all on some variables, the double maximum on others
            k = rng.integers(0, n, max(1, n // 3))
            XL[k] = -np.inf
            XU[k] = np.inf
            k2 = rng.integers(0, n, max(1, n // 5))
            XU[k2] = np.finfo(float).max
            tag.append("unbounded")
        elif r < 0.44 and mi >= 1:  # a row of zero norm, satisfied or violated
            Ain[0] = 0.0
            bin_[0] = float(rng.choice([0.3, -0.3]))
            tag.append("zero-row")
        elif r < 0.50:  # not positive definite
            w, V = np.linalg.eigh(Q)
            w[0] = -0.1
            Q = (V * w) @ V.T
            tag.append("not-pd")
        elif r < 0.56 and meq >= 2:  # a duplicated equality
            Aeq[1] = Aeq[0]
            beq[1] = beq[0]
            tag.append("dependent-eq")
        elif r < 0.62:  # box far from the unconstrained minimiser: many active bounds
            XL = xf + 0.3
            XU = xf + 0.3 + 0.2 * rng.random(n)
            beq = Aeq @ (xf + 0.35)
            tag.append("far-box")
        for k, v in zip(("Q", "c", "Aeq", "beq", "Aineq", "bineq", "XL", "XU"), (Q, c, Aeq, beq, Ain, bin_, XL, XU)):
            out[k].append(v)
        tags.append("+".join(tag) or "plain")
    return {k: np.stack(v) for k, v in out.items()}, tags, (n, meq, mi)


def run(first, count, emu=False, verbose=True):
    """-> (mismatching problems, problems, dict tag -> (count, status histogram))"""
    if emu:
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        import pyemu
        solve = pyemu.qp_dense
    else:
        from copra_amd import qp_solve_dense_batch as solve
    bad = tot = 0
    seen = {}
    for seed in range(first, first + count):
        P, tags, (n, meq, mi) = make(seed, emu)
        x, fail, it = solve(P["Q"], P["c"], P["Aeq"] if meq else None, P["beq"] if meq else None, P["Aineq"] if mi else None, P["bineq"] if mi else None,
                            P["XL"], P["XU"])
        for k, tag in enumerate(tags):
            xo, fo, ito = oracle.quadprog_dense(P["Q"][k], P["c"][k], P["Aeq"][k] if meq else None, P["beq"][k] if meq else None,
                                                P["Aineq"][k] if mi else None, P["bineq"][k] if mi else None, P["XL"][k], P["XU"][k])
            tot += 1
            h = seen.setdefault(tag, [0, {}])
            h[0] += 1
            h[1][int(fo)] = h[1].get(int(fo), 0) + 1
            dev = float(np.max(np.abs(x[k] - xo) / np.maximum(np.abs(xo), 1e-3))) if (fo == 0 and fail[k] == 0) else 0.0
            itd = fo == 0 and fail[k] == 0 and tuple(int(v) for v in it[k]) != tuple(int(v) for v in ito)
            # decided by rounding in qpgen2's own arithmetic: a pinned variable's twin bound and a duplicated equality are
            # linearly dependent on an active row -- the slack of the twin is noise, "no solution" or not with its sign; identical rows tie
            degenerate = any(t in tag for t in ("pinned", "dependent"))
            if degenerate and (fail[k] != fo or itd) and dev <= 1e-6:
                h[1]["rounding-decided"] = h[1].get("rounding-decided", 0) + 1
                continue
            if fail[k] != fo or dev > 1e-6 or itd:
                bad += 1
                if verbose:
                    print(seed, k, (n, meq, mi), tag, "status device %d oracle %d" % (fail[k], fo), "iter device %s oracle %s" % (tuple(int(v) for v in it[k]), tuple(ito)),
                          "rel x %.1e" % dev, "  <<<<<<" if (fail[k] != fo or dev > 1e-4) else "", flush=True)
    return bad, tot, seen


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    bad, tot, seen = run(first, count, emu=len(sys.argv) > 3 and sys.argv[3] == "emu")
    for tag, (cnt, hist) in sorted(seen.items()):
        print("  %-16s %6d problems, oracle statuses %s" % (tag, cnt, dict(sorted(hist.items(), key=str))))
    print("seeds %d..%d: %d mismatching problems of %d" % (first, first + count - 1, bad, tot))
