"""AutoSpan -- host-side input normalisation, mirror of the reference's include/AutoSpan.h / src/AutoSpan.cpp and of
the autoSpan() members of the cost / constraint classes (src/costFunctions.cpp:36-42,114-120,164-171;
src/constraints.cpp:38-43,99-104,161-169,232-261,326-331).  Works on the plain-dict problem descriptions."""
import numpy as np

from ._capi import CopraDomainError


def span_matrix(mat, new_dim, add_cols=0):
    """AutoSpan::spanMatrix (src/AutoSpan.cpp:10-28): block-diagonal repetition until the matrix has new_dim rows;
    add_cols extra zero block-columns (Mixed*: X has one more step than U)."""
    mat = np.atleast_2d(np.asarray(mat, dtype=np.float64))
    rows, cols = mat.shape
    if new_dim == rows:
        return mat
    steps = new_dim // rows
    if steps * rows != new_dim:
        raise CopraDomainError("spanMatrix: new dimension %d is not a multiple of %d rows" % (new_dim, rows))
    out = np.zeros((new_dim, cols * (steps + add_cols)))
    for i in range(steps):
        out[i * rows:(i + 1) * rows, i * cols:(i + 1) * cols] = mat
    return out


def span_vector(vec, new_dim):
    """AutoSpan::spanVector (src/AutoSpan.cpp:30-47): tile the vector until its size is new_dim."""
    vec = np.atleast_1d(np.asarray(vec, dtype=np.float64))
    rows = vec.shape[0]
    if new_dim == rows:
        return vec
    steps = new_dim // rows
    if steps * rows != new_dim:
        raise CopraDomainError("spanVector: new dimension %d is not a multiple of %d" % (new_dim, rows))
    return np.tile(vec, steps)


def _rows(a):
    return 0 if a is None else np.atleast_1d(np.asarray(a)).shape[0]


def autospan_cost(c):
    """autoSpan() of TrajectoryCost / ControlCost / MixedCost (TargetCost has none)"""
    c = dict(c)
    kind = c["kind"]
    w = c.get("weights")
    if w is None:
        w = np.ones(_rows(c["p"]))  # default weights (costFunctions.h:117)
    if kind == "trajectory":
        m = max(_rows(np.atleast_2d(c["M"])), _rows(w), _rows(c["p"]))
        c["M"], c["p"], c["weights"] = span_matrix(c["M"], m), span_vector(c["p"], m), span_vector(w, m)
    elif kind == "control":
        m = max(_rows(np.atleast_2d(c["N"])), _rows(w), _rows(c["p"]))
        c["N"], c["p"], c["weights"] = span_matrix(c["N"], m), span_vector(c["p"], m), span_vector(w, m)
    elif kind == "mixed":
        m = max(_rows(np.atleast_2d(c["M"])), _rows(np.atleast_2d(c["N"])), _rows(w), _rows(c["p"]))
        c["M"] = span_matrix(c["M"], m, 1)  # "This is tricky" (costFunctions.cpp:167)
        c["N"], c["p"], c["weights"] = span_matrix(c["N"], m), span_vector(c["p"], m), span_vector(w, m)
    return c


def autospan_cstr(c):
    """autoSpan() of the five constraint classes"""
    c = dict(c)
    kind = c["kind"]
    if kind == "trajectory":
        m = max(_rows(np.atleast_2d(c["E"])), _rows(c["f"]))
        c["E"], c["f"] = span_matrix(c["E"], m), span_vector(c["f"], m)
    elif kind == "control":
        m = max(_rows(np.atleast_2d(c["G"])), _rows(c["f"]))
        c["G"], c["f"] = span_matrix(c["G"], m), span_vector(c["f"], m)
    elif kind == "mixed":
        m = max(_rows(c["f"]), _rows(np.atleast_2d(c["E"])), _rows(np.atleast_2d(c["G"])))
        c["E"] = span_matrix(c["E"], m, 1)  # constraints.cpp:166
        c["G"], c["f"] = span_matrix(c["G"], m), span_vector(c["f"], m)
    else:
        m = max(_rows(c["lower"]), _rows(c["upper"]))
        c["lower"], c["upper"] = span_vector(c["lower"], m), span_vector(c["upper"], m)
    return c
