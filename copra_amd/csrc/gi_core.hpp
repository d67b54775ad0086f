// gi_core.hpp -- Goldfarb-Idnani dual active-set QP solver, ONE problem per 64-lane wavefront, n <= 64.
//
// Replaces the eigen-quadprog call of the reference (src/QuadProgSolver.cpp:71 -> Eigen::QuadProgDense::solve ->
// qpgen2).  Same algorithm and the same decisions as qpgen2 (see oracle/copra_oracle.c:gi_qpgen2 for the scalar
// restatement): Cholesky Q = R'R, J = R^-1, unconstrained minimiser, then repeatedly pick the most violated
// constraint normalised by its row norm (lowest index wins ties), compute d = J'n, z = J2 d2, r = R^-1 d1, take
// the min of the dual (t1) and primal (t2) step, add the constraint (Givens on J) or drop the blocking one.
//
// Wave mapping (lane l <-> index l):
//   * J lives in LDS with an odd leading dimension, so "lane = column" reads (d = J'n) and "lane = row" reads
//     (z = J2 d2, Givens sweeps) are both bank-conflict free;
//   * the (n - nact) Givens rotations of a constraint addition are NOT computed one after the other as in qpgen2:
//     all rotation coefficients follow from a suffix scan of d^2 (|h_q| = sqrt(sum_{k>=q} d_k^2)), so lanes compute
//     them in parallel and only the O(n) column sweep stays sequential;
//   * constraints are never materialised: a Rows policy evaluates slacks / normals / norms on the fly.
//
// Rows policy interface (all members are wave-collective unless noted):
//   void   begin_scan(const double* xs)            refresh whatever slack() needs (e.g. the trajectory); syncs
//   double slack(int i, const double* xs)          per-lane: qpgen2's  a_i'x - b_i  in the ORIGINAL orientation
//   double norm(int i)                             per-lane: ||a_i||
//   void   load_normal(int p, double sgn, double* ap)   lane j writes ap[j] = sgn-oriented normal of row p; no sync
#pragma once

#include "plan.hpp"
#include "wave_prims.hpp"

namespace copra_hip {

struct SolverLds {
    double* J;
    int ldj;
    double* R; // packed upper triangular: R(i,c) at R[c(c+1)/2 + i]
    double *xs, *dv, *zv, *uv, *ap, *coef, *cvec, *eqsgn, *scal;
    int *act, *iact;
};

COPRA_DEV SolverLds carve_solver(double* lds, const LdsLayout& L)
{
    SolverLds S;
    S.J = lds + L.J;
    S.ldj = L.ldj;
    S.R = lds + L.R;
    S.xs = lds + L.xs;
    S.dv = lds + L.dv;
    S.zv = lds + L.zv;
    S.uv = lds + L.uv;
    S.ap = lds + L.ap;
    S.coef = lds + L.coef;
    S.cvec = lds + L.cvec;
    S.eqsgn = lds + L.eqsgn;
    S.scal = lds + L.scal;
    S.act = reinterpret_cast<int*>(lds + L.act);
    S.iact = reinterpret_cast<int*>(lds + L.iact);
    return S;
}

// ------------------------------------------------------------------------------------------------
// Factorisation: S.J holds the Hessian (upper triangle), S.cvec the linear term c.
// On exit S.J = J = R^-1 (upper triangular, strict lower part zero), S.xs = -Q^-1 c.  Returns 0 or 2.
// ------------------------------------------------------------------------------------------------
COPRA_DEV int gi_factorize(const SolverLds& S, int n)
{
    const int lane = lane_id();
    double* J = S.J;
    const int ld = S.ldj;

    // right-looking Cholesky, lane = column (qpgen2: dpofa)
    for (int k = 0; k < n; ++k) {
        wave_sync();
        const double piv = J[k * ld + k];
        if (!(piv > 0.0)) return 2; // "Problems with the decomposition of Q" (QuadProgSolver.h:25)
        const double rkk = sqrt(piv);
        double rkj = 0.0;
        if (lane > k && lane < n) rkj = J[k * ld + lane] / rkk;
        wave_sync();
        if (lane > k && lane < n) J[k * ld + lane] = rkj;
        if (lane == k) J[k * ld + k] = rkk;
        wave_sync();
        for (int i = k + 1; i < n; ++i) {
            if (lane >= i && lane < n) J[i * ld + lane] -= J[k * ld + i] * rkj;
        }
    }
    wave_sync();
    // in-place inverse of the upper-triangular factor, lane = column, rows from the bottom up (qpgen2: dpori)
    for (int i = n - 1; i >= 0; --i) {
        const double rii = J[i * ld + i];
        double v = 0.0;
        if (lane >= i && lane < n) {
            double acc = (lane == i) ? 1.0 : 0.0;
            for (int k = i + 1; k <= lane; ++k) acc -= J[i * ld + k] * J[k * ld + lane];
            v = acc / rii;
        }
        wave_sync(); // every lane has read row i
        if (lane >= i && lane < n) J[i * ld + lane] = v;
        wave_sync();
    }
    // zero the strict lower triangle (qpgen2 does the same before the first rotation)
    if (lane < n)
        for (int i = lane + 1; i < n; ++i) J[i * ld + lane] = 0.0;
    wave_sync();
    // unconstrained minimiser x = -J J' c   (qpgen2: dposl on dvec = -c)
    double t = 0.0;
    if (lane < n)
        for (int i = 0; i <= lane; ++i) t += J[i * ld + lane] * S.cvec[i];
    if (lane < n) S.dv[lane] = t;
    wave_sync();
    double x = 0.0;
    if (lane < n)
        for (int j = lane; j < n; ++j) x += J[lane * ld + j] * S.dv[j];
    if (lane < n) S.xs[lane] = -x;
    wave_sync();
    return 0;
}

COPRA_DEV int rcol(int c) { return c * (c + 1) / 2; }

// ------------------------------------------------------------------------------------------------
// Active-set iterations.  Returns qpgen2's ierr (0 ok, 1 infeasible) or 3 (iteration cap).
// ------------------------------------------------------------------------------------------------
template <class Rows>
COPRA_DEV int gi_active_set(const SolverLds& S, int n, int meq, int mtotal, Rows& rows, double vsmall, int max_iter,
    int& iter_main, int& iter_drop)
{
    const int lane = lane_id();
    double* J = S.J;
    const int ld = S.ldj;
    int nact = 0;
    iter_main = 0;
    iter_drop = 0;
    for (int i = lane; i < mtotal; i += kWave) S.act[i] = 0;
    for (int i = lane; i < meq; i += kWave) S.eqsgn[i] = 1.0;
    for (int i = lane; i <= n + 1; i += kWave) S.uv[i] = 0.0;
    wave_sync();

    for (;;) {
        if (iter_main >= max_iter) return 3;
        iter_main += 1;
        // ---------------- step 1: most violated constraint ----------------
        rows.begin_scan(S.xs);
        double best = 0.0, best_s = 0.0;
        int best_i = -1;
        for (int base = 0; base < mtotal; base += kWave) {
            const int i = base + lane;
            if (i < mtotal) {
                double s = rows.slack(i, S.xs);
                if (i < meq) {
                    const double sg = S.eqsgn[i];
                    s = sg * s;
                    if (fabs(s) < vsmall) s = 0.0;
                    if (s > 0.0) S.eqsgn[i] = -sg; // qpgen2 flips the sign of the equality row in place
                    s = -fabs(s);
                } else {
                    if (fabs(s) < vsmall) s = 0.0;
                }
                if (S.act[i]) s = 0.0;
                const double ratio = s / rows.norm(i); // 0/0 = NaN never compares "<"
                if (ratio < best) {
                    best = ratio;
                    best_i = i;
                    best_s = s;
                }
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const double ob = shfl_xor_f64(best, m);
            const double os = shfl_xor_f64(best_s, m);
            const int oi = shfl_xor_i32(best_i, m);
            const bool take = (oi >= 0) && (best_i < 0 || ob < best || (ob == best && oi < best_i));
            if (take) {
                best = ob;
                best_s = os;
                best_i = oi;
            }
        }
        const int nvl = best_i;
        if (nvl < 0) return 0; // optimal
        double sv_nvl = best_s;
        wave_sync(); // eqsgn updates visible

        // ---------------- step 2 ----------------
        for (;;) {
            const double sgn = (nvl < meq) ? S.eqsgn[nvl] : 1.0;
            rows.load_normal(nvl, sgn, S.ap);
            wave_sync();
            // d = J' n+   (lane = column)
            double dj = 0.0;
            if (lane < n) {
                for (int i = 0; i < n; ++i) dj += J[i * ld + lane] * S.ap[i];
                S.dv[lane] = dj;
            }
            wave_sync();
            // z = J2 d2   (lane = row)
            double zi = 0.0;
            if (lane < n)
                for (int j = nact; j < n; ++j) zi += J[lane * ld + j] * S.dv[j];
            // r = R^-1 d1 : column-oriented back substitution, r_c broadcast from lane c
            double acc = (lane < nact) ? dj : 0.0;
            double ri = 0.0;
            for (int c = nact - 1; c >= 0; --c) {
                double rc = 0.0;
                if (lane == c) rc = acc / S.R[rcol(c) + c];
                rc = shfl_f64(rc, c);
                if (lane == c) ri = rc;
                if (lane < c) acc -= S.R[rcol(c) + lane] * rc;
            }
            // t1 = min u_i / r_i over active inequalities with r_i > 0 (lowest position wins ties)
            double t1 = 0.0;
            int it1 = -1;
            if (lane < nact && S.iact[lane] >= meq && ri > 0.0) {
                t1 = S.uv[lane] / ri;
                it1 = lane;
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const double ot = shfl_xor_f64(t1, m);
                const int oi = shfl_xor_i32(it1, m);
                const bool take = (oi >= 0) && (it1 < 0 || ot < t1 || (ot == t1 && oi < it1));
                if (take) {
                    t1 = ot;
                    it1 = oi;
                }
            }
            const bool t1inf = (it1 < 0);
            const double zz = wave_sum(zi * zi);
            bool drop = false;
            if (fabs(zz) <= vsmall) {
                // no step in primal space
                if (t1inf) return 1; // infeasible
                if (lane < nact) S.uv[lane] -= t1 * ri;
                if (lane == 0) S.uv[nact] += t1;
                drop = true;
            } else {
                const double zn = wave_sum((lane < n) ? zi * S.ap[lane] : 0.0);
                double tt = -sv_nvl / zn;
                bool t2min = true;
                if (!t1inf && t1 < tt) {
                    tt = t1;
                    t2min = false;
                }
                if (lane < n) S.xs[lane] += tt * zi;
                if (lane < nact) S.uv[lane] -= tt * ri;
                if (lane == 0) S.uv[nact] += tt;
                if (t2min) {
                    // ---- full step: constraint nvl becomes active; update R and J ----
                    if (lane < nact) S.R[rcol(nact) + lane] = dj;
                    const bool in_tail = (lane >= nact && lane < n);
                    // |h_q| = sqrt(sum_{k>=q} d_k^2) by a suffix scan, scaled by max|d| against under/overflow
                    const double dmax = wave_max(in_tail ? fabs(dj) : 0.0);
                    const double e = (in_tail && dmax > 0.0) ? dj / dmax : 0.0;
                    double suf = e * e;
#pragma unroll
                    for (int off = 1; off < kWave; off <<= 1) suf += shfl_down0_f64(suf, off);
                    double h = (lane == n - 1) ? dj : copysign(dmax * sqrt(suf), dj);
                    if (!in_tail) h = 0.0;
                    const double h_prev = shfl_up0_f64(h, 1); // h_{q-1}
                    const double d_prev = shfl_up0_f64(dj, 1); // d_{q-1}
                    // rotation q acts on columns (q-1, q), q = nact+1 .. n-1
                    if (lane > nact && lane < n) {
                        double gc = 1.0, gs = 0.0, nu_ = 0.0, skip = 1.0;
                        if (h != 0.0) {
                            gc = d_prev / h_prev;
                            gs = h / h_prev;
                            if (gc != 1.0) {
                                nu_ = gs / (1.0 + gc);
                                skip = 0.0;
                            }
                        }
                        S.coef[4 * lane + 0] = gc;
                        S.coef[4 * lane + 1] = gs;
                        S.coef[4 * lane + 2] = nu_;
                        S.coef[4 * lane + 3] = skip;
                    }
                    if (lane == nact) {
                        S.R[rcol(nact) + nact] = h; // new diagonal element of R
                        S.iact[nact] = nvl;
                        S.act[nvl] = 1;
                    }
                    wave_sync();
                    if (lane < n && nact + 1 < n) {
                        double carry = J[lane * ld + (n - 1)];
                        for (int q = n - 1; q > nact; --q) {
                            const double a = J[lane * ld + q - 1];
                            if (S.coef[4 * q + 3] != 0.0) {
                                J[lane * ld + q] = carry;
                                carry = a;
                            } else {
                                const double gc = S.coef[4 * q + 0], gs = S.coef[4 * q + 1], nu_ = S.coef[4 * q + 2];
                                const double t = gc * a + gs * carry;
                                J[lane * ld + q] = nu_ * (a + t) - carry;
                                carry = t;
                            }
                        }
                        J[lane * ld + nact] = carry;
                    }
                    nact += 1;
                    wave_sync();
                    break; // back to step 1
                } else {
                    // ---- partial step: recompute the slack of nvl, then drop the blocking constraint ----
                    wave_sync();
                    rows.begin_scan(S.xs);
                    double s = rows.slack(nvl, S.xs); // wave-uniform argument: every lane computes the same value
                    if (nvl < meq) {
                        const double sg = S.eqsgn[nvl];
                        s = sg * s;
                        wave_sync();
                        if (s > 0.0 && lane == 0) S.eqsgn[nvl] = -sg;
                        s = -fabs(s);
                    }
                    sv_nvl = s;
                    drop = true;
                }
            }
            if (drop) {
                wave_sync();
                // ---- drop the it1-th active constraint ----
                if (lane == 0) S.act[S.iact[it1]] = 0;
                for (int q = it1; q < nact - 1; ++q) {
                    wave_sync();
                    const double a = S.R[rcol(q + 1) + q]; // R(q, q+1)
                    const double b = S.R[rcol(q + 1) + q + 1]; // R(q+1, q+1)
                    bool rot = false;
                    double gc = 1.0, gs = 0.0, nu_ = 0.0;
                    if (b != 0.0) {
                        const double big = fmax(fabs(a), fabs(b)), small = fmin(fabs(a), fabs(b));
                        const double tg = copysign(big * sqrt(1.0 + (small / big) * (small / big)), a);
                        gc = a / tg;
                        gs = b / tg;
                        if (gc != 1.0) {
                            rot = true;
                            nu_ = gs / (1.0 + gc);
                        }
                    }
                    wave_sync();
                    if (rot) {
                        // rows q, q+1 of R for columns q+1 .. nact-1 (lane = column)
                        const int c = q + 1 + lane;
                        if (c < nact) {
                            const double x = S.R[rcol(c) + q], y = S.R[rcol(c) + q + 1];
                            const double t = gc * x + gs * y;
                            S.R[rcol(c) + q + 1] = nu_ * (x + t) - y;
                            S.R[rcol(c) + q] = t;
                        }
                        // columns q, q+1 of J (lane = row)
                        if (lane < n) {
                            const double x = J[lane * ld + q], y = J[lane * ld + q + 1];
                            const double t = gc * x + gs * y;
                            J[lane * ld + q + 1] = nu_ * (x + t) - y;
                            J[lane * ld + q] = t;
                        }
                    }
                    wave_sync();
                    // shift column q+1 (rows 0..q) into column q
                    if (lane <= q) S.R[rcol(q) + lane] = S.R[rcol(q + 1) + lane];
                    if (lane == 0) {
                        S.uv[q] = S.uv[q + 1];
                        S.iact[q] = S.iact[q + 1];
                    }
                }
                wave_sync();
                if (lane == 0) {
                    S.uv[nact - 1] = S.uv[nact];
                    S.uv[nact] = 0.0;
                    S.iact[nact - 1] = 0;
                }
                nact -= 1;
                iter_drop += 1;
                wave_sync();
                if (iter_drop > max_iter) return 3;
            }
        }
    }
}

} // namespace copra_hip
