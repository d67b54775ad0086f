// gi_large.hpp -- Goldfarb-Idnani dual active-set solver for n > 64 decision variables: ONE problem per WORKGROUP.
//
// Same algorithm and the same decisions as gi_core.hpp (the restatement of eigen-quadprog's qpgen2 behind
// QuadProgDenseSolver::SI_solve, reference src/QuadProgSolver.cpp:45-72), but J = L^-T (n x n) and the triangular
// factor R of the active set no longer fit the LDS of a CU, so they live in a per-workgroup workspace in HBM (L2 /
// Infinity-Cache resident while the workgroup works on it) and only the O(n) vectors stay in LDS.
//
// Data layout (column-major, leading dimension ld = n rounded up to 8 doubles = one 64-byte sector):
//   F : Hessian (both triangles) -> its Cholesky factor L (lower triangle) -> after the inversion the space is reused
//       for R (upper triangle, column c = c-th active constraint)
//   J : L^-T (upper triangle at the start, dense once constraints have been added)
// Thread t owns ROW t of J / F (workgroup size >= n), so every pass over J is a sequence of fully coalesced
// 512-byte loads per wave:  z = J2 d2 and the Givens sweeps walk the columns with thread = row;  d = J' n+ gives
// each WAVE a set of columns (lanes stride the rows, DPP reduction per column).
// Per-iteration HBM/L2 traffic: 8 n^2 (d) + 8 n (n - nact) (z) + 16 n (n - nact) (Givens sweep of an add step).
#pragma once

#include "block_prims.hpp"
#include "plan.hpp"

namespace copra_hip {

struct LargeSolver {
    int n, ld;
    double* J; // global, n x ld
    double* F; // global, n x ld
    // LDS
    double *xs, *cv, *np, *dv, *rv, *uv, *hv, *coef, *nb, *eqsgn, *red, *stage, *dblk;
    unsigned char* act; // one flag per constraint row
    int* iact;
    long long* fine; // profiling builds: 16 accumulated shader-clock counters of this instance (or nullptr)
};

// sub-phases of the active-set loop, accumulated over the iterations (profiling builds only)
enum { kLpScan = 0, kLpNormal, kLpD, kLpZ, kLpTri, kLpStep, kLpAdd, kLpDrop, kLpPartial, kLpCount };
#ifdef COPRA_FINE_PROFILE
#define COPRA_LPROF_DECL                                                                                              \
    long long lprof[kLpCount];                                                                                        \
    for (int k_ = 0; k_ < kLpCount; ++k_) lprof[k_] = 0;                                                               \
    long long lprof_last = cycle_counter()
#define COPRA_LPROF(k)                                                                                                \
    do {                                                                                                              \
        const long long now_ = cycle_counter();                                                                       \
        lprof[k] += now_ - lprof_last;                                                                                \
        lprof_last = now_;                                                                                            \
    } while (0)
#define COPRA_LPROF_FLUSH                                                                                             \
    do {                                                                                                              \
        if (S.fine && bt_tid() == 0)                                                                                  \
            for (int k_ = 0; k_ < kLpCount; ++k_) S.fine[k_] = lprof[k_];                                              \
    } while (0)
#else
#define COPRA_LPROF_DECL
#define COPRA_LPROF(k)
#define COPRA_LPROF_FLUSH
#endif

COPRA_DEV LargeSolver carve_large(double* lds, const LargeLds& L, int n, double* J, double* F)
{
    LargeSolver S;
    S.n = n;
    S.ld = (n + 7) & ~7;
    S.J = J;
    S.F = F;
    S.xs = lds + L.xs;
    S.cv = lds + L.cv;
    S.np = lds + L.np;
    S.dv = lds + L.dv;
    S.rv = lds + L.rv;
    S.uv = lds + L.uv;
    S.hv = lds + L.hv;
    S.coef = lds + L.coef;
    S.nb = lds + L.nb;
    S.eqsgn = lds + L.eqsgn;
    S.red = lds + L.red;
    S.stage = lds + L.stage;
    S.dblk = lds + L.dblk;
    S.act = (unsigned char*)(lds + L.act);
    S.iact = (int*)(lds + L.iact);
    S.fine = nullptr;
    return S;
}

// ---- workgroup reductions (every thread of the workgroup calls them; the result is uniform) ----------------------
COPRA_DEV double block_sum(double v, double* red)
{
    v = wave_sum(v);
    if (bt_lane() == 0) red[bt_wave()] = v;
    bt_sync();
    double s = 0.0;
    const int nw = bt_nwaves();
    for (int w = 0; w < nw; ++w) s += red[w];
    bt_sync();
    return s;
}
COPRA_DEV double block_max(double v, double* red)
{
    v = wave_max(v);
    if (bt_lane() == 0) red[bt_wave()] = v;
    bt_sync();
    double s = red[0];
    const int nw = bt_nwaves();
    for (int w = 1; w < nw; ++w) s = fmax(s, red[w]);
    bt_sync();
    return s;
}
// smallest key, ties -> smallest idx; threads without a candidate pass idx < 0 (same rule as wave_argmin)
COPRA_DEV void block_argmin(double& key, int& idx, double& payload, double* red)
{
    wave_argmin(key, idx, payload);
    const int nw = bt_nwaves();
    if (bt_lane() == 0) {
        const int w = bt_wave();
        red[w] = key;
        red[kLargeMaxWaves + w] = payload;
        red[2 * kLargeMaxWaves + w] = (double)idx;
    }
    bt_sync();
    double bk = red[0], bp = red[kLargeMaxWaves];
    int bi = (int)red[2 * kLargeMaxWaves];
    for (int w = 1; w < nw; ++w) {
        const double ok = red[w], op = red[kLargeMaxWaves + w];
        const int oi = (int)red[2 * kLargeMaxWaves + w];
        const bool take = (oi >= 0) && (bi < 0 || ok < bk || (ok == bk && oi < bi));
        bk = take ? ok : bk;
        bp = take ? op : bp;
        bi = take ? oi : bi;
    }
    bt_sync();
    key = bk;
    payload = bp;
    idx = bi;
}
// inclusive suffix sum over the thread index: result(t) = sum_{s >= t} v(s)
COPRA_DEV double block_suffix_sum(double v, double* red)
{
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) v += shfl_down0_f64(v, off);
    if (bt_lane() == 0) red[bt_wave()] = v;
    bt_sync();
    double add = 0.0;
    const int nw = bt_nwaves();
    for (int w = bt_wave() + 1; w < nw; ++w) add += red[w];
    bt_sync();
    return v + add;
}

// out[c] = sum_{r < rlim} M[r, c] v[r] for c in [0, n): one wave per pair of columns, lanes stride the rows.
// All loads of a group (2 columns x up to 8 row chunks) are issued before the first FMA: the passes over J are
// latency-bound with the few waves a workgroup has unless many loads are in flight per lane.
COPRA_DEV void gl_matvec_t(const LargeSolver& S, const double* M, const double* v, double* out, int rlim)
{
    const int n = S.n, ld = S.ld, lane = bt_lane();
    const int nw = bt_nwaves();
    const int nch = (rlim + kWave - 1) / kWave; // <= 8 (n <= 512)
    double vr[8];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        const int r = lane + kWave * ch;
        vr[ch] = (ch < nch && r < rlim) ? v[r] : 0.0;
    }
    for (int c0 = 2 * bt_wave(); c0 < n; c0 += 2 * nw) {
        const bool two = (c0 + 1 < n);
        double a0[8], a1[8];
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            const int r = lane + kWave * ch;
            const bool live = (ch < nch) && (r < rlim);
            a0[ch] = live ? M[(size_t)c0 * ld + r] : 0.0;
            a1[ch] = (live && two) ? M[(size_t)(c0 + 1) * ld + r] : 0.0;
        }
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            p0 += a0[ch] * vr[ch];
            p1 += a1[ch] * vr[ch];
        }
        p0 = wave_sum(p0);
        p1 = wave_sum(p1);
        if (lane == 0) {
            out[c0] = p0;
            if (two) out[c0 + 1] = p1;
        }
    }
}
// sum_{c in [c0, n)} M[row, c] v[c] for the calling thread's row (0 for threads beyond n); 16 loads in flight
COPRA_DEV double gl_matvec_n(const LargeSolver& S, const double* M, const double* v, int c0)
{
    const int n = S.n, ld = S.ld, row = bt_tid();
    if (row >= n) return 0.0;
    double z0 = 0.0, z1 = 0.0, z2 = 0.0, z3 = 0.0;
    int c = c0;
    for (; c + 15 < n; c += 16) {
        double a[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) a[u] = M[(size_t)(c + u) * ld + row];
#pragma unroll
        for (int u = 0; u < 16; u += 4) {
            z0 += a[u] * v[c + u];
            z1 += a[u + 1] * v[c + u + 1];
            z2 += a[u + 2] * v[c + u + 2];
            z3 += a[u + 3] * v[c + u + 3];
        }
    }
    for (; c + 3 < n; c += 4) {
        const double a0 = M[(size_t)c * ld + row], a1 = M[(size_t)(c + 1) * ld + row];
        const double a2 = M[(size_t)(c + 2) * ld + row], a3 = M[(size_t)(c + 3) * ld + row];
        z0 += a0 * v[c];
        z1 += a1 * v[c + 1];
        z2 += a2 * v[c + 2];
        z3 += a3 * v[c + 3];
    }
    for (; c < n; ++c) z0 += M[(size_t)c * ld + row] * v[c];
    return (z0 + z1) + (z2 + z3);
}

// ---- Cholesky factorisation F = L L' in place (lower triangle), blocked left-looking, panel width kNB -------------
// dpofa semantics (eigen-quadprog): returns 2 if a pivot is not positive.
COPRA_DEV int gl_factorize(const LargeSolver& S)
{
    const int n = S.n, ld = S.ld, tid = bt_tid(), T = bt_size();
    double* F = S.F;
    for (int jb = 0; jb < n; jb += kNB) {
        const int w = (n - jb < kNB) ? n - jb : kNB;
        // rows jb .. jb+w-1 of the finished columns 0 .. jb-1 -> LDS (broadcast operands of the panel update)
        for (int e = tid; e < jb * kNB; e += T) {
            const int k = e / kNB, c = e % kNB;
            S.stage[e] = (c < w) ? F[(size_t)k * ld + jb + c] : 0.0;
        }
        bt_sync();
        const int r = tid;
        const bool mine = (r >= jb && r < n);
        double acc[kNB];
#pragma unroll
        for (int c = 0; c < kNB; ++c) acc[c] = (mine && c < w) ? F[(size_t)(jb + c) * ld + r] : 0.0;
        if (mine) {
            for (int k = 0; k < jb; ++k) {
                const double lr = F[(size_t)k * ld + r];
#pragma unroll
                for (int c = 0; c < kNB; ++c) acc[c] -= lr * S.stage[k * kNB + c];
            }
            if (r < jb + kNB) {
#pragma unroll
                for (int c = 0; c < kNB; ++c) S.dblk[(r - jb) * kNB + c] = acc[c];
            }
        }
        bt_sync();
        // the kNB x kNB diagonal block is factorised in place in LDS: thread i < kNB owns row i; column c is final
        // after step c (one barrier per column).  dblk[kNB * kNB] flags a non-positive pivot.
        if (tid == 0) S.dblk[kNB * kNB] = 0.0;
        for (int c = 0; c < kNB; ++c) {
            if (tid >= c && tid < kNB && c < w && tid < w) {
                double v = S.dblk[tid * kNB + c];
                for (int t = 0; t < c; ++t) v -= S.dblk[tid * kNB + t] * S.dblk[c * kNB + t];
                if (tid == c) {
                    if (!(v > 0.0)) S.dblk[kNB * kNB] = 1.0;
                    S.dblk[c * kNB + c] = sqrt(v);
                } else {
                    S.dblk[tid * kNB + c] = v; // divided by the pivot after the barrier
                }
            }
            bt_sync();
            if (tid > c && tid < kNB && c < w && tid < w) S.dblk[tid * kNB + c] /= S.dblk[c * kNB + c];
            bt_sync();
        }
        if (S.dblk[kNB * kNB] != 0.0) return 2;
        if (mine) {
            double out[kNB];
#pragma unroll
            for (int c = 0; c < kNB; ++c) {
                double v = acc[c];
#pragma unroll
                for (int t = 0; t < c; ++t) v -= out[t] * S.dblk[c * kNB + t];
                out[c] = (c < w) ? v / S.dblk[c * kNB + c] : 0.0;
            }
#pragma unroll
            for (int c = 0; c < kNB; ++c)
                if (c < w && r >= jb + c) F[(size_t)(jb + c) * ld + r] = out[c];
        }
        bt_sync();
    }
    return 0;
}

// ---- J = L^-T (upper triangular, every entry of J written): rows of X = L^-1 in blocks of kNB -----------------------
// Column i of J is row i of X; thread c computes X[ib .. ib+w-1][c] from the finished rows k < ib (J[c, k]).
COPRA_DEV void gl_invert(const LargeSolver& S)
{
    const int n = S.n, ld = S.ld, tid = bt_tid(), T = bt_size();
    const double* F = S.F;
    double* J = S.J;
    for (int ib = 0; ib < n; ib += kNB) {
        const int w = (n - ib < kNB) ? n - ib : kNB;
        const int kk = ib + w;
        for (int e = tid; e < kk * kNB; e += T) { // stage[k][j] = L[ib + j][k]
            const int k = e / kNB, j = e % kNB;
            S.stage[e] = (j < w && k <= ib + j) ? F[(size_t)k * ld + ib + j] : 0.0;
        }
        bt_sync();
        const int c = tid;
        if (c < n) {
            double acc[kNB];
#pragma unroll
            for (int j = 0; j < kNB; ++j) acc[j] = 0.0;
            if (c < ib) {
                for (int k = (c & ~(kWave - 1)); k < ib; ++k) { // X[k][c] = 0 for k < c: start at the wave's first row
                    const double xk = J[(size_t)k * ld + c];
#pragma unroll
                    for (int j = 0; j < kNB; ++j) acc[j] -= S.stage[k * kNB + j] * xk;
                }
            }
            double out[kNB];
#pragma unroll
            for (int j = 0; j < kNB; ++j) {
                double v = acc[j] + ((ib + j == c) ? 1.0 : 0.0);
#pragma unroll
                for (int t = 0; t < j; ++t) v -= S.stage[(ib + t) * kNB + j] * out[t];
                out[j] = (j < w) ? v / S.stage[(ib + j) * kNB + j] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < kNB; ++j)
                if (j < w) J[(size_t)(ib + j) * ld + c] = (c <= ib + j) ? out[j] : 0.0;
        }
        bt_sync();
    }
}

// Unconstrained minimiser xs = -Q^-1 c = -J (J' c); c in S.cv.  Leaves the workgroup synchronised.
COPRA_DEV void gl_unconstrained(const LargeSolver& S)
{
    gl_matvec_t(S, S.J, S.cv, S.dv, S.n);
    bt_sync();
    const double x = gl_matvec_n(S, S.J, S.dv, 0);
    if (bt_tid() < S.n) S.xs[bt_tid()] = -x;
    bt_sync();
}

// ---- the active-set iteration.  Rows policy (workgroup-level):
//   void   begin_scan(const double* xs)                 cooperative; may synchronise
//   double slack_at(int pass, int i, const double* xs)  thread-level: row i = tid + pass * T (stacking order, equalities
//                                                       first); `pass` lets the policy keep its first rows in registers
//   double slack_uniform(int i, const double* xs)       cooperative, uniform result
//   double norm(int i)
//   void   load_normal(int i, double sgn, double* np)   cooperative: np[0..n) = sgn * row (eq) / -row (ineq); caller syncs
//   int    normal_extent(int i)                         uniform: entries of that normal at or past it are zero
//   double ub(int j), lb(int j)
// Returns 0 optimal, 1 infeasible, 3 iteration limit.
template <class Rows>
COPRA_DEV int gl_active_set(const LargeSolver& S, int meq, int mgen, Rows& rows, double vsmall, int max_iter,
    int& iter_main, int& iter_drop)
{
    const int n = S.n, ld = S.ld, tid = bt_tid(), T = bt_size(), wave = bt_wave();
    const int mtotal = mgen + 2 * n; // QuadProgSolver.cpp:51: the bounds are 2n more inequality rows
    double* J = S.J;
    double* R = S.F; // the factor's space is free once J exists
    const bool own = tid < n;
    int nact = 0;
    iter_main = 0;
    iter_drop = 0;
    for (int i = tid; i < mtotal; i += T) S.act[i] = 0;
    for (int i = tid; i < meq; i += T) S.eqsgn[i] = 1.0;
    for (int i = tid; i <= n + 1; i += T) S.uv[i] = 0.0;
    const double ubj = own ? rows.ub(tid) : 0.0, lbj = own ? rows.lb(tid) : 0.0;
    // XL_j == XU_j (the default x0 bounds of InitialStateLMPC, InitialStateLMPC.cpp:20-28): once one row of the pair is
    // active the other one is its negative -- linearly dependent, its slack is rounding noise and never a violation
    // (qpgen2 would otherwise try to add it, find no step and report "no solution" whenever the noise is negative)
    const bool pinned = (ubj - lbj) <= 1e-12 * fmax(1.0, fabs(ubj)); // (an interval narrower than the noise counts as one)
    bt_sync();
    COPRA_LPROF_DECL;

    for (;;) {
        if (iter_main >= max_iter) {
            COPRA_LPROF_FLUSH;
            return 3;
        }
        iter_main += 1;
        // ---------------- step 1: most violated constraint ----------------
        rows.begin_scan(S.xs);
        double best = 0.0, best_s = 0.0;
        int best_i = -1;
        for (int i = tid, pass = 0; i < mgen; i += T, ++pass) { // general rows (equalities first)
            double s = rows.slack_at(pass, i, S.xs);
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
        if (own) { // bound rows: unit norm; an infinite / DBL_MAX bound gives a slack that is never negative
            const double xj = S.xs[tid];
            double s = ubj - xj; // row mgen + j of [I]
            if (fabs(s) < vsmall) s = 0.0;
            if (S.act[mgen + tid] || (pinned && S.act[mgen + n + tid])) s = 0.0;
            if (s < best) {
                best = s;
                best_i = mgen + tid;
                best_s = s;
            }
            s = xj - lbj; // row mgen + n + j of [-I]
            if (fabs(s) < vsmall) s = 0.0;
            if (S.act[mgen + n + tid] || (pinned && S.act[mgen + tid])) s = 0.0;
            if (s < best) {
                best = s;
                best_i = mgen + n + tid;
                best_s = s;
            }
        }
        block_argmin(best, best_i, best_s, S.red);
        COPRA_LPROF(kLpScan);
        const int nvl = best_i;
        if (nvl < 0) {
            COPRA_LPROF_FLUSH;
            return 0; // optimal
        }
        double sv_nvl = best_s;

        // ---------------- step 2 ----------------
        for (;;) {
            if (nvl < mgen) {
                const double sgn = (nvl < meq) ? S.eqsgn[nvl] : 1.0;
                rows.load_normal(nvl, sgn, S.np);
                bt_sync();
                COPRA_LPROF(kLpNormal);
                gl_matvec_t(S, J, S.np, S.dv, rows.normal_extent(nvl)); // d = J' n+ (rows past the extent are zero)
            } else { // rows of -[I; -I]: -e_j for an upper bound, +e_j for a lower bound -> d = -/+ row j of J
                const int q = nvl - mgen;
                const int jq = (q < n) ? q : q - n;
                const double sg = (q < n) ? -1.0 : 1.0;
                if (own) {
                    S.np[tid] = (tid == jq) ? sg : 0.0;
                    S.dv[tid] = sg * J[(size_t)tid * ld + jq];
                }
                COPRA_LPROF(kLpNormal);
            }
            bt_sync();
            COPRA_LPROF(kLpD);
            const double dj = own ? S.dv[tid] : 0.0;
            const double zi = gl_matvec_n(S, J, S.dv, nact); // z = J2 d2
            COPRA_LPROF(kLpZ);
            // r = R^-1 d1: 64-row diagonal blocks are solved inside the wave that owns them (readlane broadcasts),
            // the off-diagonal part is a coalesced column sweep by the rows above
            double acc = (tid < nact) ? dj : 0.0;
            double ri = 0.0;
            for (int blk = (nact - 1) / kWave; nact > 0 && blk >= 0; --blk) {
                const int lo = blk * kWave;
                const int hi = (nact < lo + kWave) ? nact : lo + kWave;
                if (wave == blk) {
                    for (int c1 = hi; c1 > lo; c1 -= 16) { // columns c1-1 .. c1-16: one HBM round trip per 16 steps
                        double col[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            const int c = c1 - 1 - u;
                            col[u] = (c >= lo && tid <= c) ? R[(size_t)c * ld + tid] : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            const int c = c1 - 1 - u;
                            if (c >= lo) {
                                double rc = 0.0;
                                if (tid == c) rc = acc / col[u];
                                rc = bcast_f64(rc, c - lo);
                                if (tid == c) ri = rc;
                                if (tid < c) acc -= col[u] * rc;
                            }
                        }
                    }
                    if (tid < hi) S.rv[tid] = ri;
                }
                bt_sync();
                if (tid < lo) {
                    int c = lo;
                    for (; c + 15 < hi; c += 16) {
                        double col[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) col[u] = R[(size_t)(c + u) * ld + tid];
#pragma unroll
                        for (int u = 0; u < 16; ++u) acc -= col[u] * S.rv[c + u];
                    }
                    for (; c < hi; ++c) acc -= R[(size_t)c * ld + tid] * S.rv[c];
                }
            }
            COPRA_LPROF(kLpTri);
            // t1 = min u_i / r_i over active inequalities with r_i > 0 (lowest position wins ties)
            double t1 = 0.0;
            int it1 = -1;
            if (tid < nact && S.iact[tid] >= meq && ri > 0.0) {
                t1 = S.uv[tid] / ri;
                it1 = tid;
            }
            if (nact > 0) {
                double dummy = 0.0;
                block_argmin(t1, it1, dummy, S.red);
            }
            const bool t1inf = (it1 < 0);
            const double zz = block_sum(zi * zi, S.red);
            bool drop = false;
            if (fabs(zz) <= vsmall) {
                // no step in primal space
                if (t1inf) {
                    COPRA_LPROF_FLUSH;
                    return 1; // infeasible
                }
                if (tid < nact) S.uv[tid] -= t1 * ri;
                if (tid == 0) S.uv[nact] += t1;
                drop = true;
            } else {
                const double zn = block_sum(own ? zi * S.np[tid] : 0.0, S.red);
                double tt = -sv_nvl / zn;
                bool t2min = true;
                if (!t1inf && t1 < tt) {
                    tt = t1;
                    t2min = false;
                }
                if (own) S.xs[tid] += tt * zi;
                if (tid < nact) S.uv[tid] -= tt * ri;
                if (tid == 0) S.uv[nact] += tt;
                COPRA_LPROF(kLpStep);
                if (t2min) {
                    // ---- full step: constraint nvl becomes active; update R and J ----
                    if (tid < nact) R[(size_t)nact * ld + tid] = dj;
                    const bool in_tail = own && tid >= nact;
                    // |h_q| = sqrt(sum_{k>=q} d_k^2) by a suffix scan, scaled by max|d| against under/overflow
                    const double dmax = block_max(in_tail ? fabs(dj) : 0.0, S.red);
                    const double e = (in_tail && dmax > 0.0) ? dj / dmax : 0.0;
                    const double suf = block_suffix_sum(e * e, S.red);
                    double h = (tid == n - 1) ? dj : copysign(dmax * sqrt(suf), dj);
                    if (!in_tail) h = 0.0;
                    if (own) S.hv[tid] = h;
                    bt_sync();
                    // rotation q acts on columns (q-1, q), q = nact+1 .. n-1, as the reflection
                    //   col_{q-1}' = gc col_{q-1} + gs col_q ,  col_q' = gs col_{q-1} - gc col_q
                    // (qpgen2's "nu" form is the same map); identity where qpgen2 skips (h_q == 0 or gc == 1)
                    if (tid >= 1 && own) {
                        double c0 = 1.0, c1 = 0.0, c2 = 0.0, c3 = 1.0;
                        if (tid > nact && h != 0.0) {
                            const double h_prev = S.hv[tid - 1], d_prev = S.dv[tid - 1];
                            const double gc = d_prev / h_prev;
                            const double gs = h / h_prev;
                            if (gc != 1.0) {
                                c0 = gc;
                                c1 = gs;
                                c2 = gs;
                                c3 = -gc;
                            }
                        }
                        S.coef[4 * tid + 0] = c0;
                        S.coef[4 * tid + 1] = c1;
                        S.coef[4 * tid + 2] = c2;
                        S.coef[4 * tid + 3] = c3;
                    }
                    if (tid == nact) {
                        R[(size_t)nact * ld + nact] = h; // new diagonal element of R
                        S.iact[nact] = nvl;
                        S.act[nvl] = 1;
                    }
                    bt_sync();
                    if (nact + 1 < n && own) {
                        double carry = J[(size_t)(n - 1) * ld + tid];
                        int q = n - 1;
                        // sixteen rotations per round trip: the loads of a group are issued before its first store
                        for (; q - 15 > nact; q -= 16) {
                            double a[16];
#pragma unroll
                            for (int u = 0; u < 16; ++u) a[u] = J[(size_t)(q - 1 - u) * ld + tid];
#pragma unroll
                            for (int u = 0; u < 16; ++u) {
                                const double* cf = S.coef + 4 * (q - u);
                                const double t = cf[0] * a[u] + cf[1] * carry;
                                a[u] = cf[2] * a[u] + cf[3] * carry;
                                carry = t;
                            }
#pragma unroll
                            for (int u = 0; u < 16; ++u) J[(size_t)(q - u) * ld + tid] = a[u];
                        }
                        for (; q > nact; --q) {
                            const double a = J[(size_t)(q - 1) * ld + tid];
                            const double* cf = S.coef + 4 * q;
                            const double t = cf[0] * a + cf[1] * carry;
                            J[(size_t)q * ld + tid] = cf[2] * a + cf[3] * carry;
                            carry = t;
                        }
                        J[(size_t)nact * ld + tid] = carry;
                    }
                    nact += 1;
                    bt_sync();
                    COPRA_LPROF(kLpAdd);
                    break; // back to step 1
                } else {
                    // ---- partial step: recompute the slack of nvl, then drop the blocking constraint ----
                    bt_sync();
                    double s;
                    if (nvl < mgen) {
                        rows.begin_scan(S.xs);
                        s = rows.slack_uniform(nvl, S.xs);
                    } else {
                        const int q = nvl - mgen;
                        s = (q < n) ? rows.ub(q) - S.xs[q] : S.xs[q - n] - rows.lb(q - n);
                    }
                    if (nvl < meq) {
                        const double sg = S.eqsgn[nvl];
                        s = sg * s;
                        bt_sync();
                        if (s > 0.0 && tid == 0) S.eqsgn[nvl] = -sg;
                        s = -fabs(s);
                    }
                    sv_nvl = s;
                    drop = true;
                    COPRA_LPROF(kLpPartial);
                }
            }
            if (drop) {
                bt_sync();
                // ---- drop the it1-th active constraint: columns it1+1 .. last of R move one place to the left after
                // the Givens rotations q = it1 .. last-1 on rows (q, q+1) of R / columns (q, q+1) of J.  Only the
                // rotation COEFFICIENTS are sequential (rotation q is defined by column q+1 after rotation q-1), so:
                //   1. rows < it1 of R just shift (thread = row);
                //   2. thread = column keeps the running row-q element of its column in a register, prefetches the
                //      next 8 rows of its column per HBM round trip, and column q+1 publishes (a, b) through LDS:
                //      one barrier per rotation, no memory round trip; the coefficients are kept in LDS;
                //   3. J is updated in ONE coalesced sweep over columns it1 .. last (thread = row), like an add step.
                const int last = nact - 1;
                if (tid == 0) S.act[S.iact[it1]] = 0;
                if (tid < it1) {
                    int c = it1 + 1;
                    for (; c + 15 <= last; c += 16) {
                        double v[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) v[u] = R[(size_t)(c + u) * ld + tid];
#pragma unroll
                        for (int u = 0; u < 16; ++u) R[(size_t)(c + u - 1) * ld + tid] = v[u];
                    }
                    for (; c <= last; ++c) R[(size_t)(c - 1) * ld + tid] = R[(size_t)c * ld + tid];
                }
                {
                    const int c = tid;
                    const bool colact = (c > it1 && c <= last);
                    double x = colact ? R[(size_t)c * ld + it1] : 0.0;
                    for (int q0 = it1; q0 < last; q0 += 8) {
                        double yb[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int row = q0 + 1 + u;
                            yb[u] = (colact && row <= c) ? R[(size_t)c * ld + row] : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int q = q0 + u;
                            if (q < last) {
                                const double y = yb[u];
                                double* slot = S.red + 2 * (q & 1);
                                if (c == q + 1) {
                                    slot[0] = x; // R(q, q+1) after the previous rotation
                                    slot[1] = y; // R(q+1, q+1)
                                }
                                bt_sync();
                                const double a = slot[0], b = slot[1];
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
                                if (tid == 0) {
                                    S.coef[4 * q + 0] = gc;
                                    S.coef[4 * q + 1] = gs;
                                    S.coef[4 * q + 2] = nu_;
                                    S.coef[4 * q + 3] = rot ? 1.0 : 0.0;
                                }
                                if (colact && c >= q + 1) {
                                    double t = x, yn = y;
                                    if (rot) {
                                        t = gc * x + gs * y;
                                        yn = nu_ * (x + t) - y;
                                    }
                                    R[(size_t)(c - 1) * ld + q] = t;
                                    x = yn;
                                }
                            }
                        }
                    }
                }
                // multipliers and the index list move with their columns
                double uvn = 0.0;
                int ian = 0;
                if (tid >= it1 && tid <= last) {
                    uvn = S.uv[tid + 1];
                    ian = (tid < last) ? S.iact[tid + 1] : 0;
                }
                bt_sync(); // also publishes the rotation coefficients
                if (tid >= it1 && tid <= last) {
                    S.uv[tid] = uvn;
                    S.iact[tid] = ian;
                }
                if (tid == 0) S.uv[nact] = 0.0;
                if (own && it1 < last) {
                    double x = J[(size_t)it1 * ld + tid];
                    int q = it1;
                    for (; q + 15 < last; q += 16) {
                        double yv[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) yv[u] = J[(size_t)(q + 1 + u) * ld + tid];
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            const double* cf = S.coef + 4 * (q + u);
                            double t = x, yn = yv[u];
                            if (cf[3] != 0.0) {
                                t = cf[0] * x + cf[1] * yv[u];
                                yn = cf[2] * (x + t) - yv[u];
                            }
                            yv[u] = t;
                            x = yn;
                        }
#pragma unroll
                        for (int u = 0; u < 16; ++u) J[(size_t)(q + u) * ld + tid] = yv[u];
                    }
                    for (; q < last; ++q) {
                        const double y = J[(size_t)(q + 1) * ld + tid];
                        const double* cf = S.coef + 4 * q;
                        double t = x, yn = y;
                        if (cf[3] != 0.0) {
                            t = cf[0] * x + cf[1] * y;
                            yn = cf[2] * (x + t) - y;
                        }
                        J[(size_t)q * ld + tid] = t;
                        x = yn;
                    }
                    J[(size_t)last * ld + tid] = x;
                }
                nact -= 1;
                iter_drop += 1;
                bt_sync();
                COPRA_LPROF(kLpDrop);
                if (iter_drop > max_iter) {
                    COPRA_LPROF_FLUSH;
                    return 3;
                }
            }
        }
    }
}

} // namespace copra_hip
