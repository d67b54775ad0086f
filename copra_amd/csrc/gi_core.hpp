// gi_core.hpp -- Goldfarb-Idnani dual active-set QP solver, ONE problem per 64-lane wavefront, n <= 64.
//
// Replaces the eigen-quadprog call of the reference (src/QuadProgSolver.cpp:71 -> Eigen::QuadProgDense::solve ->
// qpgen2).  Same algorithm and the same decisions as qpgen2 (the test oracle holds a scalar
// restatement): Cholesky Q = R'R, J = R^-1, unconstrained minimiser, then repeatedly pick the most violated
// constraint normalised by its row norm (lowest index wins ties), compute d = J'n, z = J2 d2, r = R^-1 d1, take
// the min of the dual (t1) and primal (t2) step, add the constraint (Givens on J) or drop the blocking one.
//
// Wave mapping (lane l <-> index l):
//   * J lives in LDS with an odd leading dimension, so "lane = column" reads (d = J'n) and "lane = row" reads
//     (z = J2 d2, Givens sweeps) are both bank-conflict free;
//   * Cholesky and the triangular inverse are LEFT-LOOKING and blocked by 4 rows: one lane-private LDS read feeds
//     four FMAs whose second operands are wave-uniform (broadcast) LDS reads -- 0.75 LDS reads per FMA instead of 2,
//     no read-modify-write traffic, and the pivots use a Newton-refined v_rsq_f64 instead of sqrt + divide;
//   * the (n - nact) Givens rotations of a constraint addition are NOT computed one after the other as in qpgen2:
//     all rotation coefficients follow from a suffix scan of d^2 (|h_q| = sqrt(sum_{k>=q} d_k^2)), so lanes compute
//     them in parallel and only the O(n) column sweep stays sequential (branch-free, reflection form);
//   * constraints are never materialised: a Rows policy evaluates slacks / normals / norms on the fly.
// NV > 0 fixes the number of variables at compile time (loops unroll, LDS reads batch); NV == 0 is the generic path.
//
// Rows policy interface for the mgen GENERAL rows (the 2n bound rows [I; -I] of QuadProgSolver.cpp:59-69 are handled
// here, from per-lane register copies of XU_j / XL_j):
//   void   begin_scan(const double* xs)            refresh whatever slack() needs (e.g. the trajectory); syncs
//   double slack(int i, const double* xs)          per-lane (i = 64c + lane): qpgen2's a_i'x - b_i, ORIGINAL orientation
//   double slack_uniform(int p, const double* xs)  the same for a wave-uniform row index
//   double norm(int i)                             per-lane: ||a_i||
//   void   load_normal(int p, double sgn, double* ap)   lane j writes ap[j] = sgn-oriented normal of row p; no sync
//   double ub(int j), lb(int j)                    per-lane: XU_j, XL_j
#pragma once

#include "plan.hpp"
#include "wave_prims.hpp"
#include "ric_factor.hpp"

// Fine-grained stamps for profiling builds (make libcopra_hip_prof.so); compiled out of the product library.
#ifdef COPRA_FINE_PROFILE
#define COPRA_FINE(tag)                                                                                               \
    do {                                                                                                              \
        if (copra_fine_n < 32) copra_fine[copra_fine_n++] = cycle_counter();                                         \
    } while (0)
#define COPRA_FINE_DECL long long copra_fine[32]; int copra_fine_n = 0
#define COPRA_FINE_ARGS , long long* copra_fine, int& copra_fine_n
#define COPRA_FINE_PASS , copra_fine, copra_fine_n
#else
#define COPRA_FINE(tag) do { } while (0)
#define COPRA_FINE_DECL
#define COPRA_FINE_ARGS
#define COPRA_FINE_PASS
#endif

namespace copra_hip {

struct alignas(16) f64x2 { // one 16-byte LDS access
    double lo, hi;
};

struct SolverLds {
    double* J;
    int ldj;
    // (factor-only layout: coef does not exist, 1 / R(i,i) sits on the diagonal of the packed factor)
    double* Q1; // factor-only layout: column k of the orthonormal basis of R^-T N at Q1[64 k + lane] (else unused)
    double* R; // packed upper triangular: R(i,c) at R[c(c+1)/2 + i]
    int rcap; // columns R has room for (LdsLayout::rcap)
    double *xs, *dv, *zv, *uv, *ap, *coef, *cvec, *eqsgn, *scal;
    unsigned char* act; // one flag per row
    int* iact;
    const double* Jsrc; // shared-model path: J = R^-1 of the whole batch in HBM, copied on first need (else nullptr);
                        // factor-only layout: the packed factor R instead, and ...
    const double* rinv_src; // ... 1 / R(i,i)
    double* ricd; // Riccati-factor tier: a double nobody reads (the lanes with nothing to store write there)
    double* ricxi; // ... and, if set, where z = R^-1 v leaves its closed-loop states (nx (N + 1) doubles; StageRows::xi)
};

COPRA_DEV SolverLds carve_solver(double* lds, const LdsLayout& L)
{
    SolverLds S;
    S.J = lds + L.J;
    S.Jsrc = nullptr;
    S.rinv_src = nullptr;
    S.ldj = L.ldj;
    S.Q1 = lds + L.Q1;
    S.R = lds + L.R;
    S.rcap = L.rcap;
    S.xs = lds + L.xs;
    S.dv = lds + L.dv;
    S.zv = lds + L.zv;
    S.uv = lds + L.uv;
    S.ap = lds + L.ap;
    S.coef = lds + L.coef;
    S.cvec = lds + L.cvec;
    S.eqsgn = lds + L.eqsgn;
    S.scal = lds + L.scal;
    S.act = reinterpret_cast<unsigned char*>(lds + L.act);
    S.iact = reinterpret_cast<int*>(lds + L.iact);
    S.ricd = lds + L.ricD;
    S.ricxi = nullptr;
    return S;
}

// ------------------------------------------------------------------------------------------------
// Factorisation: S.J holds the Hessian (upper triangle), S.cvec the linear term c.
// On exit the upper triangle of S.J holds the Cholesky factor R (Q = R'R), S.coef holds 1/R(i,i) and S.xs = -Q^-1 c.
// Returns 0 or 2.  gi_invert() then turns R into J = R^-1 when (and only when) the active-set loop needs it.
// ------------------------------------------------------------------------------------------------
COPRA_DEV int rcol(int c) { return c * (c + 1) / 2; }

// where entry (i, j) of the factor lives: square storage with leading dimension ld, or (TRI) the packed upper triangle.
// Both are bank-conflict free for "lane = column" and "lane = row" accesses (c -> c (c + 1) / 2 is a bijection modulo
// 32 on any 32 consecutive columns).  In the packed form an (i, j) below the diagonal names some other entry of the
// triangle: the factorisation loops read such entries on lanes whose result is discarded, never write them.
template <bool TRI>
COPRA_DEV int fidx(int i, int j, int ld)
{
    return TRI ? rcol(j) + i : i * ld + j;
}

// Two-level left-looking Cholesky: before the first 4-row panel of every 16-row super-panel, rows [k0, k0+16) x
// columns [k0, n) of the Hessian receive the contribution of ALL finished rows 0..k0-1 at once,
//     H(k0+i, c) -= sum_{t<k0} R(t, k0+i) R(t, c),
// as 16x16x4 FP64 matrix-core products: per four rows t one lane-private LDS read feeds the A operand and one per column
// tile the B operand -- no wave-uniform (broadcast) reads at all, which is what the one-level loop spends its time on
// (5 LDS reads per 4 FMAs; 105 of its 4-row blocks at n = 60 become 24, plus 40 MFMAs).  The panels of the super-panel
// then only sweep the rows of their own super-panel.  (MI355X: the FP64 MFMA rate equals the VALU FMA rate; the gain
// is the operand traffic, not the arithmetic.)
#ifndef COPRA_CHOL_SUPER
#define COPRA_CHOL_SUPER 1
#endif
// K0 > 0 (with NV > 0): the super-panel's first row at compile time -- trip count and tile count are constants, the
// loop unrolls and every operand is on its way before the first product issues.
template <int NV, bool TRI, int K0 = 0>
COPRA_DEV void chol_super_update(double* J, int n, int ld, int k0_rt)
{
    const int lane = lane_id();
    const int k0 = K0 ? K0 : k0_rt;
    const int kk = lane >> 4, col = lane & 15;
    const int ra = k0 + col; // the row of the super-panel this lane feeds into the A operand
    const int rac = (ra < n) ? ra : n - 1;
    const int tile0 = k0 >> 4;
    // k0 >= 16 and n <= 64: at most the column tiles 1..3
    constexpr int kMaxTiles = (NV > 0 && K0 > 0) ? ((NV + 15) / 16 - K0 / 16) : 3;
    const int ntile = ((n + 15) >> 4) - tile0;
    mfma_acc acc[kMaxTiles];
#pragma unroll
    for (int c = 0; c < kMaxTiles; ++c) acc[c].v[0] = acc[c].v[1] = acc[c].v[2] = acc[c].v[3] = 0.0;
    int cb[kMaxTiles];
    bool cin[kMaxTiles];
#pragma unroll
    for (int c = 0; c < kMaxTiles; ++c) {
        const int cc = 16 * (tile0 + c) + col;
        cin[c] = (c < ntile) && (cc < n);
        cb[c] = cin[c] ? cc : n - 1;
    }
#pragma clang loop unroll_count(K0 > 0 ? 16 : 1)
    for (int t0 = 0; t0 < k0; t0 += 4) {
        double a = J[fidx<TRI>(t0 + kk, rac, ld)];
        if (ra >= n) a = 0.0;
        double b[kMaxTiles];
#pragma unroll
        for (int c = 0; c < kMaxTiles; ++c) {
            b[c] = J[fidx<TRI>(t0 + kk, cb[c], ld)];
            if (!cin[c]) b[c] = 0.0;
        }
#pragma unroll
        for (int c = 0; c < kMaxTiles; ++c)
            if (c < ntile) mfma_f64_16x16x4(a, b[c], acc[c]); // (ntile is wave-uniform)
    }
    // every entry has one owner: all reads first, then all writes (the LDS reads batch)
    double h[kMaxTiles][4];
#pragma unroll
    for (int c = 0; c < kMaxTiles; ++c) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int i = k0 + kk + 4 * reg, cc = 16 * (tile0 + c) + col;
            const bool mine = c < ntile && i < n && cc < n && cc >= i;
            h[c][reg] = J[fidx<TRI>(mine ? i : 0, mine ? cc : 0, ld)];
        }
    }
#pragma unroll
    for (int c = 0; c < kMaxTiles; ++c) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int i = k0 + kk + 4 * reg, cc = 16 * (tile0 + c) + col;
            if (c < ntile && i < n && cc < n && cc >= i) J[fidx<TRI>(i, cc, ld)] = h[c][reg] - acc[c].v[reg];
        }
    }
    wave_sync();
}

template <int NV, bool TRI = false>
COPRA_DEV int gi_factorize(const SolverLds& S, int n_rt, long long* t_chol COPRA_FINE_ARGS)
{
    const int lane = lane_id();
    const int n = NV ? NV : n_rt;
    const int ld = NV ? (NV | 1) : S.ldj;
    const int lj = (lane < n) ? lane : n - 1; // clamped lane: idle lanes re-read column n-1, never store
    double* J = S.J;
    double* rinvd = S.coef;
    wave_sync();
    // ---- blocked left-looking Cholesky Q = R'R, lane = column (qpgen2: dpofa) ----
    // (compile-time shapes: four panels per trip -- the inner loops over the finished rows of the super-panel get constant
    //  trip counts and the packed-triangle offsets fold; Cholesky 36.4 k -> 32.0 k cycles at n = 60)
#pragma clang loop unroll_count(NV > 0 ? 4 : 1)
    for (int k0 = 0; k0 < n; k0 += 4) {
        const int pw = (n - k0 < 4) ? n - k0 : 4;
        int tfirst = 0;
        if constexpr (COPRA_CHOL_SUPER && kWave == 64) { // (the packed builds have no 64-lane matrix-core operand)
            if (n > 16) {
                tfirst = k0 & ~15;
                if (k0 > 0 && k0 == tfirst) {
                    if constexpr (NV > 16) { // (compile-time shapes: one instantiation per super-panel)
                        if (k0 == 16)
                            chol_super_update<NV, TRI, 16>(J, n, ld, k0);
                        else if (k0 == 32)
                            chol_super_update<NV, TRI, (NV > 32 ? 32 : 0)>(J, n, ld, k0);
                        else
                            chol_super_update<NV, TRI, (NV > 48 ? 48 : 0)>(J, n, ld, k0);
                    } else {
                        chol_super_update<NV, TRI>(J, n, ld, k0);
                    }
                }
            }
        }
        double acc[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p] = (p < pw) ? J[fidx<TRI>(k0 + p, lj, ld)] : 0.0;
#pragma unroll 2
        for (int t = tfirst; t < k0; t += 4) { // k0 is a multiple of 4: no remainder
            double rt[4], bb[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                rt[u] = J[fidx<TRI>(t + u, lj, ld)]; // R(t+u, lane)
#pragma unroll
                for (int p = 0; p < 4; ++p) bb[u][p] = (p < pw) ? J[fidx<TRI>(t + u, k0 + p, ld)] : 0.0; // wave-uniform address
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[p] -= bb[u][p] * rt[u];
        }
        if (k0 == 28 || k0 == 56) COPRA_FINE("chol:tloop");
        // The 4x4 diagonal block of the panel (entries acc[p] of lanes k0+p..k0+3) is broadcast to every lane and
        // factorised redundantly (wave-uniform arithmetic, no cross-lane dependency inside the chain).
        double Dg[4][4], Rd[4][4], ri[4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = p; q < 4; ++q) Dg[p][q] = (q < pw) ? bcast_f64(acc[p], k0 + q) : ((p == q) ? 1.0 : 0.0);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            double dpp = Dg[p][p];
#pragma unroll
            for (int t = 0; t < p; ++t) dpp -= Rd[t][p] * Rd[t][p];
            if (p < pw && !(dpp > 0.0)) return 2; // "Problems with the decomposition of Q" (QuadProgSolver.h:25)
            ri[p] = fast_rsqrt(dpp);
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                double v = Dg[p][q];
#pragma unroll
                for (int t = 0; t < p; ++t) v -= Rd[t][p] * Rd[t][q];
                Rd[p][q] = v * ri[p];
            }
        }
        // this lane's entries of the four new rows of R
        double rp[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            double v = acc[p];
#pragma unroll
            for (int t = 0; t < p; ++t) v -= Rd[t][p] * rp[t];
            rp[p] = v * ri[p]; // R(k0+p, lane); the diagonal lane gets sqrt(pivot)
            // (factor-only layout: nothing ever reads the diagonal of R itself, so its slot carries 1 / R(i,i))
            if (p < pw && lane >= k0 + p && lane < n) J[fidx<TRI>(k0 + p, lane, ld)] = (TRI && lane == k0 + p) ? ri[p] : rp[p];
            if (!TRI && p < pw && lane == 0) rinvd[k0 + p] = ri[p];
        }
        wave_sync();
        if (k0 == 24 || k0 == 28 || k0 == 52 || k0 == 56) COPRA_FINE("chol:panel");
    }
    if (t_chol) *t_chol = cycle_counter();
    // ---- unconstrained minimiser: R'R x = -c by two substitutions (qpgen2: dposl).  J = R^-1 is NOT formed here:
    //      an instance whose unconstrained minimiser violates nothing never needs it (gi_invert is called lazily). ----
    {
        // forward  R' y = -c :  y_k = acc_k / R(k,k),  acc_j -= R(k,j) y_k  (j > k), lane = column
        double acc = (lane < n) ? -S.cvec[lj] : 0.0;
        double yk = 0.0;
        for (int k0 = 0; k0 < n; k0 += 4) {
            double row[4], ri[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = (k0 + u < n) ? k0 + u : n - 1;
                row[u] = J[fidx<TRI>(k, lj, ld)]; // R(k, lane)
                ri[u] = TRI ? J[fidx<true>(k, k, ld)] : rinvd[k];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u;
                if (k < n) {
                    const double y = bcast_f64(acc, k) * ri[u];
                    if (lane > k) acc -= row[u] * y;
                }
            }
        }
        // (lane k's accumulator is final once step k-1 is done: y_k = acc_k / R(k,k), no select inside the loop)
        const double rinv_own = TRI ? J[fidx<true>(lj, lj, ld)] : rinvd[lj];
        yk = acc * rinv_own;
        // backward  R x = y :  x_k = acc_k / R(k,k),  acc_i -= R(i,k) x_k  (i < k), lane = row
        acc = yk;
        double xk = 0.0;
        for (int k0 = n - 1; k0 >= 0; k0 -= 4) {
            double colv[4], ri[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = (k0 - u >= 0) ? k0 - u : 0;
                colv[u] = J[fidx<TRI>(lj, k, ld)]; // R(lane, k)
                ri[u] = TRI ? J[fidx<true>(k, k, ld)] : rinvd[k];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 - u;
                if (k >= 0) {
                    const double x = bcast_f64(acc, k) * ri[u];
                    if (lane < k) acc -= colv[u] * x;
                }
            }
        }
        xk = acc * rinv_own;
        if (lane < n) S.xs[lane] = xk;
        wave_sync();
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// J = R^-1 in place (upper triangular, strict lower part zero).  S.coef still holds 1/R(i,i) from gi_factorize.
// ------------------------------------------------------------------------------------------------
template <int NV>
COPRA_DEV void gi_invert(const SolverLds& S, int n_rt)
{
    const int lane = lane_id();
    const int n = NV ? NV : n_rt;
    const int ld = NV ? (NV | 1) : S.ldj;
    const int lj = (lane < n) ? lane : n - 1;
    double* J = S.J;
    const double* rinvd = S.coef;
    // zero the strict lower triangle (qpgen2 does the same before the first rotation)
    for (int i = 1; i < n; ++i)
        if (i > lane) J[i * ld + lane] = 0.0;
    wave_sync();
    // ---- in-place inverse of the upper-triangular factor, 4 rows at a time from the bottom (qpgen2: dpori) ----
#pragma unroll 1
    for (int i0 = ((n - 1) / 4) * 4; i0 >= 0; i0 -= 4) {
        const int pw = (n - i0 < 4) ? n - i0 : 4;
        double acc[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p] = (lane == i0 + p) ? 1.0 : 0.0;
        int k = i0 + pw;
#pragma unroll 2
        for (; k + 4 <= n; k += 4) {
            double vk[4], bb[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                vk[u] = J[(k + u) * ld + lj]; // V(k+u, lane), zero below the diagonal
#pragma unroll
                for (int p = 0; p < 4; ++p) bb[u][p] = (p < pw) ? J[(i0 + p) * ld + k + u] : 0.0; // wave-uniform address
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[p] -= bb[u][p] * vk[u];
        }
        for (; k < n; ++k) {
            const double vk = J[k * ld + lj];
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (p < pw) acc[p] -= J[(i0 + p) * ld + k] * vk;
        }
        double v[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
        for (int p = 3; p >= 0; --p) {
            if (p < pw) {
                double a = acc[p];
#pragma unroll
                for (int q = p + 1; q < 4; ++q)
                    if (q < pw) a -= J[(i0 + p) * ld + i0 + q] * v[q];
                v[p] = a * rinvd[i0 + p];
            }
        }
        wave_sync(); // every lane has read the panel rows
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (p < pw && lane < n) J[(i0 + p) * ld + lane] = v[p];
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------------
// Active-set iterations.  Returns qpgen2's ierr (0 ok, 1 infeasible), 3 (iteration cap) or 4 (internal: the
// compact layout's R is full; the caller queues the instance for the full-layout launch).
// ------------------------------------------------------------------------------------------------
//
// TRI (factor-only layout): J = R^-1 is never formed.  With N the active normals, R^-T N = Q1 Rq (Q1 orthonormal, n x
// nact; Rq is S.R) replaces the first nact columns of qpgen2's J = R^-1 [Q1 Q2]:
//     d1 = Q1' w,  w = R^-T n+ (forward substitution);   z = J2 J2' n+ = R^-1 (w - Q1 d1) (back substitution);
//     adding a constraint appends (w - Q1 d1) / |.| to Q1 and [d1; |.|] to Rq -- no sweep over an n x n matrix;
//     dropping one rotates columns of Q1 exactly as it rotates those of J.
// Same pivoting rule, same steps, the same iterates up to rounding; 15 KB of LDS per instance instead of 29 KB at
// n = 60 and no triangular inverse (48 k cycles) for the instances that activate a constraint.
// QR > 0 (with TRI): the columns of Q1 live in REGISTERS -- every access to Q1 is lane-private (lane j holds row j of each
// column), so QR columns cost 2 QR VGPRs per lane and no LDS: 320 doubles less per instance at QR = 5, which is what lets
// an eighth instance share a CU at the headline shape (LdsLayout::q1regs).  S.rcap <= QR then.
// Warm start (warm_list != nullptr; receding-horizon ticks of the shared-model path): the rows of warm_list -- the active
// set of the previous tick, shifted by one step -- are the FIRST CANDIDATES of step 1.  The dual method may pick any violated
// constraint (qpgen2 takes the most violated one, a heuristic), so while the list lasts the next candidate is simply its next
// row that is violated at the current iterate: ONE slack evaluation instead of a scan over all rows, then the ordinary
// iteration (dual blocking test, drops) -- every invariant of the method holds throughout, nothing is ever restarted.
// After the list the ordinary scans take over and finish.  Same optimum as a cold start; the iterates differ.
// a row policy may want to know that the iterate moved by t z (StageRows::moved); the others do not have the member
template <class R>
COPRA_DEV auto rows_moved(const R& r, double t, int) -> decltype(r.moved(t), void()) { r.moved(t); }
template <class R>
COPRA_DEV void rows_moved(const R&, double, long) { }

// RNX, RNU > 0 (with TRI and a compile-time shape): the factor in S.J is in Riccati form (ric_factor.hpp) -- the two
// substitutions become the closed-loop recursions over the NV / RNU stages; everything else is unchanged.
template <int NV, bool TRI = false, int QR = 0, int RNX = 0, int RNU = 0, class Rows>
COPRA_DEV int gi_active_set(const SolverLds& S, int n_rt, int meq, int mgen, Rows& rows, double vsmall, int max_iter,
    int& iter_main, int& iter_drop COPRA_FINE_ARGS, bool j_ready = false, int warm_mine = -1, int warm_n = 0,
    int* nact_out = nullptr)
{
    // warm_mine: lane k holds entry k of the warm list (-1: none), warm_n entries in all (0: cold start)
    double q1r[QR > 0 ? QR : 1]; // this lane's element of every Q1 column (QR > 0)
#pragma unroll
    for (int u = 0; u < (QR > 0 ? QR : 1); ++u) q1r[u] = 0.0;
    // (accessed only through fully unrolled, compile-time-indexed code below -- no lambdas, no address taken -- so that the
    //  array stays in registers: a first version went through closures and the compiler kept it in scratch memory, 24 MB of
    //  extra HBM writes per launch)
#define COPRA_Q1_GET(dst, kk)                                                                                         \
    do {                                                                                                              \
        if constexpr (QR > 0) {                                                                                       \
            dst = q1r[0];                                                                                             \
            _Pragma("unroll") for (int u_ = 1; u_ < QR; ++u_) dst = (u_ == (kk)) ? q1r[u_] : dst;                    \
        } else {                                                                                                      \
            dst = S.Q1[(kk) * kWave + lane_id()];                                                                     \
        }                                                                                                             \
    } while (0)
#define COPRA_Q1_SET(kk, val)                                                                                         \
    do {                                                                                                              \
        if constexpr (QR > 0) {                                                                                       \
            const double v_ = (val);                                                                                  \
            _Pragma("unroll") for (int u_ = 0; u_ < QR; ++u_) q1r[u_] = (u_ == (kk)) ? v_ : q1r[u_];                 \
        } else {                                                                                                      \
            S.Q1[(kk) * kWave + lane_id()] = (val);                                                                   \
        }                                                                                                             \
    } while (0)
    const int lane = lane_id();
    const int n = NV ? NV : n_rt;
    const int ld = NV ? (NV | 1) : S.ldj;
    const int lj = (lane < n) ? lane : n - 1;
    const int mtotal = mgen + 2 * n; // QuadProgSolver.cpp:51: the bounds are 2n more inequality rows
    double* J = S.J;
    int nact = 0;
    bool have_J = j_ready; // the shared-model path arrives with J = R^-1 in place
    iter_main = 0;
    iter_drop = 0;
    if constexpr (RNX > 0) { // (layout_lds_ric: the flags start on a double and own whole doubles -- eight rows per store)
        double* const act8 = reinterpret_cast<double*>(S.act);
        for (int i = lane; i < (mtotal + 7) / 8; i += kWave) act8[i] = 0.0;
    } else {
        for (int i = lane; i < mtotal; i += kWave) S.act[i] = 0;
    }
    for (int i = lane; i < meq; i += kWave) S.eqsgn[i] = 1.0;
    for (int i = lane; i <= S.rcap + 1 && i <= n + 1; i += kWave) S.uv[i] = 0.0;
    // rows [I; -I] with right-hand sides [XU; -XL] (QuadProgSolver.cpp:61-69): lane j keeps XU_j, XL_j
    const double ubj = rows.ub(lj), lbj = rows.lb(lj);
    // XL_j == XU_j (the default x0 bounds of InitialStateLMPC, InitialStateLMPC.cpp:20-28): once one row of the pair is
    // active the other one is its negative -- linearly dependent, its slack is rounding noise and never a violation
    // (qpgen2 would otherwise try to add it, find no step and report "no solution" whenever the noise is negative)
    const bool pinned = (ubj - lbj) <= 1e-12 * fmax(1.0, fabs(ubj)); // (an interval narrower than the noise counts as one)
    wave_sync();

    int warm_i = 0; // next entry of warm_list
    bool forced = false; // the current candidate comes from the warm list (no scan needed)
    for (;;) {
        if (iter_main >= max_iter) return 3;
        iter_main += 1;
        // ---------------- step 1: most violated constraint ----------------
        if (iter_main <= 1) COPRA_FINE("as:init");
        rows.begin_scan(S.xs);
        double best = 0.0, best_s = 0.0;
        int best_i = -1;
        forced = false;
        while (warm_i < warm_n && best_i < 0) { // the next row of the warm list that is violated here
            const int p = bcast_i32(warm_mine, warm_i++);
            if (p < meq || p >= mtotal || S.act[p]) continue; // (gaps are -1; equality rows are left to the ordinary scans)
            double s;
            if (p < mgen) {
                s = rows.slack_uniform(p, S.xs);
            } else {
                const int q = p - mgen;
                const double sl = (q < n) ? ubj - S.xs[lj] : S.xs[lj] - lbj;
                s = bcast_f64(sl, (q < n) ? q : q - n);
            }
            if (fabs(s) < vsmall) s = 0.0;
            if (!(s < 0.0)) continue; // satisfied at the current iterate: not a candidate
            best_i = p;
            best_s = s;
            forced = true;
        }
        if (!forced) {
        for (int base = 0; base < mgen; base += kWave) { // general rows (equalities first)
            const int i = base + lane;
            if (i < mgen) {
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
        if (iter_main <= 1) COPRA_FINE("as:rows");
        if (lane < n) { // bound rows: unit norm; an infinite / DBL_MAX bound gives a slack that is never negative
            const double xj = S.xs[lane];
            double s = ubj - xj; // row mgen + j of [I]
            if (fabs(s) < vsmall) s = 0.0;
            if (S.act[mgen + lane] || (pinned && S.act[mgen + n + lane])) s = 0.0;
            if (s < best) {
                best = s;
                best_i = mgen + lane;
                best_s = s;
            }
            s = xj - lbj; // row mgen + n + j of [-I]
            if (fabs(s) < vsmall) s = 0.0;
            if (S.act[mgen + n + lane] || (pinned && S.act[mgen + lane])) s = 0.0;
            if (s < best) {
                best = s;
                best_i = mgen + n + lane;
                best_s = s;
            }
        }
        if (iter_main <= 1) COPRA_FINE("as:bounds");
        wave_argmin(best, best_i, best_s);
        } // (!forced)
        if (iter_main <= 2) COPRA_FINE("as:scan");
        const int nvl = best_i;
        if (nvl < 0) { // optimal
            if (nact_out) *nact_out = nact;
            return 0;
        }
        if (TRI && !have_J && S.Jsrc) { // shared model, factor-only: the batch-wide factor comes into LDS on first need
            wave_sync();
            for (int e = lane; e < n * (n + 1) / 2; e += kWave) S.J[e] = S.Jsrc[e];
            wave_sync();
            if (lane < n) S.J[fidx<true>(lane, lane, ld)] = S.rinv_src[lane]; // (the diagonal slots carry 1 / R(i,i))
            have_J = true;
            wave_sync();
        }
        if (!TRI && !have_J) { // first violated constraint: only now is J = R^-1 needed
            if (S.Jsrc) { // shared model: the batch-wide J is in HBM / L2, take a private copy (the updates rotate it)
                wave_sync();
                for (int e = lane; e < n * ld; e += kWave) S.J[e] = S.Jsrc[e];
                wave_sync();
            } else {
                gi_invert<NV>(S, n);
            }
            have_J = true;
        }
        double sv_nvl = best_s;
        wave_sync(); // eqsgn updates visible

        // ---------------- step 2 ----------------
        for (;;) {
            int inj_stage = 0, inj_comp = 0; // Riccati form, compact variant: the state part of the normal as  inj_val e_comp  at stage inj_stage
            double inj_val = 0.0;
            if (nvl < mgen) {
                const double sgn = (nvl < meq) ? S.eqsgn[nvl] : 1.0;
                if constexpr (RNX > 0)
                    rows.load_normal_split(nvl, sgn, S.ap, inj_stage, inj_comp, inj_val);
                else
                    rows.load_normal(nvl, sgn, S.ap);
            } else if (lane < n) { // rows of -[I; -I]: -e_j for an upper bound, +e_j for a lower bound
                const int q = nvl - mgen;
                S.ap[lane] = (q < n) ? ((lane == q) ? -1.0 : 0.0) : ((lane == q - n) ? 1.0 : 0.0);
            }
            wave_sync();
            if (iter_main <= 1) COPRA_FINE("as:normal");
            double dj, zi;
            double vj = 0.0; // TRI: component `lane` of w - Q1 d1
            double napl = 0.0; // TRI, Riccati form: component `lane` of n+
            if constexpr (TRI) {
                // w = R^-T n+ : forward substitution, lane = column; a bound row's normal starts at its own index
                double acc = (lane < n) ? S.ap[lj] : 0.0;
                double wk = 0.0;
                napl = acc;
                double rinv_own = 0.0;
                if constexpr (RNX > 0) {
                    // stages past the last non-zero of the normal contribute nothing (a row at step k has no component beyond
                    // stage k - 1): the backward recursion starts there.  In place: this lane's n is in `acc` / `napl`.
                    const double last = wave_max((lane < n && acc != 0.0) ? (double)lane : 0.0);
                    const int nst = uniform_i32((int)last / RNU + 1);
                    // (NV == 0: the horizon n / RNU is a run-time value)
                    wk = ric_apply_mfma4<RNX, RNU, NV / RNU, true>(J, S.ap, S.ap, S.ricd, nst > inj_stage ? nst : inj_stage, nullptr, inj_stage, inj_comp, inj_val,
                        n / RNU);
                } else {
                auto forward = [&](int kfirst) {
                    for (int k0 = kfirst; k0 < n; k0 += 4) {
                        double row[4], ri4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int k = (k0 + u < n) ? k0 + u : n - 1;
                            row[u] = J[fidx<true>(k, lj, ld)];
                            ri4[u] = J[fidx<true>(k, k, ld)];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int k = k0 + u;
                            if (k < n) {
                                const double y = bcast_f64(acc, k) * ri4[u];
                                if (lane > k) acc -= row[u] * y;
                            }
                        }
                    }
                };
                // A bound row's normal starts at its own index, so its sweep could start there -- but a run-time start
                // keeps the loop from unrolling, and the unrolled sweep (offsets folded, loads batched) is twice as
                // fast as the rolled one (5.5 k vs 10.4 k cycles at n = 60): compile-time shapes always sweep from 0.
                if (NV > 0 || nvl < mgen)
                    forward(0);
                else
                    forward(((nvl - mgen) % n) & ~3);
                rinv_own = J[fidx<true>(lj, lj, ld)];
                wk = (lane < n) ? acc * rinv_own : 0.0; // (lane k's accumulator is final once step k-1 is done)
                }
                if (iter_main <= 1) COPRA_FINE("as:w");
                // d1 = Q1' w and v = w - Q1 d1, by modified Gram-Schmidt, twice (keeps Q1 orthonormal to rounding)
                vj = wk;
                dj = 0.0;
                for (int pass = 0; pass < 2; ++pass) {
                    if constexpr (QR > 0) {
#pragma unroll
                        for (int k = 0; k < QR; ++k) {
                            if (k < nact) {
                                const double qk = q1r[k];
                                const double e = wave_sum(qk * vj);
                                vj -= e * qk;
                                if (lane == k) dj += e;
                            }
                        }
                    } else {
                        for (int k = 0; k < nact; ++k) {
                            const double qk = S.Q1[k * kWave + lane];
                            const double e = wave_sum(qk * vj);
                            vj -= e * qk;
                            if (lane == k) dj += e;
                        }
                    }
                }
                if (iter_main <= 1) COPRA_FINE("as:d");
                // z = R^-1 v : back substitution, lane = row
                acc = vj;
                double zk = 0.0;
                if constexpr (RNX > 0) {
                    if (lane < n) S.ap[lane] = vj;
                    wave_sync();
                    zk = ric_apply_mfma4<RNX, RNU, NV / RNU, false>(J, S.ap, S.ap, S.ricd, n / RNU, S.ricxi, 0, 0, 0.0, n / RNU);
                } else {
                for (int k0 = n - 1; k0 >= 0; k0 -= 4) {
                    double colv[4], ri4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = (k0 - u >= 0) ? k0 - u : 0;
                        colv[u] = J[fidx<true>(lj, k, ld)];
                        ri4[u] = J[fidx<true>(k, k, ld)];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = k0 - u;
                        if (k >= 0) {
                            const double x = bcast_f64(acc, k) * ri4[u];
                            if (lane < k) acc -= colv[u] * x;
                        }
                    }
                }
                zk = acc * rinv_own;
                }
                zi = (lane < n) ? zk : 0.0;
                if (iter_main <= 1) COPRA_FINE("as:z");
            } else {
            // d = J' n+   (lane = column)
            {
                double d0 = 0.0, d1 = 0.0;
                int i = 0;
                for (; i + 1 < n; i += 2) {
                    d0 += J[i * ld + lj] * S.ap[i];
                    d1 += J[(i + 1) * ld + lj] * S.ap[i + 1];
                }
                if (i < n) d0 += J[i * ld + lj] * S.ap[i];
                dj = (lane < n) ? d0 + d1 : 0.0;
            }
            if (lane < n) S.dv[lane] = (lane >= nact) ? dj : 0.0; // d2 (d1 stays in registers)
            wave_sync();
            if (iter_main <= 1) COPRA_FINE("as:d");
            // z = J2 d2   (lane = row; the d1 part of dv is zero)
            {
                double z0 = 0.0, z1 = 0.0;
                int j = 0;
                for (; j + 1 < n; j += 2) {
                    z0 += J[lj * ld + j] * S.dv[j];
                    z1 += J[lj * ld + j + 1] * S.dv[j + 1];
                }
                if (j < n) z0 += J[lj * ld + j] * S.dv[j];
                zi = (lane < n) ? z0 + z1 : 0.0;
            }
            if (iter_main <= 1) COPRA_FINE("as:z");
            }
            // r = R^-1 d1 : column-oriented back substitution, r_c broadcast from lane c
            double acc = (lane < nact) ? dj : 0.0;
            double ri = 0.0;
            for (int c = nact - 1; c >= 0; --c) {
                double rc = 0.0;
                if (lane == c) rc = acc / S.R[rcol(c) + c];
                rc = bcast_f64(rc, c);
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
            if (nact > 0) {
                double dummy = 0.0;
                wave_argmin(t1, it1, dummy);
            }
            const bool t1inf = (it1 < 0);
            if (iter_main <= 1) COPRA_FINE("as:t1");
            const double zz = wave_sum(zi * zi);
            bool drop = false;
            if (fabs(zz) <= vsmall) {
                // no step in primal space
                if (t1inf) return 1; // infeasible
                if (lane < nact) S.uv[lane] -= t1 * ri;
                if (lane == 0) S.uv[nact] += t1;
                drop = true;
            } else {
                // (Riccati-factor tier: S.ap was the hand-over buffer of the two recursions; n+ is in `napl`)
                double zn = wave_sum((lane < n) ? zi * (RNX > 0 ? napl : S.ap[lane]) : 0.0);
                if constexpr (RNX > 0) { // the state part of the normal did not go through S.ap: n'z = (Psi z)(row) = a closed-loop state of z
                    if (inj_stage > 0) zn += inj_val * S.ricxi[inj_stage * RNX + inj_comp];
                }
                double tt = -sv_nvl / zn;
                bool t2min = true;
                if (!t1inf && t1 < tt) {
                    tt = t1;
                    t2min = false;
                }
                if (lane < n) S.xs[lane] += tt * zi;
                rows_moved(rows, tt, 0);
                if (lane < nact) S.uv[lane] -= tt * ri;
                if (lane == 0) S.uv[nact] += tt;
                if (iter_main <= 1) COPRA_FINE("as:step");
                if (t2min) {
                    // ---- full step: constraint nvl becomes active; update R and J ----
                    if (nact >= S.rcap) return 4; // compact layout: R is full -> the instance is redone with the full one
                    if (lane < nact) S.R[rcol(nact) + lane] = dj;
                    if constexpr (TRI) {
                        const double h = sqrt(wave_sum(vj * vj));
                        COPRA_Q1_SET(nact, vj / h); // (lanes >= n hold 0)
                        if (lane == nact) {
                            S.R[rcol(nact) + nact] = h;
                            S.iact[nact] = nvl;
                            S.act[nvl] = 1;
                        }
                        nact += 1;
                        wave_sync();
                        if (iter_main <= 1) COPRA_FINE("as:givens");
                        break; // back to step 1
                    }
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
                    // rotation q acts on columns (q-1, q), q = nact+1 .. n-1, as the reflection
                    //   col_{q-1}' = gc col_{q-1} + gs col_q ,  col_q' = gs col_{q-1} - gc col_q
                    // (qpgen2's "nu" form is the same map); identity where qpgen2 skips (h_q == 0 or gc == 1)
                    if (lane >= 1 && lane < n) {
                        double c0 = 1.0, c1 = 0.0, c2 = 0.0, c3 = 1.0;
                        if (lane > nact && h != 0.0) {
                            const double gc = d_prev / h_prev;
                            const double gs = h / h_prev;
                            if (gc != 1.0) {
                                c0 = gc;
                                c1 = gs;
                                c2 = gs;
                                c3 = -gc;
                            }
                        }
                        S.coef[4 * lane + 0] = c0;
                        S.coef[4 * lane + 1] = c1;
                        S.coef[4 * lane + 2] = c2;
                        S.coef[4 * lane + 3] = c3;
                    }
                    if (lane == nact) {
                        S.R[rcol(nact) + nact] = h; // new diagonal element of R
                        S.iact[nact] = nvl;
                        S.act[nvl] = 1;
                    }
                    wave_sync();
                    if (iter_main <= 1) COPRA_FINE("as:coef");
                    if (nact + 1 < n) {
                        double carry = J[lj * ld + (n - 1)];
                        const int qlo = NV ? 0 : nact; // compile-time shape: sweep everything (identity below nact)
                        int q = n - 1;
                        // four rotations per round trip: all LDS reads of a group are issued before its first store
                        // (the compiler cannot move loads across the stores by itself: everything aliases in LDS)
                        for (; q - 3 > qlo; q -= 4) {
                            double a[4], c[4][4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                a[u] = J[lj * ld + q - 1 - u];
#pragma unroll
                                for (int v = 0; v < 4; ++v) c[u][v] = S.coef[4 * (q - u) + v];
                            }
                            double w[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const double t = c[u][0] * a[u] + c[u][1] * carry;
                                w[u] = c[u][2] * a[u] + c[u][3] * carry;
                                carry = t;
                            }
                            if (lane < n) {
#pragma unroll
                                for (int u = 0; u < 4; ++u) J[lane * ld + q - u] = w[u];
                            }
                        }
                        for (; q > qlo; --q) {
                            const double a = J[lj * ld + q - 1];
                            const double c0 = S.coef[4 * q + 0], c1 = S.coef[4 * q + 1];
                            const double c2 = S.coef[4 * q + 2], c3 = S.coef[4 * q + 3];
                            const double t = c0 * a + c1 * carry;
                            const double w = c2 * a + c3 * carry;
                            if (lane < n) J[lane * ld + q] = w;
                            carry = t;
                        }
                        if (lane < n) J[lane * ld + qlo] = carry;
                    }
                    nact += 1;
                    wave_sync();
                    if (iter_main <= 1) COPRA_FINE("as:givens");
                    break; // back to step 1
                } else {
                    // ---- partial step: recompute the slack of nvl, then drop the blocking constraint ----
                    wave_sync();
                    double s;
                    if (nvl < mgen) {
                        rows.begin_scan(S.xs);
                        s = rows.slack_uniform(nvl, S.xs); // every lane computes the same value
                    } else {
                        const int q = nvl - mgen;
                        const double sl = (q < n) ? ubj - S.xs[lj] : S.xs[lj] - lbj;
                        s = bcast_f64(sl, (q < n) ? q : q - n);
                    }
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
                        if constexpr (TRI) {
                            double x, y;
                            COPRA_Q1_GET(x, q);
                            COPRA_Q1_GET(y, q + 1);
                            const double t = gc * x + gs * y;
                            COPRA_Q1_SET(q + 1, nu_ * (x + t) - y);
                            COPRA_Q1_SET(q, t);
                        } else if (lane < n) {
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
