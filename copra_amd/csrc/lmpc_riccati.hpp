// lmpc_riccati.hpp -- long-horizon LMPC / InitialStateLMPC WITHOUT condensing: a primal-dual interior-point method whose
// Newton systems are solved by a Riccati recursion over the stages.  ONE instance per 64-lane
// wavefront, persistent grid over the batch; the stage plan (stage_plan.hpp) is shared by the batch.
//
// What it replaces, for controllers with more than 64 decision variables whose pieces are all stage-wise
// (stage_plan.hpp): the whole call stack of LMPC::solve (src/LMPC.cpp:79-101) --
//   PreviewSystem::updateSystem + every cost->update / constraint->update + makeQPForm + QuadProgDense
// The reference builds Psi (X x n), the n x n Hessian and hands the dense QP to Goldfarb-Idnani: at BASELINE config 5
// (nx=12, nu=6, N=50, InitialStateLMPC) that is a 312 x 312 inverse factor swept ~1000 times per solve (600 additions,
// 440 drops).  Here the same optimum is the fixed point of ~20 Newton steps, each ONE backward factorisation sweep over
// the 50 stages (18 x 18 blocks) plus three vector sweeps: O(N (nx+nu)^3) per step, nothing of size n x n exists.
// The QP is strictly convex, so the optimum is unique: U, X, x0* equal the reference's to solver accuracy (measured
// against 60-digit truth vectors, tests/golden/config5_truth.npz: 5e-15 in the numpy prototype of this algorithm, where
// the CPU Goldfarb-Idnani path is 1e-5 away); iteration counts are NOT comparable (status: 0, iter = (Newton steps, 0)).
// An instance that does not converge (infeasible problems make an interior-point method stall) is queued for the
// condensed Goldfarb-Idnani kernel, which reports the reference's status code for it.
//
// Algorithm (Mehrotra predictor-corrector on  min sum_k 1/2 z_k'W_k z_k + q_k'z_k  s.t. dynamics, rows a_i'z <= f_i):
//   slacks s, multipliers lam per inequality row; D = lam / s; equality rows by a proximal multiplier iteration
//   (weight 1/delta on the row, nu += residual / delta);
//   stage Hessian  H_k = W_k + sum_i D_i a_i a_i',  gradient  g_k = W_k z_k + q_k + sum_i a_i c_i;
//   backward:  M = H_k + [A B]' P_{k+1} [A B],  h = g_k + [A B]' p_{k+1};   K = -Muu^-1 Mux,  kv = -Muu^-1 hu;
//              P_k = Mxx + Mux' K,  p_k = hx + Mux' kv;     forward:  du = K dx + kv,  dx+ = A dx + B du.
//   The iterate always satisfies the dynamics (the steps do), so no costates are carried.
// InitialStateLMPC (src/InitialStateLMPC.cpp:77-122): the reference's objective is
//   1/2 U'QU + x0'E U + f'U + 1/2 x0'(R + E Q^-1 E')x0 + r'x0,  while the stage-wise cost is
//   1/2 U'QU + (E'x0 + f)'U + 1/2 x0'S x0 + g0'x0;  the difference  1/2 x0'(R - P0) x0 + (r - g0)'x0  with
//   P0 = S - E Q^-1 E' (the UNCONSTRAINED cost-to-go Hessian: one Riccati sweep without rows) and g0 (one adjoint sweep)
//   is added at stage 0.
#pragma once

#include "plan.hpp"
#include "stage_plan.hpp"
#include "wave_prims.hpp"

namespace copra_hip {

// Convergence test of both interior-point kernels (this one and lmpc_riccati_mfma.hpp), after a Newton step of inf-norm `step` (the FULL
// step, before it is cut to keep s and lam positive; `z` = inf-norm of the iterate): residuals of the rows <= 1e-9, complementarity
// measure mu <= mu_tol, step <= step_tol (1 + z).
// History, all of it found by the random differential tests (tests/random_controllers.py): rounds 2-3 also left on mu <= 1e-15 ALONE -- a
// controller without a single active row was accepted 5e-3 from the optimum while its steps were still 1e-3; a version of round 4 left on
// mu <= 1e-15 with a superlinear contraction of the last two steps -- 3e-4 (entry-wise) off on a controller whose contraction was
// superlinear but not quadratic.  One door now -- and ric_tail_ok for the case that the NEXT factorisation breaks down.
COPRA_DEV bool ric_converged(const StagePlan& S, double res, double mu, double step, double prev, double z)
{
    (void)prev;
    return res <= 1e-9 && mu <= S.mu_tol && step <= S.step_tol * (1.0 + z);
}
// At mu = 1e-18 the weights lam / s of the next factorisation are 1e16 and it may break down (on the device, not in the emulator: config 5
// lost 8 of 16 384 instances to the Goldfarb-Idnani kernel that way, 100 ms for them).  If that happens right after an iterate whose barrier
// was gone (mu <= 1e-15) and whose last two steps contracted superlinearly (r = step / prev <= 0.05, step r <= 1e-6 (1 + z)), that
// iterate is taken -- the one case in which the step test above cannot be waited for.
// Crossover (both kernels): once mu <= kRicSwitchMu the rows with lam > s become regularised equality rows, the others are switched off,
// and the iteration finishes as Newton's method on that equality-constrained QP.  Its solution is the optimum iff every held row pushes
// (multiplier >= 0) and every row left out is satisfied: a held row that pulls is released, a row left out that is violated is taken, and
// the iteration goes on on the corrected set -- at most kRicRefinements rounds, then the Goldfarb-Idnani kernel.  The factorisations never
// see a weight above 1e11 (the barrier's reach 1e16 at mu = 1e-18: a state row with such a weight takes the curvature of every direction
// it touches with it in the Riccati recursion, and the iteration stalls off the optimum or breaks down).
// (measured on config 5, 16 384 instances: crossover at 1e-10: 13.97 Newton steps, none given up; at 1e-8: 14.37 and 22 given up -- more
//  wrong guesses to correct; at 1e-6: 15.11 and 271)
constexpr double kRicSwitchMu = 1e-10;
constexpr double kRicEarlySwitchMu = 1e-5; // ... or earlier when the factorisation under the barrier breaks down (both kernels)
constexpr double kRicWasActive = -1.0, kRicWasIdle = -2.0; // markers in the slack array of rows that the crossover converted
constexpr int kRicRefinements = 6; // rounds of "release what pulls, take what is violated" after the crossover before the instance is given up
COPRA_DEV bool ric_tail_ok(double res, double mu, double step, double prev, double z)
{
    return res <= 1e-9 && mu <= 1e-15 && step <= 0.05 * prev && step * (step / prev) <= 1e-6 * (1.0 + z);
}

struct RicLds {
    double *AB, *Pm, *T, *M, *pv, *h, *g, *zk, *dzk, *dxn, *dv, *Kl, *Mi, *rowD, *rowC;
    // tables of the current stage class: W and the three sparse views of its rows (stage_plan.hpp)
    double *Wc, *rval, *gval, *eval;
    int *rptr, *rcol, *gptr, *grow, *eptr, *erow;
};

COPRA_DEV RicLds carve_riccati(double* lds, const StagePlan& S, int nx, int nu)
{
    const int nz = nx + nu;
    auto a2 = [](int v) { return (v + 1) & ~1; };
    RicLds L;
    double* p = lds;
    L.AB = p, p += a2(nx * nz);
    L.Pm = p, p += a2(nx * nx);
    L.T = p, p += a2(nx * nz > 2 * nu * nu ? nx * nz : 2 * nu * nu); // (also the Gauss-Jordan scratch, nu x 2 nu)
    L.M = p, p += a2(nz * nz);
    L.pv = p, p += a2(nz);
    L.h = p, p += a2(nz);
    L.g = p, p += a2(nz);
    L.zk = p, p += a2(nz);
    L.dzk = p, p += a2(nz);
    L.dxn = p, p += a2(nz);
    L.dv = p, p += a2(nz);
    L.Kl = p, p += a2(nu * nx);
    L.Mi = p, p += a2(nu * nu);
    const int mr = S.max_stage_rows > 0 ? S.max_stage_rows : 1;
    L.rowD = p, p += a2(mr);
    L.rowC = p, p += a2(mr);
    const int nn = S.max_nnz > 0 ? S.max_nnz : 1, ne = S.max_nnze > 0 ? S.max_nnze : 1;
    L.Wc = p, p += a2(nz * nz);
    L.rval = p, p += a2(nn);
    L.gval = p, p += a2(nn);
    L.eval = p, p += a2(ne);
    int* q = reinterpret_cast<int*>(p);
    L.rptr = q, q += mr + 1;
    L.rcol = q, q += nn;
    L.gptr = q, q += nz + 1;
    L.grow = q, q += nn;
    L.eptr = q, q += nz * nz + 1;
    L.erow = q, q += ne;
    return L;
}

// flags of a row of ONE instance
enum { kRowIneq = 0, kRowEq = 1, kRowOff = 2 };

// <NXT, NUT> = compile-time (xDim, uDim): index arithmetic folds and the small loops unroll; <0, 0> is the run-time-shape
// instantiation (every `e / nz` is then an integer division, ~40 instructions: several times slower on the same problem)
template <int NXT, int NUT>
COPRA_DEV void lmpc_riccati_body(const FusedPlan& P, const StagePlan& S)
{
    const int lane = lane_id();
    const int nx = NXT ? NXT : S.nx, nu = NUT ? NUT : S.nu, nz = nx + nu, N = S.N, m = S.m;
    const int NZ = (N + 1) * nz;
    double* lds = lds_base();
    const RicLds L = carve_riccati(lds, S, nx, nu);
    double* ws = S.ws + (size_t)instance_id() * (size_t)S.ws_total;
    double *Z = ws + S.oZ, *DZ = ws + S.oDZ, *Q = ws + S.oQ, *GB = ws + S.oGB;
    double *F = ws + S.oF, *Sv = ws + S.oS, *Lam = ws + S.oLam, *DS = ws + S.oDS, *DL = ws + S.oDL, *RP = ws + S.oRP;
    double* Flag = ws + S.oFlag;
    double *Kg = ws + S.oK, *Mig = ws + S.oMi, *Kvg = ws + S.oKv, *H0 = ws + S.oH0, *G0 = ws + S.oG0;
    const double* blob = S.blob;
    const double delta = S.delta;
    const double BIGF = 1e299;

    for (int witem = instance_id();; witem += instance_stride()) {
        // the batch is a queue: a wave that finishes early (fewer Newton steps) takes the next instance (S.next_instance
        // counts them; without it each wave walks its own arithmetic progression)
        int inst = witem;
        if (S.next_instance) {
            int v = 0;
            if (lane == 0) v = atomic_append(S.next_instance);
            inst = bcast_i32(v, 0);
        }
        if (inst >= P.batch) break;
        // optional phase profile (copra_batch_phase_profile): set-up | sweep 1 rows | sweep 1 gradient | sweep 1 factor |
        // forward sweeps | sweep 3 | update + results | total, shader-clock cycles of this instance
        long long prof[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        long long tprev = P.prof ? cycle_counter() : 0;
        const long long tstart = tprev;
        auto stamp = [&](int slot) {
            if (P.prof) {
                const long long t = cycle_counter();
                prof[slot] += t - tprev;
                tprev = t;
            }
        };
        // ------------------------------------------------------------------ 0. this instance's data
        const bool x0_free = S.x0_free && P.x0lb && P.x0ub; // (bounds never set: x0lb = x0ub = ps->x0, i.e. x0 is fixed)
        for (int e = lane; e < nx * nx; e += kWave) L.AB[e] = P.A[(size_t)inst * nx * nx + e];
        for (int e = lane; e < nx * nu; e += kWave) L.AB[nx * nx + e] = P.B[(size_t)inst * nx * nu + e];
        double* dvec = L.dv;
        // right-hand sides and row states
        for (int k = 0; k <= N; ++k) {
            const int c = S.cls_of_stage[k], r0 = S.cls_row0[c], nr = S.cls_row0[c + 1] - r0, gi0 = S.stage_row0[k];
            for (int r = lane; r < nr; r += kWave) {
                const int t = r0 + r, idx = S.r_sidx[t] + S.r_sstride[t] * k;
                double f;
                switch (S.r_src[t]) {
                case kSrcRowF: f = P.row_f_inst ? P.row_f_inst[(size_t)inst * P.mgen + idx] : P.row_f[idx]; break;
                case kSrcUb: f = P.ub_inst ? P.ub_inst[(size_t)inst * P.n + idx] : P.ub[idx]; break;
                case kSrcNegLb: f = -(P.lb_inst ? P.lb_inst[(size_t)inst * P.n + idx] : P.lb[idx]); break;
                case kSrcX0Ub: f = x0_free ? P.x0ub[(size_t)inst * nx + idx] : BIGF; break;
                default: f = x0_free ? -P.x0lb[(size_t)inst * nx + idx] : BIGF; break;
                }
                F[gi0 + r] = f;
                Flag[gi0 + r] = (f >= BIGF) ? (double)kRowOff : (S.r_eq[t] ? (double)kRowEq : (double)kRowIneq);
            }
        }
        wave_sync_full(); // (F / Flag are read across lanes below)
        // a bound pair  lb == ub  (up to rounding) is one equality row: the upper row becomes it, the lower row is off
        for (int k = 0; k <= N; ++k) {
            const int c = S.cls_of_stage[k], r0 = S.cls_row0[c], nr = S.cls_row0[c + 1] - r0, gi0 = S.stage_row0[k];
            for (int r = lane; r < nr; r += kWave) {
                const int t = r0 + r;
                if (r == 0 || S.r_kind[t] != 1 || S.r_kind[t - 1] != 1 || S.r_aoff[t] != S.r_aoff[t - 1]) continue;
                if (!(S.r_sign[t - 1] == 1.0 && S.r_sign[t] == -1.0)) continue;
                if (Flag[gi0 + r] != (double)kRowIneq || Flag[gi0 + r - 1] != (double)kRowIneq) continue;
                const double up = F[gi0 + r - 1], lo = -F[gi0 + r];
                if (up - lo <= 1e-12 * fmax(1.0, fabs(up))) {
                    Flag[gi0 + r - 1] = (double)kRowEq;
                    Flag[gi0 + r] = (double)kRowOff;
                }
            }
        }
        // q_k = - sum_rows w p a  (costFunctions.cpp: the p-dependent part of the gradient)
        for (int e = lane; e < NZ; e += kWave) {
            const int k = e / nz, i = e - k * nz;
            const int c = S.cls_of_stage[k];
            double acc = 0.0;
            for (int t = S.cls_crow0[c]; t < S.cls_crow0[c + 1]; ++t) {
                const int ct = S.cr_cost[t];
                const double pv = P.cost_p[ct] ? P.cost_p[ct][(size_t)inst * P.cost[ct].prows + S.cr_pidx[t]]
                                               : P.params[P.cost[ct].offP + S.cr_pidx[t]];
                acc -= S.cr_w[t] * pv * blob[S.cr_aoff[t] + i];
            }
            Q[e] = acc;
        }
        for (int e = lane; e < nx; e += kWave) dvec[e] = P.d[(size_t)inst * nx + e];
        wave_sync_full();

        // ---- helpers -------------------------------------------------------------------------------------------
        // tables of stage class c -> LDS (W, the dense rows' coefficients, the unit rows' component and sign); stages of
        // one class are mostly consecutive, so this happens a few times per sweep
        int cur_cls = -1;
        auto load_class = [&](int c) {
            if (c == cur_cls) return;
            cur_cls = c;
            const int nr = S.cls_row0[c + 1] - S.cls_row0[c];
            const int* ib = S.iblob;
            wave_sync(); // (readers of the previous class's tables are done)
            for (int e = lane; e < nz * nz; e += kWave) L.Wc[e] = blob[S.cls_W[c] + e];
            for (int e = lane; e <= nr; e += kWave) L.rptr[e] = ib[S.cls_rptr[c] + e];
            for (int e = lane; e <= nz; e += kWave) L.gptr[e] = ib[S.cls_gptr[c] + e];
            for (int e = lane; e <= nz * nz; e += kWave) L.eptr[e] = ib[S.cls_eptr[c] + e];
            const int nnz = ib[S.cls_rptr[c] + nr], nnze = ib[S.cls_eptr[c] + nz * nz];
            for (int e = lane; e < nnz; e += kWave) {
                L.rcol[e] = ib[S.cls_rcol[c] + e];
                L.rval[e] = blob[S.cls_rval[c] + e];
                L.grow[e] = ib[S.cls_grow[c] + e];
                L.gval[e] = blob[S.cls_gval[c] + e];
            }
            for (int e = lane; e < nnze; e += kWave) {
                L.erow[e] = ib[S.cls_erow[c] + e];
                L.eval[e] = blob[S.cls_eval[c] + e];
            }
            wave_sync();
        };
        // a_r' v for row r of the CURRENT class (v: nz values in LDS); bound rows have one term, mixed rows two
        auto row_dot = [&](int r, int, const double* v) -> double {
            double acc = 0.0;
            for (int q = L.rptr[r]; q < L.rptr[r + 1]; ++q) acc += L.rval[q] * v[L.rcol[q]];
            return acc;
        };
        // x_{k+1} = A x_k + B u_k + d along Z (the controls as stored in Z), from the x_0 stored in Z[0..nx)
        auto rollout = [&]() {
            wave_sync_full();
            for (int i = lane; i < nx; i += kWave) L.zk[i] = Z[i];
            for (int k = 0; k < N; ++k) {
                for (int i = nx + lane; i < nz; i += kWave) L.zk[i] = Z[k * nz + i];
                wave_sync();
                for (int i = lane; i < nx; i += kWave) {
                    double acc = dvec[i];
                    for (int j = 0; j < nz; ++j) acc += L.AB[i + nx * j] * L.zk[j];
                    L.dxn[i] = acc;
                    Z[(k + 1) * nz + i] = acc;
                }
                wave_sync();
                for (int i = lane; i < nx; i += kWave) L.zk[i] = L.dxn[i];
            }
            wave_sync_full();
        };
        // One stage of the backward factorisation.  Before: zk holds z_k (u-part 0 at k = N), rowD / rowC the weights and
        // gradient coefficients of the stage's rows, Pm / pv the cost-to-go of stage k + 1.  After: Pm / pv of stage k;
        // K, Muu^-1, kv stored for the forward sweeps.  Returns false when Muu is not positive definite.
        auto stage_gradient = [&](int k, bool with_rows, bool store_gb) {
            const double* Wk = L.Wc;
            for (int i = lane; i < nz; i += kWave) {
                double gb;
                if (store_gb) {
                    gb = Q[k * nz + i];
                    for (int j = 0; j < nz; ++j) gb += Wk[i + nz * j] * L.zk[j];
                    if (k == 0 && x0_free && i < nx) { // InitialStateLMPC: + (R - P0) x0 + (r - g0)
                        gb += G0[i];
                        for (int j = 0; j < nx; ++j) gb += H0[i + nx * j] * L.zk[j];
                    }
                    GB[k * nz + i] = gb;
                } else {
                    gb = GB[k * nz + i];
                }
                if (with_rows)
                    for (int q = L.gptr[i]; q < L.gptr[i + 1]; ++q) gb += L.gval[q] * L.rowC[L.grow[q]];
                L.g[i] = gb;
            }
        };
        auto stage_factor = [&](int k, bool with_rows) -> bool {
            const double* Wk = L.Wc;
            // H = W + sum D a a'
            for (int e = lane; e < nz * nz; e += kWave) { // entry e = i + nz j
                double acc = Wk[e];
                if (with_rows)
                    for (int q = L.eptr[e]; q < L.eptr[e + 1]; ++q) acc += L.eval[q] * L.rowD[L.erow[q]];
                L.M[e] = acc;
            }
            if (k == N) { // P_N = Hxx, p_N = g_x
                wave_sync();
                for (int e = lane; e < nx * nx; e += kWave) {
                    const int j = e / nx, i = e - j * nx;
                    L.Pm[e] = L.M[i + nz * j];
                }
                for (int i = lane; i < nx; i += kWave) L.pv[i] = L.g[i];
                wave_sync();
                return true;
            }
            // T = P [A B]
            for (int e = lane; e < nx * nz; e += kWave) {
                const int j = e / nx, i = e - j * nx;
                double acc = 0.0;
                for (int l = 0; l < nx; ++l) acc += L.Pm[i + nx * l] * L.AB[l + nx * j];
                L.T[e] = acc;
            }
            wave_sync();
            // M = H + [A B]' T ;  h = g + [A B]' p
            for (int e = lane; e < nz * nz; e += kWave) {
                const int b = e / nz, a = e - b * nz;
                double acc = L.M[e];
                for (int l = 0; l < nx; ++l) acc += L.AB[l + nx * a] * L.T[l + nx * b];
                L.M[e] = acc;
            }
            for (int a = lane; a < nz; a += kWave) {
                double acc = L.g[a];
                for (int l = 0; l < nx; ++l) acc += L.AB[l + nx * a] * L.pv[l];
                L.h[a] = acc;
            }
            wave_sync();
            bool ok = true;
            if constexpr (NUT > 0) {
                // Muu^-1 from a Cholesky factor held in registers: every lane factorises the nu x nu block redundantly
                // (NUT (NUT + 1) / 2 broadcast LDS reads, no round trips through LDS in between), then lane c solves for
                // column c of the inverse
                double Lc[NUT][NUT];
#pragma unroll
                for (int i = 0; i < NUT; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) Lc[i][j] = L.M[(nx + i) + nz * (nx + j)];
#pragma unroll
                for (int j = 0; j < NUT; ++j) {
                    double dg = Lc[j][j];
#pragma unroll
                    for (int t = 0; t < j; ++t) dg -= Lc[j][t] * Lc[j][t];
                    if (!(dg > 0.0)) ok = false;
                    const double inv = 1.0 / sqrt(dg);
                    Lc[j][j] = inv; // (the diagonal holds 1 / L_jj)
#pragma unroll
                    for (int i = j + 1; i < NUT; ++i) {
                        double v = Lc[i][j];
#pragma unroll
                        for (int t = 0; t < j; ++t) v -= Lc[i][t] * Lc[j][t];
                        Lc[i][j] = v * inv;
                    }
                }
                const int cc = lane % NUT;
                double y[NUT];
#pragma unroll
                for (int i = 0; i < NUT; ++i) { // L y = e_cc
                    double v = (i == cc) ? 1.0 : 0.0;
#pragma unroll
                    for (int t = 0; t < i; ++t) v -= Lc[i][t] * y[t];
                    y[i] = v * Lc[i][i];
                }
#pragma unroll
                for (int i = NUT - 1; i >= 0; --i) { // L' x = y
                    double v = y[i];
#pragma unroll
                    for (int t = i + 1; t < NUT; ++t) v -= Lc[t][i] * y[t];
                    y[i] = v * Lc[i][i];
                }
                if (lane < NUT) {
#pragma unroll
                    for (int i = 0; i < NUT; ++i) {
                        L.Mi[i + NUT * lane] = y[i];
                        Mig[(size_t)k * nu * nu + i + NUT * lane] = y[i];
                    }
                }
                wave_sync();
            } else {
            // Muu^-1 by Gauss-Jordan on [Muu | I] (Muu is symmetric positive definite: no pivoting); Kl is the scratch
            // (nu x 2 nu <= nu x nx is not guaranteed: use T, which is free again)
            double* GJ = L.T; // nu x 2nu, row-major with leading dimension 2 nu
            const int w2 = 2 * nu;
            for (int e = lane; e < nu * w2; e += kWave) {
                const int r = e / w2, cc = e - r * w2;
                GJ[e] = (cc < nu) ? L.M[(nx + r) + nz * (nx + cc)] : ((cc - nu == r) ? 1.0 : 0.0);
            }
            wave_sync();
            for (int p = 0; p < nu; ++p) {
                const double piv = GJ[p * w2 + p];
                if (!(piv > 0.0)) ok = false;
                const double ip = 1.0 / piv;
                wave_sync();
                // eliminate column p from every other row; scale row p
                for (int e = lane; e < nu * w2; e += kWave) {
                    const int r = e / w2, cc = e - r * w2;
                    if (r == p) continue;
                    const double fct = GJ[r * w2 + p] * ip;
                    if (cc != p) GJ[e] -= fct * GJ[p * w2 + cc];
                }
                wave_sync();
                for (int e = lane; e < nu * w2; e += kWave) {
                    const int r = e / w2, cc = e - r * w2;
                    if (r == p)
                        GJ[e] *= ip;
                    else if (cc == p)
                        GJ[e] = 0.0;
                }
                wave_sync();
            }
            for (int e = lane; e < nu * nu; e += kWave) {
                const int cc = e / nu, r = e - cc * nu;
                const double v = 0.5 * (GJ[r * w2 + nu + cc] + GJ[cc * w2 + nu + r]);
                L.Mi[e] = v;
                Mig[(size_t)k * nu * nu + e] = v;
            }
            wave_sync();
            }
            // K = -Muu^-1 Mux (nu x nx), kv = -Muu^-1 hu
            for (int e = lane; e < nu * nx; e += kWave) {
                const int j = e / nu, i = e - j * nu;
                double acc = 0.0;
                for (int l = 0; l < nu; ++l) acc += L.Mi[i + nu * l] * L.M[(nx + l) + nz * j];
                L.Kl[e] = -acc;
                Kg[(size_t)k * nu * nx + e] = -acc;
            }
            for (int i = lane; i < nu; i += kWave) {
                double acc = 0.0;
                for (int l = 0; l < nu; ++l) acc += L.Mi[i + nu * l] * L.h[nx + l];
                L.dxn[i] = -acc; // kv
                Kvg[(size_t)k * nu + i] = -acc;
            }
            wave_sync();
            // P = Mxx + Mux' K (symmetrised), p = hx + Mux' kv
            for (int e = lane; e < nx * nx; e += kWave) {
                const int j = e / nx, i = e - j * nx;
                double a1 = L.M[i + nz * j], a2 = a1;
                for (int l = 0; l < nu; ++l) {
                    a1 += L.M[(nx + l) + nz * i] * L.Kl[l + nu * j];
                    a2 += L.M[(nx + l) + nz * j] * L.Kl[l + nu * i];
                }
                L.Pm[e] = 0.5 * (a1 + a2);
            }
            for (int i = lane; i < nx; i += kWave) {
                double acc = L.h[i];
                for (int l = 0; l < nu; ++l) acc += L.M[(nx + l) + nz * i] * L.dxn[l];
                L.pv[i] = acc;
            }
            wave_sync();
            return ok;
        };
        // dx_0 = -(P_0 + R - P0)^-1 p_0  (InitialStateLMPC) into dzk[0..nx); Gauss-Jordan on [P | -p] in M (nx x (nx+1))
        auto solve_x0 = [&]() -> bool {
            double* GJ = L.M;
            const int w1 = nx + 1;
            for (int e = lane; e < nx * w1; e += kWave) {
                const int r = e / w1, cc = e - r * w1;
                GJ[e] = (cc < nx) ? L.Pm[r + nx * cc] + H0[r + nx * cc] : -L.pv[r];
            }
            wave_sync();
            bool ok = true;
            for (int p = 0; p < nx; ++p) {
                const double piv = GJ[p * w1 + p];
                if (!(piv > 0.0)) ok = false;
                const double ip = 1.0 / piv;
                wave_sync();
                for (int e = lane; e < nx * w1; e += kWave) {
                    const int r = e / w1, cc = e - r * w1;
                    if (r == p || cc == p) continue;
                    GJ[e] -= GJ[r * w1 + p] * ip * GJ[p * w1 + cc];
                }
                wave_sync();
                for (int e = lane; e < nx * w1; e += kWave) {
                    const int r = e / w1, cc = e - r * w1;
                    if (r == p)
                        GJ[e] *= ip;
                    else if (cc == p)
                        GJ[e] = 0.0;
                }
                wave_sync();
            }
            for (int i = lane; i < nx; i += kWave) L.dzk[i] = GJ[i * w1 + nx];
            wave_sync();
            return ok;
        };

        // ------------------------------------------------------------------ 1. starting point
        for (int e = lane; e < NZ; e += kWave) Z[e] = 0.0;
        bool good = true;
        if (x0_free) {
            // P0 (unconstrained cost-to-go Hessian) and g0 = dJ/dx0 at (x0, U) = 0
            for (int e = lane; e < nx * nx; e += kWave) H0[e] = 0.0;
            for (int e = lane; e < nx; e += kWave) G0[e] = 0.0;
            rollout(); // x0 = 0, U = 0
            for (int k = N; k >= 0; --k) {
                load_class(S.cls_of_stage[k]);
                for (int e = lane; e < nz; e += kWave) L.zk[e] = Z[k * nz + e];
                wave_sync();
                stage_gradient(k, false, true); // g = W z + q  (H0 = G0 = 0 for now), stored in GB
                wave_sync();
                good = stage_factor(k, false) && good;
            }
            // adjoint sweep for g0: lam_N = g_N,x ; lam_k = g_k,x + A' lam_{k+1}
            for (int i = lane; i < nx; i += kWave) L.h[i] = GB[N * nz + i];
            wave_sync();
            for (int k = N - 1; k >= 0; --k) {
                for (int i = lane; i < nx; i += kWave) {
                    double acc = GB[k * nz + i];
                    for (int l = 0; l < nx; ++l) acc += L.AB[l + nx * i] * L.h[l];
                    L.g[i] = acc;
                }
                wave_sync();
                for (int i = lane; i < nx; i += kWave) L.h[i] = L.g[i];
                wave_sync();
            }
            for (int e = lane; e < nx * nx; e += kWave) H0[e] = P.is_R[e] - L.Pm[e];
            for (int i = lane; i < nx; i += kWave) G0[i] = P.is_r[i] - L.h[i];
        }
        wave_sync_full();
        for (int i = lane; i < nx; i += kWave) {
            double v = P.x0[(size_t)inst * nx + i];
            if (x0_free) v = fmin(fmax(v, P.x0lb[(size_t)inst * nx + i]), P.x0ub[(size_t)inst * nx + i]);
            Z[i] = v;
        }
        rollout();
        int n_ineq = 0;
        for (int k = 0; k <= N; ++k) {
            const int c = S.cls_of_stage[k], nr = S.cls_row0[c + 1] - S.cls_row0[c], nd = S.cls_ndense[c], gi0 = S.stage_row0[k];
            load_class(c);
            for (int e = lane; e < nz; e += kWave) L.zk[e] = Z[k * nz + e];
            wave_sync();
            for (int r = lane; r < nr; r += kWave) {
                const int gi = gi0 + r;
                const int fl = (int)Flag[gi];
                double sv = 1.0, lv = 0.0;
                if (fl == kRowIneq) { // (centred start: stage_plan.hpp, s_floor / lam0; the multipliers follow below)
                    sv = fmax(F[gi] - row_dot(r, nd, L.zk), S.s_floor);
                    lv = S.lam0 > 0.0 ? S.lam0 : 1.0;
                    n_ineq += 1;
                }
                Sv[gi] = sv;
                Lam[gi] = lv;
            }
            wave_sync();
        }
        n_ineq = (int)(wave_sum((double)n_ineq) + 0.5);
        wave_sync_full();
        const double inv_mi = n_ineq > 0 ? 1.0 / (double)n_ineq : 0.0;
        if (!(S.lam0 > 0.0)) { // every complementarity product s lam starts at |lam0| x the mean slack (lmpc_riccati_mfma.hpp)
            double ssum = 0.0;
            for (int gi = lane; gi < m; gi += kWave)
                if ((int)Flag[gi] == kRowIneq) ssum += Sv[gi];
            const double mu0 = -S.lam0 * wave_sum(ssum) * inv_mi;
            for (int gi = lane; gi < m; gi += kWave)
                if ((int)Flag[gi] == kRowIneq) Lam[gi] = mu0 / Sv[gi];
            wave_sync_full();
        }

        // ---- crossover: the barrier has told which rows are active (lam > s).  From there on those are EQUALITY rows of the regularised
        // kind (weight 1 / delta, an explicit multiplier), the others are off: Newton's method on an equality-constrained QP, whose
        // factorisations carry weights of 1e6 instead of the barrier's 1e16 -- see ric_converged.
        auto cross_over = [&]() {
            for (int gi = lane; gi < m; gi += kWave)
                if ((int)Flag[gi] == kRowIneq) {
                    if (Lam[gi] > Sv[gi]) {
                        Flag[gi] = (double)kRowEq;
                        Sv[gi] = kRicWasActive; // (marker: an inequality row held as an equality -- its multiplier must come out >= 0)
                    } else {
                        Flag[gi] = (double)kRowOff;
                        Sv[gi] = kRicWasIdle; // (marker: an inequality row left out -- it must come out satisfied)
                        Lam[gi] = 0.0;
                    }
                }
            wave_sync_full();
        };
        stamp(0);
        // ------------------------------------------------------------------ 2. Newton iterations
        int it = 0;
        double prev_step = 1.0e300; // the step before
        bool tail_ok = false; // ric_tail_ok of the iterate the loop stands on
        bool polishing = false; // after the crossover (below): active rows as equalities, the others off
        int polish_it = 0, refinements = 0;
        bool converged = false;
        for (it = 1; it <= S.max_iter && good; ++it) {
            // ---- sweep 1 (backward): residuals, barrier weights, factorisation, predictor right-hand side
            double musum = 0.0, maxr = 0.0;
            bool done1 = false;
            if constexpr (NXT > 0 && NUT > 0) {
                if (S.max_stage_rows <= kWave) { // operands of stage k - 1 requested before stage k is computed (see sweep 3)
                    struct Regs1 {
                        int fl;
                        double f, sv, lv, z, q;
                    };
                    auto issue1 = [&](int k, Regs1& R) {
                        const int c = S.cls_of_stage[k], nr = S.cls_row0[c + 1] - S.cls_row0[c], gi = S.stage_row0[k] + lane;
                        R.fl = kRowOff;
                        R.f = R.sv = R.lv = R.z = R.q = 0.0;
                        if (lane < nr) {
                            R.fl = (int)Flag[gi];
                            R.f = F[gi], R.sv = Sv[gi], R.lv = Lam[gi];
                        }
                        if (lane < nz) R.z = Z[k * nz + lane], R.q = Q[k * nz + lane];
                    };
                    Regs1 cur, nxt;
                    issue1(N, cur);
                    for (int k = N; k >= 0; --k) {
                        if (k > 0) issue1(k - 1, nxt);
                        load_class(S.cls_of_stage[k]);
                        if (lane < nz) L.zk[lane] = cur.z;
                        wave_sync();
                        {
                            double Dv = 0.0, Cv = 0.0;
                            if (cur.fl != kRowOff) {
                                const int gi = S.stage_row0[k] + lane;
                                const double az = row_dot(lane, 0, L.zk);
                                if (cur.fl == kRowEq) {
                                    const double re = az - cur.f;
                                    Dv = 1.0 / delta;
                                    Cv = cur.lv + re / delta;
                                    RP[gi] = re;
                                } else {
                                    const double rp = az + cur.sv - cur.f;
                                    Dv = cur.lv / cur.sv;
                                    Cv = Dv * rp;
                                    RP[gi] = rp;
                                    maxr = fmax(maxr, fabs(rp));
                                    musum += cur.sv * cur.lv;
                                }
                            }
                            if (lane < S.max_stage_rows) {
                                L.rowD[lane] = Dv;
                                L.rowC[lane] = Cv;
                            }
                        }
                        wave_sync();
                        stamp(1);
                        if (lane < nz) { // g = W z + q (+ the InitialStateLMPC terms at stage 0) + A' c
                            double gb = cur.q;
#pragma unroll
                            for (int j = 0; j < NXT + NUT; ++j) gb += L.Wc[lane + nz * j] * L.zk[j];
                            if (k == 0 && x0_free && lane < nx) {
                                gb += G0[lane];
                                for (int j = 0; j < nx; ++j) gb += H0[lane + nx * j] * L.zk[j];
                            }
                            GB[k * nz + lane] = gb;
                            for (int q = L.gptr[lane]; q < L.gptr[lane + 1]; ++q) gb += L.gval[q] * L.rowC[L.grow[q]];
                            L.g[lane] = gb;
                        }
                        wave_sync();
                        stamp(2);
                        good = stage_factor(k, true) && good;
                        stamp(3);
                        cur = nxt;
                    }
                    done1 = true;
                }
            }
            if (!done1)
            for (int k = N; k >= 0; --k) {
                const int c = S.cls_of_stage[k], nr = S.cls_row0[c + 1] - S.cls_row0[c], nd = S.cls_ndense[c], gi0 = S.stage_row0[k];
                load_class(c);
                for (int e = lane; e < nz; e += kWave) L.zk[e] = Z[k * nz + e];
                wave_sync();
                for (int r = lane; r < nr; r += kWave) {
                    const int gi = gi0 + r;
                    const int fl = (int)Flag[gi];
                    double Dv = 0.0, Cv = 0.0;
                    if (fl != kRowOff) {
                        const double az = row_dot(r, nd, L.zk);
                        if (fl == kRowEq) {
                            const double re = az - F[gi];
                            Dv = 1.0 / delta;
                            Cv = Lam[gi] + re / delta;
                            RP[gi] = re;
                        } else {
                            const double sv = Sv[gi], lv = Lam[gi];
                            const double rp = az + sv - F[gi];
                            Dv = lv / sv;
                            Cv = Dv * rp;
                            RP[gi] = rp;
                            maxr = fmax(maxr, fabs(rp));
                            musum += sv * lv;
                        }
                    }
                    L.rowD[r] = Dv;
                    L.rowC[r] = Cv;
                }
                wave_sync();
                stamp(1);
                stage_gradient(k, true, true);
                wave_sync();
                stamp(2);
                good = stage_factor(k, true) && good;
                stamp(3);
            }
            const double mu = wave_sum(musum) * inv_mi;
            const double maxres = wave_max(maxr);
            if (!good) { // (the factorisation broke down; the iterate itself is untouched)
                // ... under barrier weights that are large already (mu small, rows feasible): cross over HERE instead of giving the
                // instance up -- wrong guesses about weakly active rows are corrected afterwards like any others
                if (!polishing && n_ineq > 0 && maxres <= 1e-9 && mu <= kRicEarlySwitchMu && it > 1) {
                    cross_over();
                    polishing = true;
                    good = true;
                    prev_step = 1.0e300;
                    tail_ok = false;
                    continue;
                }
                converged = tail_ok;
                break;
            }
            wave_sync_full(); // (K, Muu^-1, kv of this sweep are read across lanes from here on)
            // ---- two forward sweeps (predictor, corrector) with one backward vector sweep in between
            double alpha = 1.0, sigma_mu = 0.0, step_inf = 0.0, z_inf = 0.0;
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) { // ---- sweep 3 (backward): corrector right-hand side through the stored factors
                    bool done3 = false;
                    if constexpr (NXT > 0 && NUT > 0) {
                        if (S.max_stage_rows <= kWave && NXT * NUT <= 2 * kWave) {
                            // compile-time shape, at most one row per lane: every operand of stage k - 1 is requested from
                            // the workspace BEFORE stage k is computed, so its HBM / L2 round trip runs underneath
                            // (K and Muu^-1 travel ONE or TWO elements per lane -- element e of the nu x nx block in lane
                            //  e % 64 -- and are put into LDS when their stage starts: two registers instead of nu + nu)
                            struct Regs3 {
                                int fl;
                                double sv, lv, rp, ds, dl, gb, mi, k0, k1;
                            };
                            auto issue3 = [&](int k, Regs3& R) {
                                const int c = S.cls_of_stage[k], nr = S.cls_row0[c + 1] - S.cls_row0[c], gi = S.stage_row0[k] + lane;
                                R.fl = kRowOff;
                                if (lane < nr) {
                                    R.fl = (int)Flag[gi];
                                    R.sv = Sv[gi], R.lv = Lam[gi], R.rp = RP[gi], R.ds = DS[gi], R.dl = DL[gi];
                                }
                                if (lane < nz) R.gb = GB[k * nz + lane];
                                R.mi = R.k0 = R.k1 = 0.0;
                                if (k < N) {
                                    if (lane < nu * nu) R.mi = Mig[(size_t)k * nu * nu + lane];
                                    if (lane < nu * nx) R.k0 = Kg[(size_t)k * nu * nx + lane];
                                    if (lane + kWave < nu * nx) R.k1 = Kg[(size_t)k * nu * nx + lane + kWave];
                                }
                            };
                            Regs3 cur, nxt;
                            issue3(N, cur);
                            for (int k = N; k >= 0; --k) {
                                if (k > 0) issue3(k - 1, nxt);
                                load_class(S.cls_of_stage[k]);
                                if (k < N) {
                                    if (lane < nu * nu) L.Mi[lane] = cur.mi;
                                    if (lane < nu * nx) L.Kl[lane] = cur.k0;
                                    if (lane + kWave < nu * nx) L.Kl[lane + kWave] = cur.k1;
                                }
                                {
                                    double Cv = 0.0;
                                    if (cur.fl == kRowEq)
                                        Cv = cur.lv + cur.rp / delta;
                                    else if (cur.fl == kRowIneq)
                                        Cv = (sigma_mu - cur.ds * cur.dl) / cur.sv + (cur.lv / cur.sv) * cur.rp;
                                    if (lane < S.max_stage_rows) L.rowC[lane] = Cv;
                                }
                                wave_sync();
                                if (lane < nz) {
                                    double gb = cur.gb;
                                    for (int q = L.gptr[lane]; q < L.gptr[lane + 1]; ++q) gb += L.gval[q] * L.rowC[L.grow[q]];
                                    L.g[lane] = gb;
                                }
                                wave_sync();
                                if (k == N) {
                                    if (lane < nx) L.pv[lane] = L.g[lane];
                                } else {
                                    if (lane < nz) { // h = g + [A B]' p
                                        double acc = L.g[lane];
#pragma unroll
                                        for (int l = 0; l < NXT; ++l) acc += L.AB[l + nx * lane] * L.pv[l];
                                        L.h[lane] = acc;
                                    }
                                    wave_sync();
                                    double kvv = 0.0, pn = 0.0;
                                    if (lane < nu) { // kv = -Muu^-1 hu
#pragma unroll
                                        for (int l = 0; l < NUT; ++l) kvv += L.Mi[lane + nu * l] * L.h[nx + l];
                                        Kvg[(size_t)k * nu + lane] = -kvv;
                                    }
                                    if (lane < nx) { // p = hx + K' hu
                                        pn = L.h[lane];
#pragma unroll
                                        for (int l = 0; l < NUT; ++l) pn += L.Kl[l + nu * lane] * L.h[nx + l];
                                    }
                                    wave_sync();
                                    if (lane < nx) L.pv[lane] = pn;
                                }
                                wave_sync();
                                cur = nxt;
                            }
                            done3 = true;
                        }
                    }
                    if (!done3)
                    for (int k = N; k >= 0; --k) {
                        const int c = S.cls_of_stage[k], nr = S.cls_row0[c + 1] - S.cls_row0[c], gi0 = S.stage_row0[k];
                        load_class(c);
                        for (int r = lane; r < nr; r += kWave) {
                            const int gi = gi0 + r;
                            const int fl = (int)Flag[gi];
                            double Cv = 0.0;
                            if (fl == kRowEq)
                                Cv = Lam[gi] + RP[gi] / delta;
                            else if (fl == kRowIneq) {
                                const double sv = Sv[gi];
                                Cv = (sigma_mu - DS[gi] * DL[gi]) / sv + (Lam[gi] / sv) * RP[gi];
                            }
                            L.rowC[r] = Cv;
                        }
                        wave_sync();
                        stage_gradient(k, true, false);
                        wave_sync();
                        if (k == N) {
                            for (int i = lane; i < nx; i += kWave) L.pv[i] = L.g[i];
                            wave_sync();
                            continue;
                        }
                        for (int a = lane; a < nz; a += kWave) { // h = g + [A B]' p
                            double acc = L.g[a];
                            for (int l = 0; l < nx; ++l) acc += L.AB[l + nx * a] * L.pv[l];
                            L.h[a] = acc;
                        }
                        wave_sync();
                        for (int i = lane; i < nu; i += kWave) { // kv = -Muu^-1 hu
                            double acc = 0.0;
                            for (int l = 0; l < nu; ++l) acc += Mig[(size_t)k * nu * nu + i + nu * l] * L.h[nx + l];
                            Kvg[(size_t)k * nu + i] = -acc;
                        }
                        for (int i = lane; i < nx; i += kWave) { // p = hx + K' hu
                            double acc = L.h[i];
                            for (int l = 0; l < nu; ++l) acc += Kg[(size_t)k * nu * nx + l + nu * i] * L.h[nx + l];
                            L.dxn[i] = acc;
                        }
                        wave_sync();
                        for (int i = lane; i < nx; i += kWave) L.pv[i] = L.dxn[i];
                        wave_sync();
                    }
                    wave_sync_full();
                    stamp(5);
                }
                // dx_0
                if (x0_free) {
                    good = solve_x0() && good;
                } else {
                    for (int i = lane; i < nx; i += kWave) L.dzk[i] = 0.0;
                    wave_sync();
                }
                double amin = 1.0e300;
                step_inf = 0.0;
                bool donef = false;
                if constexpr (NXT > 0 && NUT > 0) {
                    if (S.max_stage_rows <= kWave && NXT * NUT <= 2 * kWave) {
                        struct RegsF {
                            int fl;
                            double sv, lv, rp, ds, dl, kv, k0, k1;
                        };
                        auto issuef = [&](int k, RegsF& R) {
                            const int c = S.cls_of_stage[k], nr = S.cls_row0[c + 1] - S.cls_row0[c], gi = S.stage_row0[k] + lane;
                            R.fl = kRowOff;
                            if (lane < nr) {
                                R.fl = (int)Flag[gi];
                                R.sv = Sv[gi], R.lv = Lam[gi], R.rp = RP[gi], R.ds = DS[gi], R.dl = DL[gi];
                            }
                            R.kv = R.k0 = R.k1 = 0.0;
                            if (k < N) {
                                if (lane < nu) R.kv = Kvg[(size_t)k * nu + lane];
                                if (lane < nu * nx) R.k0 = Kg[(size_t)k * nu * nx + lane];
                                if (lane + kWave < nu * nx) R.k1 = Kg[(size_t)k * nu * nx + lane + kWave];
                            }
                        };
                        RegsF cur, nxt;
                        issuef(0, cur);
                        for (int k = 0; k <= N; ++k) {
                            if (k < N) issuef(k + 1, nxt);
                            load_class(S.cls_of_stage[k]);
                            if (k < N) {
                                if (lane < nu * nx) L.Kl[lane] = cur.k0;
                                if (lane + kWave < nu * nx) L.Kl[lane + kWave] = cur.k1;
                            }
                            wave_sync();
                            if (lane < nu) { // du = K dx + kv
                                double acc = 0.0;
                                if (k < N) {
                                    acc = cur.kv;
#pragma unroll
                                    for (int j = 0; j < NXT; ++j) acc += L.Kl[lane + nu * j] * L.dzk[j];
                                }
                                L.dzk[nx + lane] = acc;
                            }
                            wave_sync();
                            if (lane < nz) {
                                const double v = L.dzk[lane];
                                DZ[k * nz + lane] = v;
                                step_inf = fmax(step_inf, fabs(v));
                            }
                            if (cur.fl != kRowOff) {
                                const int gi = S.stage_row0[k] + lane;
                                const double adz = row_dot(lane, 0, L.dzk);
                                if (cur.fl == kRowEq) {
                                    DS[gi] = adz; // (kept for the multiplier update)
                                } else {
                                    const double ds = -cur.rp - adz;
                                    const double corr = (pass == 1) ? (sigma_mu - cur.ds * cur.dl) : 0.0;
                                    const double dl = (corr - cur.lv * cur.sv - cur.lv * ds) / cur.sv;
                                    DS[gi] = ds; // (pass 1 overwrites the predictor's direction, which this lane holds in cur)
                                    DL[gi] = dl;
                                    if (ds < 0.0) amin = fmin(amin, -cur.sv / ds);
                                    if (dl < 0.0) amin = fmin(amin, -cur.lv / dl);
                                }
                            }
                            if (k < N) {
                                double acc = 0.0;
                                if (lane < nx) { // dx+ = A dx + B du
#pragma unroll
                                    for (int j = 0; j < NXT + NUT; ++j) acc += L.AB[lane + nx * j] * L.dzk[j];
                                }
                                wave_sync();
                                if (lane < nx) L.dzk[lane] = acc;
                            }
                            wave_sync();
                            cur = nxt;
                        }
                        donef = true;
                    }
                }
                if (!donef)
                for (int k = 0; k <= N; ++k) {
                    const int c = S.cls_of_stage[k], nr = S.cls_row0[c + 1] - S.cls_row0[c], nd = S.cls_ndense[c], gi0 = S.stage_row0[k];
                    load_class(c);
                    if (k < N) {
                        for (int i = lane; i < nu; i += kWave) { // du = K dx + kv
                            double acc = Kvg[(size_t)k * nu + i];
                            for (int j = 0; j < nx; ++j) acc += Kg[(size_t)k * nu * nx + i + nu * j] * L.dzk[j];
                            L.dzk[nx + i] = acc;
                        }
                    } else {
                        for (int i = lane; i < nu; i += kWave) L.dzk[nx + i] = 0.0;
                    }
                    wave_sync();
                    for (int e = lane; e < nz; e += kWave) {
                        DZ[k * nz + e] = L.dzk[e];
                        step_inf = fmax(step_inf, fabs(L.dzk[e]));
                    }
                    for (int r = lane; r < nr; r += kWave) {
                        const int gi = gi0 + r;
                        const int fl = (int)Flag[gi];
                        if (fl == kRowOff) continue;
                        const double adz = row_dot(r, nd, L.dzk);
                        if (fl == kRowEq) {
                            DS[gi] = adz; // (kept for the multiplier update)
                            continue;
                        }
                        const double sv = Sv[gi], lv = Lam[gi];
                        const double ds = -RP[gi] - adz;
                        const double corr = (pass == 1) ? (sigma_mu - DS[gi] * DL[gi]) : 0.0;
                        const double dl = (corr - lv * sv - lv * ds) / sv;
                        if (pass == 1) { // final direction: overwrite the predictor's
                            L.rowD[r] = ds;
                            L.rowC[r] = dl;
                        } else {
                            DS[gi] = ds;
                            DL[gi] = dl;
                        }
                        if (ds < 0.0) amin = fmin(amin, -sv / ds);
                        if (dl < 0.0) amin = fmin(amin, -lv / dl);
                    }
                    wave_sync();
                    if (pass == 1) // (the corrector's own DS * DL product was read above: now the direction can be stored)
                        for (int r = lane; r < nr; r += kWave) {
                            const int gi = gi0 + r;
                            if ((int)Flag[gi] == kRowIneq) {
                                DS[gi] = L.rowD[r];
                                DL[gi] = L.rowC[r];
                            }
                        }
                    if (k < N) {
                        for (int i = lane; i < nx; i += kWave) { // dx+ = A dx + B du
                            double acc = 0.0;
                            for (int j = 0; j < nz; ++j) acc += L.AB[i + nx * j] * L.dzk[j];
                            L.dxn[i] = acc;
                        }
                        wave_sync();
                        for (int i = lane; i < nx; i += kWave) L.dzk[i] = L.dxn[i];
                    }
                    wave_sync();
                }
                amin = -wave_max(-amin);
                wave_sync_full(); // (DS / DL / DZ are read across lanes below)
                stamp(4);
                if (pass == 0) { // Mehrotra's centring parameter from the affine step
                    const double aaff = fmin(1.0, amin);
                    double acc = 0.0;
                    for (int gi = lane; gi < m; gi += kWave)
                        if ((int)Flag[gi] == kRowIneq) acc += (Sv[gi] + aaff * DS[gi]) * (Lam[gi] + aaff * DL[gi]);
                    const double mu_aff = wave_sum(acc) * inv_mi;
                    const double ratio = mu > 0.0 ? mu_aff / mu : 0.0;
                    sigma_mu = ratio * ratio * ratio * mu;
                } else {
                    const double tau = mu > 1e-10 ? 0.995 : 0.9999;
                    alpha = amin < 1.0 ? fmin(1.0, tau * amin) : 1.0;
                }
            }
            // (the convergence test looks at the FULL Newton step: a step cut short by the positivity of s and lam -- alpha of 1e-4 -- is
            //  small without the iterate being anywhere near a fixed point)
            step_inf = wave_max(step_inf);
            if (!good) break;
            // ---- update
            z_inf = 0.0;
            for (int e = lane; e < NZ; e += kWave) {
                const double v = Z[e] + alpha * DZ[e];
                Z[e] = v;
                z_inf = fmax(z_inf, fabs(v));
            }
            z_inf = wave_max(z_inf);
            double musum2 = 0.0, maxe = 0.0;
            for (int gi = lane; gi < m; gi += kWave) {
                const int fl = (int)Flag[gi];
                if (fl == kRowIneq) {
                    const double sv = Sv[gi] + alpha * DS[gi], lv = Lam[gi] + alpha * DL[gi];
                    Sv[gi] = sv;
                    Lam[gi] = lv;
                    musum2 += sv * lv;
                } else if (fl == kRowEq) {
                    const double re = RP[gi] + alpha * DS[gi]; // residual of the row at the new point
#if !defined(__HIP_DEVICE_COMPILE__) && defined(COPRA_EMU_TRACE)
                    fprintf(stderr, "      eq row %d: residual %.3e, a'dz %.3e -> %.3e; multiplier %.6e -> %.6e\n", gi, RP[gi], DS[gi], re, Lam[gi], Lam[gi] + alpha * (RP[gi] + DS[gi]) / delta);
#endif
                    // Newton direction of the multiplier of the regularised row (a'dz - delta dnu = -(a'z - f)): dnu = (rp + a'dz) / delta, damped
                    // like every other unknown.  Rounds 2-3 added re / delta here -- the same after a full step; after a damped one it differs by
                    // (1 - alpha) rp / delta, 1e9 x a residual that is not small yet: the multiplier was thrown to +-1e7, the iterate with it, and
                    // the complementarity measure reached 1e-15 while the steps were still 1e-4 (a random controller of the differential
                    // test, tests/random_controllers.py seed 1920: accepted 5e-3 from the optimum)
                    Lam[gi] += alpha * (RP[gi] + DS[gi]) / delta;
                    maxe = fmax(maxe, fabs(re));
                }
            }
            wave_sync_full();
            const double mu_new = wave_sum(musum2) * inv_mi;
            // inequality rows carry slacks, so their residuals shrink by exactly 1 - alpha; equality rows are only
            // penalised: their residual is what the new point leaves
            const double res_new = fmax((1.0 - alpha) * maxres, wave_max(maxe));
            if (!(mu_new == mu_new) || !(step_inf == step_inf)) {
                good = false;
                break;
            }
            stamp(6);
#if !defined(__HIP_DEVICE_COMPILE__) && defined(COPRA_EMU_TRACE)
            if (lane == 0) fprintf(stderr, "it %2d alpha %.4f mu %.3e -> %.3e res %.3e (maxres %.3e) step %.3e z %.3e\n", it, alpha, mu, mu_new, res_new, maxres, step_inf, z_inf);
#endif
            if (!polishing && n_ineq > 0 && res_new <= 1e-9 && mu_new <= kRicSwitchMu) {
                cross_over();
                polishing = true;
                prev_step = 1.0e300;
                tail_ok = false;
                continue;
            }
            // (after the crossover at least two steps: the first one removes the residuals of the new equality rows, the multipliers
            //  they inherited -- right to 1e-5 -- are corrected by the second)
            polish_it += polishing ? 1 : 0;
            bool conv = ric_converged(S, res_new, mu_new, step_inf, prev_step, z_inf) && (!polishing || polish_it >= 2);
            tail_ok = !polishing && ric_tail_ok(res_new, mu_new, step_inf, prev_step, z_inf);
            if (conv && polishing) {
                // ---- was the active set the right one?  The solution of the equality-constrained QP is THE optimum iff every held row pushes
                // (multiplier >= 0) and every row left out is satisfied.  A held row that pulls is released, a row left out that is violated
                // is taken, and the iteration goes on on the corrected set -- at most kRicRefinements times, then the Goldfarb-Idnani kernel.
                double flips = 0.0;
                for (int k = 0; k <= N; ++k) {
                    const int c = S.cls_of_stage[k], nr = S.cls_row0[c + 1] - S.cls_row0[c], nd = S.cls_ndense[c], gi0 = S.stage_row0[k];
                    load_class(c);
                    for (int e = lane; e < nz; e += kWave) L.zk[e] = (k == N && e >= nx) ? 0.0 : Z[k * nz + e];
                    wave_sync();
                    for (int r = lane; r < nr; r += kWave) {
                        const int gi = gi0 + r, fl = (int)Flag[gi];
                        if (fl == kRowEq && Sv[gi] == kRicWasActive) {
                            const double lv = Lam[gi] + (row_dot(r, nd, L.zk) - F[gi]) / delta; // (the multiplier the regularised row stands for)
                            if (!(lv >= -1e-9 * (1.0 + fabs(Lam[gi])))) {
                                Flag[gi] = (double)kRowOff;
                                Sv[gi] = kRicWasIdle;
                                Lam[gi] = 0.0;
                                flips += 1.0;
                            }
                        } else if (fl == kRowOff && Sv[gi] == kRicWasIdle && !(row_dot(r, nd, L.zk) - F[gi] <= 1e-9 * (1.0 + fabs(F[gi])))) {
                            Flag[gi] = (double)kRowEq;
                            Sv[gi] = kRicWasActive;
                            Lam[gi] = 0.0;
                            flips += 1.0;
                        }
                    }
                    wave_sync();
                }
                wave_sync_full();
                if (wave_sum(flips) > 0.0) {
                    refinements += 1;
                    if (refinements > kRicRefinements) {
                        good = false;
                        break;
                    }
                    polish_it = 0;
                    prev_step = 1.0e300;
                    conv = false;
                }
            }
            prev_step = step_inf;
            if (conv) {
                converged = true;
                break;
            }
        }

        // (2b of earlier versions -- the check of the crossover's active set -- now happens inside the loop, where a wrong guess is corrected.
        //  History: rounds 2-3 let the barrier run to mu = 1e-15; with weights lam / s of 1e16 a state row takes the curvature of every
        //  direction it touches with it in the Riccati recursion and the computed step is zero there whatever the gradient: the random
        //  differential tests met iterates whose steps were 4e-13 at a point 3e-4 (entry-wise) from the optimum, a bound that should have been
        //  active left 2.6e-5 inside.  A check of the stationarity by an adjoint sweep was tried in between: with multipliers that sit on
        //  slacks of 1e-20, and gradients that unstable dynamics amplify by 2^N, it turned away converged instances.)
        // ------------------------------------------------------------------ 3. results (LMPC.cpp:282-286)
        if (converged) {
            for (int e = lane; e < N * nu; e += kWave) {
                const int k = e / nu, i = e - k * nu;
                P.control[(size_t)inst * P.n + e] = Z[k * nz + nx + i];
            }
            for (int e = lane; e < P.X; e += kWave) {
                const int k = e / nx, i = e - k * nx;
                P.trajectory[(size_t)inst * P.X + e] = Z[k * nz + i];
            }
            if (P.initial_state && P.x0_opt)
                for (int e = lane; e < nx; e += kWave) P.x0_opt[(size_t)inst * nx + e] = Z[e];
            if (lane == 0) {
                P.status[inst] = 0;
                P.iter[2 * (size_t)inst] = it;
                P.iter[2 * (size_t)inst + 1] = 0;
            }
        } else {
            // not converged (infeasible / degenerate): the condensed Goldfarb-Idnani kernel decides its status
            if (lane == 0) {
                P.status[inst] = 3;
                P.iter[2 * (size_t)inst] = it;
                P.iter[2 * (size_t)inst + 1] = 0;
                if (P.ovf_count) P.ovf_list[atomic_append(P.ovf_count)] = inst;
            }
        }
        if (P.prof && lane == 0) {
            stamp(6);
            prof[7] = cycle_counter() - tstart;
            for (int q = 0; q < 8; ++q) P.prof[8 * (size_t)inst + q] = prof[q];
        }
        wave_sync();
    }
}

} // namespace copra_hip
