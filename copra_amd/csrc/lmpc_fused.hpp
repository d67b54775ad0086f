// lmpc_fused.hpp -- body of the fused "condense + solve" kernel: one MPC instance per 64-lane wavefront.
//
// What one wave does for its instance (reference call stack: LMPC::solve, src/LMPC.cpp:79-101):
//   0. coalesced load of A, B, d, x0 (66 doubles at (6,3)) into LDS;
//   1. PreviewSystem::updateSystem (src/PreviewSystem.cpp:57-74) WITHOUT the O(N^2) Toeplitz fill: only the first
//      block column G_k = A^k B of Psi (Psi_{i,j} = G_{i-1-j}), Phi_k = A^k and xi_k are formed, by the same
//      left-multiplication recursion as the reference;
//   2. cost functions (src/costFunctions.cpp:63-215) summed as LMPC::makeQPForm does (src/LMPC.cpp:228-230,
//      252-255): the Hessian uses the block recursion  Q_{a,b} = Q_{a+1,b+1} + (M G_{K-1-a})' W (M G_{K-1-b}),
//      which adds the per-step terms of the reference's loop in the same (ascending step) order but skips its
//      "lot of sums of zero" (costFunctions.cpp:73);
//   3. constraints (src/constraints.cpp) stay implicit rows (plan.hpp); only their norms are computed;
//   4. QuadProgDenseSolver::SI_solve (src/QuadProgSolver.cpp:54-72): bounds become the 2n implicit rows [I; -I];
//      Goldfarb-Idnani in gi_core.hpp;
//   5. LMPC::updateResults (src/LMPC.cpp:282-286): control = U, trajectory = Phi x0 + Psi U + xi.
#pragma once

#include "gi_core.hpp"

namespace copra_hip {

// Implicit constraint rows of an LMPC problem.
struct StageRows {
    const FusedPlan& P;
    const double* G; // LDS
    const double* Xbar; // LDS
    double* Xcur; // LDS
    const double* nb; // LDS

    // coefficient j of row i in the reference's orientation (row i of Aeq / Aineq), i < mgen
    COPRA_DEV double coeff(int i, int j) const
    {
        const int nx = P.nx, nu = P.nu;
        const int k = P.row_step[i], ek = P.row_ekind[i], eo = P.row_eoff[i];
        const int gk = P.row_gkind[i], go = P.row_goff[i];
        const int jb = j / nu, jc = j - jb * nu;
        double v = 0.0;
        if (ek == kEDense) {
            if (jb < k) {
                const double* Gk = G + (k - 1 - jb) * nx * nu + nx * jc;
                for (int c = 0; c < nx; ++c) v += P.params[eo + c] * Gk[c];
            }
        } else if (ek == kEOneHot) {
            if (jb < k) v = G[(k - 1 - jb) * nx * nu + nx * jc + eo];
        } else if (ek == kEFull) {
            for (int s = jb + 1; s <= P.N; ++s) {
                const double* Gk = G + (s - 1 - jb) * nx * nu + nx * jc;
                for (int c = 0; c < nx; ++c) v += P.params[eo + s * nx + c] * Gk[c];
            }
        }
        if (gk == kGStep) {
            if (jb == k) v += P.params[go + jc];
        } else if (gk == kGFull) {
            v += P.params[go + j];
        }
        return v;
    }

    // E_row . Xv (+ G_row . u when u != nullptr)
    COPRA_DEV double lhs(int i, const double* Xv, const double* u) const
    {
        const int nx = P.nx, nu = P.nu;
        const int k = P.row_step[i], ek = P.row_ekind[i], eo = P.row_eoff[i];
        const int gk = P.row_gkind[i], go = P.row_goff[i];
        double ax = 0.0;
        if (ek == kEDense) {
            for (int c = 0; c < nx; ++c) ax += P.params[eo + c] * Xv[k * nx + c];
        } else if (ek == kEOneHot) {
            ax = Xv[k * nx + eo];
        } else if (ek == kEFull) {
            for (int r = 0; r < P.X; ++r) ax += P.params[eo + r] * Xv[r];
        }
        if (u) {
            if (gk == kGStep) {
                for (int c = 0; c < nu; ++c) ax += P.params[go + c] * u[k * nu + c];
            } else if (gk == kGFull) {
                for (int j = 0; j < P.n; ++j) ax += P.params[go + j] * u[j];
            }
        }
        return ax;
    }

    // trajectory at the current iterate: Xcur = Xbar + Psi U  (Psi implicit)
    COPRA_DEV void refresh_trajectory(const double* xs) const
    {
        const int nx = P.nx, nu = P.nu;
        for (int row = lane_id(); row < P.X; row += kWave) {
            const int k = row / nx, comp = row - k * nx;
            double acc = 0.0;
            for (int jb = 0; jb < k; ++jb) {
                const double* Gk = G + (k - 1 - jb) * nx * nu + comp;
                for (int jc = 0; jc < nu; ++jc) acc += Gk[nx * jc] * xs[jb * nu + jc];
            }
            Xcur[row] = Xbar[row] + acc;
        }
    }

    COPRA_DEV void begin_scan(const double* xs) const
    {
        if (P.any_state_rows) refresh_trajectory(xs);
        wave_sync();
    }

    COPRA_DEV double slack(int i, const double* xs) const
    {
        if (i < P.mgen) {
            const double ax = lhs(i, Xcur, xs);
            const double f = P.row_f[i];
            return (i < P.meq) ? (ax - f) : (f - ax);
        }
        const int j = i - P.mgen;
        if (j < P.n) return P.ub[j] - xs[j]; // row of [I]:  x_j <= XU_j   (QuadProgSolver.cpp:64,67)
        return xs[j - P.n] - P.lb[j - P.n]; // row of [-I]: -x_j <= -XL_j (QuadProgSolver.cpp:65,68)
    }

    COPRA_DEV double norm(int i) const { return (i < P.mgen) ? nb[i] : 1.0; }

    COPRA_DEV void load_normal(int p, double sgn, double* ap) const
    {
        const int j = lane_id();
        if (j >= P.n) return;
        double v;
        if (p < P.mgen) {
            v = coeff(p, j);
            v = (p < P.meq) ? sgn * v : -v;
        } else {
            const int q = p - P.mgen;
            if (q < P.n)
                v = (j == q) ? -1.0 : 0.0;
            else
                v = (j == q - P.n) ? 1.0 : 0.0;
        }
        ap[j] = v;
    }
};

COPRA_DEV void lmpc_fused_body(const FusedPlan& P, int inst)
{
    double* lds = lds_base();
    const LdsLayout& L = P.lds;
    const int lane = lane_id();
    const int nx = P.nx, nu = P.nu, N = P.N, n = P.n, X = P.X;
    double* A = lds + L.A;
    double* B = lds + L.B;
    double* D = lds + L.D;
    double* X0 = lds + L.X0;
    double* G = lds + L.G;
    double* Xbar = lds + L.Xbar;
    double* Xcur = lds + L.Xcur;
    double* nb = lds + L.nb;
    SolverLds S = carve_solver(lds, L);
    const int ld = S.ldj;

    // ---- 0. coalesced loads of this instance's system ----
    for (int e = lane; e < nx * nx; e += kWave) A[e] = P.A[(size_t)inst * nx * nx + e];
    for (int e = lane; e < nx * nu; e += kWave) B[e] = P.B[(size_t)inst * nx * nu + e];
    for (int e = lane; e < nx; e += kWave) {
        D[e] = P.d[(size_t)inst * nx + e];
        X0[e] = P.x0[(size_t)inst * nx + e];
    }
    // ---- 1. preview recursion (PreviewSystem.cpp:57-74) ----
    double* Phi = lds + L.BldPhi; // (N+1) blocks nx x nx
    double* Xi = lds + L.BldXi; // X
    {
        const int nPhi = nx * nx, nG = nx * nu;
        wave_sync();
        for (int e = lane; e < nPhi; e += kWave) {
            const int r = e % nx, c = e / nx;
            Phi[e] = (r == c) ? 1.0 : 0.0; // Phi_0 = I (PreviewSystem.cpp:51)
            Phi[nPhi + e] = A[e]; // Phi_1 = A (:59)
        }
        for (int e = lane; e < nG; e += kWave) G[e] = B[e]; // Psi_{1,0} = B (:60)
        for (int e = lane; e < nx; e += kWave) {
            Xi[e] = 0.0;
            Xi[nx + e] = D[e]; // xi_1 = d (:61)
        }
        const int per_step = nPhi + nG + nx;
        for (int s = 2; s <= N; ++s) {
            wave_sync();
            for (int e = lane; e < per_step; e += kWave) {
                if (e < nPhi) { // Phi_s = A Phi_{s-1} (:64)
                    const int r = e % nx, c = e / nx;
                    const double* prev = Phi + (s - 1) * nPhi + c * nx;
                    double acc = 0.0;
                    for (int t = 0; t < nx; ++t) acc += A[r + nx * t] * prev[t];
                    Phi[s * nPhi + e] = acc;
                } else if (e < nPhi + nG) { // Psi_{s,0} = A Psi_{s-1,0}, i.e. G_{s-1} = A G_{s-2} (:65)
                    const int g = e - nPhi;
                    const int r = g % nx, c = g / nx;
                    const double* prev = G + (s - 2) * nG + c * nx;
                    double acc = 0.0;
                    for (int t = 0; t < nx; ++t) acc += A[r + nx * t] * prev[t];
                    G[(s - 1) * nG + g] = acc;
                } else { // xi_s = A xi_{s-1} + d (:70)
                    const int r = e - nPhi - nG;
                    const double* prev = Xi + (s - 1) * nx;
                    double acc = 0.0;
                    for (int t = 0; t < nx; ++t) acc += A[r + nx * t] * prev[t];
                    Xi[s * nx + r] = acc + D[r];
                }
            }
        }
        wave_sync();
        // free response  Xbar = Phi x0 + xi
        for (int row = lane; row < X; row += kWave) {
            const int k = row / nx, r = row - k * nx;
            const double* Pk = Phi + k * nPhi + r;
            double acc = 0.0;
            for (int c = 0; c < nx; ++c) acc += Pk[nx * c] * X0[c];
            Xbar[row] = acc + Xi[row];
        }
    }
    // ---- 2. Hessian and gradient: Q = 1e-6 I + sum Q_k, c = sum c_k (LMPC.cpp:228-230, 252-255) ----
    {
        double* Q = S.J;
        for (int e = lane; e < n * ld; e += kWave) Q[e] = 0.0;
        wave_sync();
        if (lane < n) {
            double one = 1.0;
            one *= 1e-6; // Q_.setIdentity(); Q_ *= 1e-6;
            Q[lane * ld + lane] = one;
        }
        double cj = 0.0; // lane j accumulates c_j
        double* Y = lds + L.BldY;
        double* We = lds + L.BldWe;
        const int a_blk = lane / nu, ic = lane - a_blk * nu; // lane as row index i = (a, ic) / column j = (b, jc)
        for (int t = 0; t < P.ncost; ++t) {
            const CostTerm& ct = P.cost[t];
            const int r = ct.rows;
            const double* w = P.params + ct.offW;
            const double* p = P.params + ct.offP;
            wave_sync();
            if (ct.kind == kCostControl) {
                // ControlCost::update (costFunctions.cpp:148-156): block-diagonal N'WN, c = -p'WN
                const double* Nm = P.params + ct.offN; // r x nu
                if (lane < n) {
                    const int jc = ic, b = a_blk;
                    for (int i2 = 0; i2 < nu; ++i2) {
                        double acc = 0.0;
                        for (int k = 0; k < r; ++k) acc += (Nm[k + r * i2] * w[k]) * Nm[k + r * jc];
                        Q[(b * nu + i2) * ld + lane] += acc;
                    }
                    double acc = 0.0;
                    for (int k = 0; k < r; ++k) acc += ((-p[k]) * w[k]) * Nm[k + r * jc];
                    cj += acc;
                }
                continue;
            }
            const double* M = P.params + ct.offM; // r x nx
            const double* Nm = (ct.kind == kCostMixed) ? P.params + ct.offN : nullptr;
            // Y_k = M G_k (the block  M * Psi_{i, j}  with k = i-1-j), We_k = w .* (M xbar_k - p)
            for (int e = lane; e < N * r * nu; e += kWave) {
                const int k = e / (r * nu), rem = e - k * r * nu;
                const int jc = rem / r, row = rem - jc * r;
                const double* Gk = G + k * nx * nu + nx * jc;
                double acc = 0.0;
                for (int c = 0; c < nx; ++c) acc += M[row + r * c] * Gk[c];
                Y[e] = acc; // Y[k][row + r*jc]
            }
            for (int e = lane; e < (N + 1) * r; e += kWave) {
                const int k = e / r, row = e - k * r;
                double acc = 0.0;
                for (int c = 0; c < nx; ++c) acc += M[row + r * c] * Xbar[k * nx + c];
                We[e] = (acc - p[row]) * w[row];
            }
            wave_sync();
            // last state index K that enters the sum: trajectory K = N, mixed K = N-1, target only K = N
            const int K = (ct.kind == kCostMixed) ? N - 1 : N;
            const bool recur = (ct.kind != kCostTarget);
            double val[kMaxNu];
#pragma unroll
            for (int jc = 0; jc < kMaxNu; ++jc) val[jc] = 0.0;
            for (int b = N - 1; b >= 0; --b) {
                const int ka = K - 1 - a_blk, kb = K - 1 - b;
                const bool have = (lane < n) && ka >= 0 && kb >= 0;
#pragma unroll
                for (int jc = 0; jc < kMaxNu; ++jc) {
                    if (jc < nu) {
                        double pterm = 0.0;
                        if (have) {
                            const double* Ya = Y + ka * r * nu + r * ic;
                            const double* Yb = Y + kb * r * nu + r * jc;
                            for (int k = 0; k < r; ++k) pterm += (Ya[k] * w[k]) * Yb[k];
                        }
                        const double carried = recur ? shfl_down0_f64(val[jc], nu) : 0.0;
                        val[jc] = (lane < n) ? carried + pterm : 0.0;
                        if (lane < n) {
                            double add = val[jc];
                            if (Nm) { // MixedCost cross terms of step k = b (costFunctions.cpp:207)
                                double cross = 0.0;
                                bool has_cross = false;
                                if (a_blk < b) { // (M G_{b-1-a})' W N
                                    const double* Ya = Y + (b - 1 - a_blk) * r * nu + r * ic;
                                    for (int k = 0; k < r; ++k) cross += (Ya[k] * w[k]) * Nm[k + r * jc];
                                    has_cross = true;
                                } else if (a_blk == b) { // N' W N
                                    for (int k = 0; k < r; ++k) cross += (Nm[k + r * ic] * w[k]) * Nm[k + r * jc];
                                    has_cross = true;
                                }
                                if (has_cross) add += cross;
                            }
                            Q[lane * ld + b * nu + jc] += add;
                        }
                    }
                }
            }
            // gradient: c_j = sum_k tmp_k(:,j)' We_k  (ascending step order, costFunctions.cpp:78,80 / :106 / :211)
            if (lane < n) {
                const int b = a_blk, jc = ic;
                double acc = 0.0;
                if (ct.kind == kCostTarget) {
                    const double* Yb = Y + (N - 1 - b) * r * nu + r * jc;
                    for (int k = 0; k < r; ++k) acc += We[N * r + k] * Yb[k];
                } else {
                    if (Nm) { // step k = b of MixedCost: tmp_b(:, j) = N
                        double s0 = 0.0;
                        for (int k = 0; k < r; ++k) s0 += We[b * r + k] * Nm[k + r * jc];
                        acc += s0;
                    }
                    for (int s = b + 1; s <= K; ++s) {
                        const double* Yb = Y + (s - 1 - b) * r * nu + r * jc;
                        const double* Ws = We + s * r;
                        double sk = 0.0;
                        for (int k = 0; k < r; ++k) sk += Ws[k] * Yb[k];
                        acc += sk;
                    }
                }
                cj += acc;
            }
        }
        wave_sync();
        if (lane < n) S.cvec[lane] = cj;
        wave_sync();
        if (inst == P.dump_instance && P.dumpQ) { // parity hook (LMPC::Q(), LMPC::c(); LMPC.h:113-115)
            if (lane < n) {
                for (int i = 0; i < n; ++i) {
                    const double v = (i <= lane) ? Q[i * ld + lane] : Q[lane * ld + i];
                    P.dumpQ[(size_t)lane * n + i] = v;
                }
                P.dumpc[lane] = cj;
            }
        }
    }
    // ---- 3. implicit rows: norms (qpgen2: column norms of amat) ----
    StageRows rows { P, G, Xbar, Xcur, nb };
    for (int i = lane; i < P.mgen; i += kWave) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) {
            const double a = rows.coeff(i, j);
            s += a * a;
        }
        nb[i] = sqrt(s);
    }
    if (inst == P.dump_instance && P.dumpA) { // parity hook (LMPC::Aeq/beq/Aineq/bineq; LMPC.h:116-123)
        for (int i = lane; i < P.mgen; i += kWave) {
            for (int j = 0; j < n; ++j) P.dumpA[(size_t)j * P.mgen + i] = rows.coeff(i, j);
            P.dumpb[i] = P.row_f[i] - rows.lhs(i, Xbar, nullptr); // b = z - Y x0 (constraints.cpp:81)
        }
    }
    wave_sync();
    if (P.dump_only) return;
    // ---- 4. + 5. solve ----
    int status = gi_factorize(S, n);
    int it_main = 0, it_drop = 0;
    if (status == 0) status = gi_active_set(S, n, P.meq, P.mtotal, rows, P.vsmall, P.max_iter, it_main, it_drop);
    wave_sync();
    // ---- 6. results (LMPC.cpp:95-97: outputs only on success; failures are flagged with NaN) ----
    if (status == 0) {
        rows.refresh_trajectory(S.xs);
        wave_sync();
        for (int e = lane; e < n; e += kWave) P.control[(size_t)inst * n + e] = S.xs[e];
        for (int e = lane; e < X; e += kWave) P.trajectory[(size_t)inst * X + e] = Xcur[e];
    } else {
        const double qnan = __builtin_nan("");
        for (int e = lane; e < n; e += kWave) P.control[(size_t)inst * n + e] = qnan;
        for (int e = lane; e < X; e += kWave) P.trajectory[(size_t)inst * X + e] = qnan;
    }
    if (lane == 0) {
        P.status[inst] = status;
        P.iter[2 * (size_t)inst] = it_main;
        P.iter[2 * (size_t)inst + 1] = it_drop;
    }
}

} // namespace copra_hip
