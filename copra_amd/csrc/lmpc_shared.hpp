// lmpc_shared.hpp -- shared-model receding-horizon fast path: every instance of the batch has the
// SAME preview system (A, B, d) and the same costs / constraints; only x0 differs (PreviewSystem::xInit between
// solves, include/PreviewSystem.h:52-54, with isUpdated left true so that LMPC::updateSystem does not rebuild Psi,
// src/LMPC.cpp:233).  Then Psi, the Hessian, its Cholesky factor and J = R^-1 are the same for the whole batch: a
// one-workgroup "prepare" launch of the fused kernel computes them once (lmpc_fused.hpp, model_out) and each wave here
// only
//   * forms the free response  xbar = Phi x0 + xi  (the reference factors the x0-dependence the same way: c = E' x0 + f,
//     costFunctions.cpp:80; b = z - Y x0, constraints.cpp:81),
//   * takes the unconstrained minimiser x = -Qinv (c0 + C1 x0 + C2 p) = xu0 + K1 x0 + K2 p (multiplied out once per
//     batch) and copies the factor into its LDS only if a constraint is violated,
//   * runs the same Goldfarb-Idnani loop (gi_core.hpp) and writes the results.
// Same LDS layouts and the same two-tier overflow scheme as the fused kernel.
#pragma once

#include "lmpc_fused.hpp"

namespace copra_hip {

// TRI_: factor-only first tier (LdsLayout::tri) -- the instance never owns an n x n matrix: the batch-wide Cholesky
// factor is copied (packed, half the size of J) on first need and stays read-only; Q1 / Rq carry the active set.
template <int NX_, int NU_, int NH_, bool TRI_ = false>
COPRA_DEV void lmpc_shared_body(const FusedPlan& P, int inst)
{
    double* lds = lds_base();
    const LdsLayout& L = P.lds;
    const int lane = lane_id();
    const int nx = NX_ ? NX_ : P.nx, nu = NU_ ? NU_ : P.nu, N = NH_ ? NH_ : P.N;
    const int n = nu * N, X = nx * (N + 1);
    constexpr int NV = NU_ * NH_;
    double* G = lds + L.G;
    double* Xbar = lds + L.Xbar;
    double* Xcur = lds + L.Xcur;
    double* nb = lds + L.nb;
    SolverLds S = carve_solver(lds, L);
    const int ld = NV ? (NV | 1) : S.ldj;
    const ModelLayout m = model_layout(nx, nu, N, n, X, ld, P.mgen);
    const double* M = P.model;
    COPRA_FINE_DECL;

    StageRows<NX_, NU_, NH_> rows { P, G, Xbar, Xcur, nb, RowDesc {}, 0.0, 0.0 };
    rows.inst = inst;
    rows.zero = S.scal;
    if (lane == 0) S.scal[0] = 0.0;
    rows.cache_own_row();
    int status = (int)M[m.status];
    // ---- this instance's x0; shared tables -> LDS ----
    const double* x0 = P.x0 + (size_t)inst * nx;
    double x0r[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) x0r[c] = (c < nx) ? x0[c] : 0.0;
    S.Jsrc = TRI_ ? M + m.Rtri : M + m.J; // copied into LDS only if a constraint turns out to be violated (gi_active_set)
    S.rinv_src = M + m.rinv;
    for (int e = lane; e < N * nx * nu; e += kWave) G[e] = M[m.G + e];
    for (int e = lane; e < P.mgen; e += kWave) nb[e] = M[m.nb + e];
    for (int row = lane; row < X; row += kWave) { // free response  xbar = Phi x0 + xi
        const int k = row / nx, r = row - k * nx;
        const double* Pk = M + m.Phi + k * nx * nx + r;
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < nx) acc += Pk[nx * c] * x0r[c];
        for (int c = 16; c < nx; ++c) acc += Pk[nx * c] * x0[c]; // (xDim > 16: the tail straight from HBM)
        Xbar[row] = acc + M[m.Xi + row];
    }
    // ---- unconstrained minimiser: x = -Qinv (c0 + C1 x0 + C2 p) is affine in (x0, p); the prepare step multiplied it
    //      out once for the batch (xu0, K1, K2: nx + R terms per lane instead of an n x n product; c itself is not
    //      needed any more -- the active-set loop works from x and the constraint rows) ----
    auto unconstrained = [&]() {
    if (lane < n) {
        double acc = M[m.xu0 + lane];
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < nx) acc += M[m.K1 + (size_t)c * n + lane] * x0r[c];
        for (int c = 16; c < nx; ++c) acc += M[m.K1 + (size_t)c * n + lane] * x0[c];
        for (int t = 0; t < P.ncost; ++t) { // costs with per-instance references
            if (P.cost_p[t] && P.model_ref_off[t] >= 0) {
                // (prows: the rows of the cost, or rows x steps for a reference trajectory -- CostTerm::pstride)
                const double* pt = P.cost_p[t] + (size_t)inst * P.cost[t].prows;
                const double* K2 = M + m.C2 + (size_t)(P.model_rtot + P.model_ref_off[t]) * n;
                for (int i = 0; i < P.cost[t].prows; ++i) acc += K2[(size_t)i * n + lane] * pt[i];
            }
        }
        S.xs[lane] = acc;
    }
    };
    unconstrained();
    wave_sync();
    int it_main = 0, it_drop = 0;
    int* wset = P.warm_set ? P.warm_set + (size_t)inst * kWarmCap : nullptr; // previous tick's active set, shifted
    int nact = 0;
    if (status == 0) {
        const int wmine = (wset && lane < kWarmCap) ? wset[lane] : -1; // one coalesced load; read back with v_readlane
        status = gi_active_set<NV, TRI_>(S, n, P.meq, P.mgen, rows, P.vsmall, P.max_iter, it_main, it_drop COPRA_FINE_PASS,
            false, wmine, wset ? kWarmCap : 0, &nact);
    }
    wave_sync();
    if (wset && status == 0) { // remember the active set, moved one step towards the present (step 0 rows leave)
        for (int k = lane; k < kWarmCap; k += kWave) {
            int nw = -1;
            if (k < nact) {
                const int idx = S.iact[k];
                if (idx < P.mgen) {
                    nw = P.row_prev[idx];
                } else {
                    const int q = idx - P.mgen, side = q / n, j = q - side * n;
                    nw = (j >= nu) ? P.mgen + side * n + (j - nu) : -1;
                }
            }
            wset[k] = nw; // (rows that moved out of the horizon leave -1 gaps: the reader skips them)
        }
    }
    if (status == 4) { // R outgrew the compact layout: queue for the second (full-layout) launch
        if (lane == 0) {
            const int slot = atomic_append(P.ovf_count);
            P.ovf_list[slot] = inst;
            P.status[inst] = 4;
        }
        return;
    }
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
