// qp_dense.hpp -- body of the batched dense-QP kernel: plug-in point 1 of the reference, i.e. the arguments of
// SolverInterface::SI_solve (include/SolverInterface.h:54-80) for `batch` independent problems, solved exactly as
// QuadProgDenseSolver does (src/QuadProgSolver.cpp:45-72: bounds appended as [I; -I] rows, then eigen-quadprog).
// One problem per 64-lane wavefront, n <= 64; the constraint matrices stay in HBM (column-major per instance, so
// "lane = row" slack evaluation is coalesced) and only Q/J, R and the vectors live in LDS.
#pragma once

#include "gi_core.hpp"

namespace copra_hip {

struct DenseRows {
    const DensePlan& P;
    const double *Aeq, *beq, *Aineq, *bineq, *XL, *XU; // this instance
    const double* nb; // LDS

    COPRA_DEV void begin_scan(const double*) const { wave_sync(); }

    COPRA_DEV double slack(int i, const double* xs) const
    {
        const int n = P.n;
        if (i < P.meq) {
            double ax = 0.0;
            for (int j = 0; j < n; ++j) ax += Aeq[(size_t)j * P.meq + i] * xs[j];
            return ax - beq[i];
        }
        const int r = i - P.meq;
        double ax = 0.0;
        for (int j = 0; j < n; ++j) ax += Aineq[(size_t)j * P.mineq + r] * xs[j];
        return bineq[r] - ax;
    }

    COPRA_DEV double slack_uniform(int p, const double* xs) const { return slack(p, xs); }
    COPRA_DEV double norm(int i) const { return nb[i]; }
    COPRA_DEV double ub(int j) const { return XU[j]; }
    COPRA_DEV double lb(int j) const { return XL[j]; }

    COPRA_DEV double coeff(int i, int j) const
    {
        return (i < P.meq) ? Aeq[(size_t)j * P.meq + i] : Aineq[(size_t)j * P.mineq + (i - P.meq)];
    }

    COPRA_DEV void load_normal(int p, double sgn, double* ap) const
    {
        const int j = lane_id();
        if (j >= P.n) return;
        ap[j] = (p < P.meq) ? sgn * Aeq[(size_t)j * P.meq + p] : -Aineq[(size_t)j * P.mineq + (p - P.meq)];
    }
};

// NV > 0: number of variables fixed at compile time (copra_qp_dense_specialise), 0: taken from the plan
template <int NV = 0>
COPRA_DEV void qp_dense_body(const DensePlan& P, int inst)
{
    double* lds = lds_base();
    const LdsLayout& L = P.lds;
    const int lane = lane_id();
    const int n = NV ? NV : P.n;
    SolverLds S = carve_solver(lds, L);
    const int ld = NV ? (NV | 1) : S.ldj;
    double* nb = lds + L.nb;
    const double* Q = P.Q + (size_t)inst * n * n;
    for (int e = lane; e < n * n; e += kWave) {
        const int i = e % n, j = e / n; // column-major
        S.J[i * ld + j] = Q[e];
    }
    for (int e = lane; e < n; e += kWave) S.cvec[e] = P.c[(size_t)inst * n + e];
    DenseRows rows { P, P.Aeq + (size_t)inst * P.meq * n, P.beq + (size_t)inst * P.meq,
        P.Aineq + (size_t)inst * P.mineq * n, P.bineq + (size_t)inst * P.mineq, P.XL + (size_t)inst * n,
        P.XU + (size_t)inst * n, nb };
    for (int i = lane; i < P.mgen; i += kWave) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) {
            const double a = rows.coeff(i, j);
            s += a * a;
        }
        nb[i] = sqrt(s);
    }
    wave_sync();
    COPRA_FINE_DECL;
    int status = gi_factorize<NV>(S, n, nullptr COPRA_FINE_PASS);
    int it_main = 0, it_drop = 0;
    if (status == 0) status = gi_active_set<NV>(S, n, P.meq, P.mgen, rows, P.vsmall, P.max_iter, it_main, it_drop COPRA_FINE_PASS);
    wave_sync();
    const double qnan = __builtin_nan("");
    for (int e = lane; e < n; e += kWave) P.x[(size_t)inst * n + e] = (status == 0) ? S.xs[e] : qnan;
    if (lane == 0) {
        P.fail[inst] = status;
        P.iter[2 * (size_t)inst] = it_main;
        P.iter[2 * (size_t)inst + 1] = it_drop;
    }
    wave_sync();
}

} // namespace copra_hip
