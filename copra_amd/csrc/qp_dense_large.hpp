// qp_dense_large.hpp -- the batched dense-QP kernel for 64 < n <= 512: SolverInterface::SI_solve arguments
// (reference include/SolverInterface.h:54-80) solved as QuadProgDenseSolver does (src/QuadProgSolver.cpp:45-72), one
// problem per WORKGROUP (gi_large.hpp).  A persistent grid walks the batch so that the per-workgroup J / factor
// workspace stays small and L2 / Infinity-Cache resident.
#pragma once

#include "gi_large.hpp"

namespace copra_hip {

struct DenseRowsLarge {
    const DensePlan& P;
    const double *Aeq, *beq, *Aineq, *bineq, *XL, *XU; // this instance
    const double* nb; // LDS
    double* red;

    COPRA_DEV void begin_scan(const double*) const { }

    COPRA_DEV double coeff(int i, int j) const
    {
        return (i < P.meq) ? Aeq[(size_t)j * P.meq + i] : Aineq[(size_t)j * P.mineq + (i - P.meq)];
    }
    COPRA_DEV double slack(int i, const double* xs) const
    {
        const int n = P.n;
        double ax = 0.0;
        for (int j = 0; j < n; ++j) ax += coeff(i, j) * xs[j];
        return (i < P.meq) ? ax - beq[i] : bineq[i - P.meq] - ax;
    }
    COPRA_DEV double slack_at(int, int i, const double* xs) const { return slack(i, xs); }
    COPRA_DEV double slack_uniform(int p, const double* xs) const
    {
        const int j = bt_tid();
        const double ax = block_sum((j < P.n) ? coeff(p, j) * xs[j] : 0.0, red);
        return (p < P.meq) ? ax - beq[p] : bineq[p - P.meq] - ax;
    }
    COPRA_DEV double norm(int i) const { return nb[i]; }
    COPRA_DEV int normal_extent(int) const { return P.n; }
    COPRA_DEV double ub(int j) const { return XU[j]; }
    COPRA_DEV double lb(int j) const { return XL[j]; }
    COPRA_DEV void load_normal(int p, double sgn, double* np) const
    {
        const int j = bt_tid();
        if (j >= P.n) return;
        np[j] = (p < P.meq) ? sgn * coeff(p, j) : -coeff(p, j);
    }
};

COPRA_DEV void qp_dense_large_body(const DensePlan& P)
{
    double* lds = lds_base();
    const int n = P.n, tid = bt_tid(), T = bt_size();
    const int ld = (n + 7) & ~7;
    double* wsJ = P.ws + (size_t)instance_id() * 2 * n * ld;
    double* wsF = wsJ + (size_t)n * ld;
    LargeSolver S = carve_large(lds, P.llds, n, wsJ, wsF);
    for (int inst = instance_id(); inst < P.batch; inst += instance_stride()) {
        const double* Q = P.Q + (size_t)inst * n * n;
        if (tid < n) { // only the UPPER triangle of Q is read (as by qp_dense.hpp / qpgen2): L(r, c) <- Q(c, r), c <= r
            for (int c = 0; c <= tid; ++c) S.F[(size_t)c * ld + tid] = Q[(size_t)tid * n + c];
            S.cv[tid] = P.c[(size_t)inst * n + tid];
        }
        DenseRowsLarge rows { P, P.Aeq + (size_t)inst * P.meq * n, P.beq + (size_t)inst * P.meq,
            P.Aineq + (size_t)inst * P.mineq * n, P.bineq + (size_t)inst * P.mineq, P.XL + (size_t)inst * n,
            P.XU + (size_t)inst * n, S.nb, S.red };
        for (int i = tid; i < P.mgen; i += T) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) {
                const double a = rows.coeff(i, j);
                s += a * a;
            }
            S.nb[i] = sqrt(s);
        }
        bt_sync();
        int status = gl_factorize(S);
        int it_main = 0, it_drop = 0;
        if (status == 0) {
            gl_invert(S);
            gl_unconstrained(S);
            status = gl_active_set(S, P.meq, P.mgen, rows, P.vsmall, P.max_iter, it_main, it_drop);
        }
        bt_sync();
        const double qnan = __builtin_nan("");
        if (tid < n) P.x[(size_t)inst * n + tid] = (status == 0) ? S.xs[tid] : qnan;
        if (tid == 0) {
            P.fail[inst] = status;
            P.iter[2 * (size_t)inst] = it_main;
            P.iter[2 * (size_t)inst + 1] = it_drop;
        }
        bt_sync();
    }
}

} // namespace copra_hip
