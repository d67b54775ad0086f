// islmpc_fused.hpp -- fused kernel body of the InitialStateLMPC variant (reference: include/InitialStateLMPC.h,
// src/InitialStateLMPC.cpp), one instance per wavefront, decision vector v = [x0; U] with nx + nu*N <= 64.
//
// What changes with respect to lmpc_fused.hpp (InitialStateLMPC::makeQPForm, InitialStateLMPC.cpp:77-122):
//   * costs contribute  Q (bottom-right), E = sum_k (M Phi_k)' W tmp_k (top-right, :82) and f = sum_k (M xi_k - p)' W tmp_k
//     (linear term of U, :83 -- NOT c = E'x0 + f);  the top-left block is  R + E Q^-1 E'  (:117).  The reference forms
//     Q.inverse() explicitly; here Q = Rq'Rq is factorised once more by the same wave-level Cholesky and
//     E Q^-1 E' = (E Jq)(E Jq)' with Jq = Rq^-1;
//   * constraint rows are [Y | A] v <=|= z with Y = E_row Phi_k, z = f - E_row xi_k (:88-102): with the implicit-row
//     machinery this is the same  E_row x_k + G_row u_k <=|= f  as before, only x_k now depends on the variable x0;
//   * bounds: [x0lb; lb] <= v <= [x0ub; ub] (:105-121);  results: control = tail, trajectory = Phi x0* + Psi U + xi (:124-128).
// Generic (run-time shape) code only: this variant is an API-completeness path, not the headline.
#pragma once

#include "lmpc_fused.hpp"

namespace copra_hip {

struct StageRowsIS {
    const FusedPlan& P;
    StageRows<0, 0, 0> base; // U-part coefficients and the E.x + G.u evaluation
    const double* Phi; // LDS: (N+1) blocks nx x nx
    const double* Xi; // LDS
    double ubv, lbv; // this lane's bounds of v

    COPRA_DEV int nx() const { return P.nx; }
    COPRA_DEV int nvar() const { return P.nx + P.n; }

    // coefficient of variable j (0..nx-1: x0 part = Y_row, then the U part = A_row)
    COPRA_DEV double coeff(const RowDesc& d, int j) const
    {
        if (j >= nx()) return base.coeff(d, j - nx());
        const int a = j, k = d.k, eo = d.eo, nPhi = nx() * nx();
        double v = 0.0;
        if (e_onehot(d.ek)) {
            v = e_sign(d.ek) * Phi[k * nPhi + eo + nx() * a];
        } else if (d.ek == kEDense) {
            for (int c = 0; c < nx(); ++c) v += P.params[eo + c] * Phi[k * nPhi + c + nx() * a];
        } else if (d.ek == kEFull) {
            for (int s = 0; s <= P.N; ++s)
                for (int c = 0; c < nx(); ++c) v += P.params[eo + s * nx() + c] * Phi[s * nPhi + c + nx() * a];
        }
        return v;
    }
    COPRA_DEV double norm2(const RowDesc& d) const
    {
        double s = 0.0;
        for (int j = 0; j < nvar(); ++j) {
            const double a = coeff(d, j);
            s += a * a;
        }
        return s;
    }
    // X = Phi x0 + xi + Psi U with x0 = xs[0..nx), U = xs[nx..)
    COPRA_DEV void refresh_trajectory(const double* xs) const
    {
        const int nPhi = nx() * nx(), nu = P.nu;
        for (int row = lane_id(); row < P.X; row += kWave) {
            const int k = row / nx(), comp = row - k * nx();
            double a0 = Xi[row];
            for (int a = 0; a < nx(); ++a) a0 += Phi[k * nPhi + comp + nx() * a] * xs[a];
            const double* g = base.G + comp + (k - 1) * nx() * nu;
            for (int jb = 0; jb < k; ++jb)
                for (int jc = 0; jc < nu; ++jc) a0 += g[-jb * nx() * nu + nx() * jc] * xs[nx() + jb * nu + jc];
            base.Xcur[row] = a0;
        }
    }
    COPRA_DEV void begin_scan(const double* xs) const
    {
        refresh_trajectory(xs);
        wave_sync();
    }
    COPRA_DEV double slack(int i, const double* xs) const
    {
        const RowDesc d = base.load_desc(i);
        const double ax = base.lhs(d, base.Xcur, xs + nx());
        return (i < P.meq) ? (ax - d.f) : (d.f - ax);
    }
    COPRA_DEV double slack_uniform(int p, const double* xs) const { return slack(uniform_i32(p), xs); }
    COPRA_DEV double norm(int i) const { return base.nb[i]; }
    COPRA_DEV double ub(int) const { return ubv; }
    COPRA_DEV double lb(int) const { return lbv; }
    COPRA_DEV void load_normal(int p, double sgn, double* ap) const
    {
        const int j = lane_id();
        if (j >= nvar()) return;
        const RowDesc d = base.load_desc(uniform_i32(p));
        const double v = coeff(d, j);
        ap[j] = (p < P.meq) ? sgn * v : -v;
    }
};

COPRA_DEV void islmpc_fused_body(const FusedPlan& P, int inst)
{
    double* lds = lds_base();
    const LdsLayout& L = P.lds;
    const IsLayout& I = P.isl;
    const int lane = lane_id();
    const int nx = P.nx, nu = P.nu, N = P.N, n = P.n, X = P.X;
    const int nv = nx + n;
    COPRA_FINE_DECL;
    double* A = lds + L.A;
    double* B = lds + L.B;
    double* D = lds + L.D;
    double* G = lds + L.G;
    double* Xcur = lds + L.Xcur;
    double* nb = lds + L.nb;
    double* Phi = lds + L.BldPhi;
    double* Xi = lds + L.BldXi;
    double* Jq = lds + I.Jq; // n x ldq: Hessian of the U block -> Rq -> Jq = Rq^-1
    double* Eb = lds + I.E; // nx x n, column-major: top-right block
    double* MPhi = lds + I.MPhi; // (N+1) blocks r x nx
    SolverLds S = carve_solver(lds, L); // nv variables
    const int ld = S.ldj, ldq = I.ldq;

    // ---- 0./1. load + preview recursion (identical to lmpc_fused.hpp, run-time shapes) ----
    for (int e = lane; e < nx * nx; e += kWave) A[e] = P.A[(size_t)inst * nx * nx + e];
    for (int e = lane; e < nx * nu; e += kWave) B[e] = P.B[(size_t)inst * nx * nu + e];
    for (int e = lane; e < nx; e += kWave) D[e] = P.d[(size_t)inst * nx + e];
    const int nPhi = nx * nx, nG = nx * nu;
    wave_sync();
    for (int e = lane; e < nPhi; e += kWave) {
        const int r = e % nx, c = e / nx;
        Phi[e] = (r == c) ? 1.0 : 0.0;
        Phi[nPhi + e] = A[e];
    }
    for (int e = lane; e < nG; e += kWave) G[e] = B[e];
    for (int e = lane; e < nx; e += kWave) {
        Xi[e] = 0.0;
        Xi[nx + e] = D[e];
    }
    const int per_step = nx * (nx + nu + 1);
    for (int s = 2; s <= N; ++s) {
        wave_sync();
        for (int e = lane; e < per_step; e += kWave) {
            const int c = e / nx, r = e - c * nx;
            const double* src;
            double* dst;
            double add = 0.0;
            if (c < nx) {
                src = Phi + (s - 1) * nPhi + c * nx;
                dst = Phi + s * nPhi + c * nx + r;
            } else if (c < nx + nu) {
                src = G + (s - 2) * nG + (c - nx) * nx;
                dst = G + (s - 1) * nG + (c - nx) * nx + r;
            } else {
                src = Xi + (s - 1) * nx;
                dst = Xi + s * nx + r;
                add = D[r];
            }
            double acc = 0.0;
            for (int t = 0; t < nx; ++t) acc += A[r + nx * t] * src[t];
            *dst = acc + add;
        }
    }
    wave_sync();

    // ---- 2. costs: Q (U block, into Jq), E (nx x n), f ----
    for (int e = lane; e < n * ldq; e += kWave) Jq[e] = 0.0;
    for (int e = lane; e < nx * n; e += kWave) Eb[e] = 0.0;
    for (int e = lane; e < nv * ld; e += kWave) S.J[e] = 0.0;
    wave_sync();
    if (lane < n) {
        double one = 1.0;
        one *= 1e-6; // LMPC.cpp:228-229 on the whole Hessian; only the U block survives (InitialStateLMPC.cpp:117)
        Jq[lane * ldq + lane] = one;
    }
    double fj = 0.0; // lane j < n accumulates f_j
    double* Y = lds + L.BldY;
    double* Wf = lds + L.BldWe;
    double* Cp = lds + L.BldCp;
    const int blk = lane / nu, sub = lane - blk * nu;
    for (int t = 0; t < P.ncost; ++t) {
        const CostTerm& ct = P.cost[t];
        const int r = ct.rows;
        wave_sync();
        if (ct.full) {
            // Full-size entry (costFunctions.cpp:65-71, 141-146, 197-203): tmp = M Psi (+ N) is R x n and dense;
            //   Q += tmp' W tmp,  E += (M Phi)' W tmp,  f += (M xi - p)' W tmp        (one cost row at a time:
            // row `rr` of tmp goes through LDS, every lane adds its column of the rank-1 update).  API-completeness
            // path (TestLMPC_InitialState.cpp runs all nine classes with full-size entries), not a fast one.
            const double* Mr = (ct.offM >= 0) ? P.params + ct.offM : nullptr; // R x X, row-major
            const double* Nr = (ct.offN >= 0) ? P.params + ct.offN : nullptr; // R x n, row-major
            const double* pp = cost_reference(P, t, inst);
            const double* ww = P.params + ct.offW;
            double* rowbuf = S.dv; // n doubles (the solver vectors are not live yet)
            double* mphi = S.xs; // (M Phi)(rr, 0..nx-1) and the residual (M xi - p)(rr)
            const int jb = lane / nu, jc = lane - jb * nu;
            for (int rr = 0; rr < r; ++rr) {
                wave_sync();
                double tv = 0.0;
                if (lane < n) {
                    if (Mr) { // row rr of M times column `lane` of Psi: Psi_{s, jb} = G_{s-1-jb} for s > jb
                        const double* mrow = Mr + (size_t)rr * X;
                        for (int s2 = jb + 1; s2 <= N; ++s2) {
                            const double* Gk = G + (s2 - 1 - jb) * nx * nu + nx * jc;
                            for (int c = 0; c < nx; ++c) tv += mrow[s2 * nx + c] * Gk[c];
                        }
                    }
                    if (Nr) tv += Nr[(size_t)rr * n + lane];
                    rowbuf[lane] = tv;
                }
                if (lane <= nx) {
                    double acc = 0.0;
                    if (Mr) {
                        const double* mrow = Mr + (size_t)rr * X;
                        if (lane < nx) {
                            for (int s2 = 0; s2 <= N; ++s2)
                                for (int c = 0; c < nx; ++c) acc += mrow[s2 * nx + c] * Phi[s2 * nPhi + c + nx * lane];
                        } else {
                            for (int col = 0; col < X; ++col) acc += mrow[col] * Xi[col];
                        }
                    }
                    mphi[lane] = (lane < nx) ? acc : acc - pp[rr];
                }
                wave_sync();
                const double wr = ww[rr];
                if (lane < n) {
                    for (int i = 0; i <= lane; ++i) Jq[i * ldq + lane] += (rowbuf[i] * wr) * tv;
                    fj += (mphi[nx] * wr) * tv;
                    for (int a = 0; a < nx; ++a) Eb[a + nx * lane] += (mphi[a] * wr) * tv;
                }
            }
            continue;
        }
        double* Mx = Cp;
        double* Nm = Cp + r * nx;
        double* p = Nm + r * nu;
        double* w = p + r;
        for (int e = lane; e < r * nx; e += kWave) Mx[e] = (ct.offM >= 0) ? P.params[ct.offM + e] : 0.0;
        for (int e = lane; e < r * nu; e += kWave) Nm[e] = (ct.offN >= 0) ? P.params[ct.offN + e] : 0.0;
        for (int e = lane; e < r; e += kWave) {
            p[e] = cost_reference(P, t, inst)[e];
            w[e] = P.params[ct.offW + e];
        }
        wave_sync();
        if (ct.kind == kCostControl) { // costFunctions.cpp:148-156: Q blocks, E = 0, f = -p'WN
            if (lane < n) {
                for (int i2 = 0; i2 < nu; ++i2) {
                    double acc = 0.0;
                    for (int k = 0; k < r; ++k) acc += (Nm[k + r * i2] * w[k]) * Nm[k + r * sub];
                    Jq[(blk * nu + i2) * ldq + lane] += acc;
                }
                double acc = 0.0;
                for (int k = 0; k < r; ++k) acc += ((-p[k]) * w[k]) * Nm[k + r * sub];
                fj += acc;
            }
            continue;
        }
        const bool mixed = (ct.kind == kCostMixed);
        for (int e = lane; e < N * r * nu; e += kWave) { // Y_k = M G_k
            const int k = e / (r * nu), rem = e - k * r * nu;
            const int jc = rem / r, row = rem - jc * r;
            const double* Gk = G + k * nx * nu + nx * jc;
            double acc = 0.0;
            for (int c = 0; c < nx; ++c) acc += Mx[row + r * c] * Gk[c];
            Y[e] = acc;
        }
        for (int e = lane; e < (N + 1) * r; e += kWave) { // Wf_k = w .* (M xi_k - p)   (costFunctions.cpp:78)
            const int k = e / r, row = e - k * r;
            double acc = 0.0;
            for (int c = 0; c < nx; ++c) acc += Mx[row + r * c] * Xi[k * nx + c];
            Wf[e] = (acc - p[row]) * w[row];
        }
        for (int e = lane; e < (N + 1) * r * nx; e += kWave) { // MPhi_k = M Phi_k          (costFunctions.cpp:77)
            const int k = e / (r * nx), rem = e - k * r * nx;
            const int a = rem / r, row = rem - a * r;
            double acc = 0.0;
            for (int c = 0; c < nx; ++c) acc += Mx[row + r * c] * Phi[k * nPhi + c + nx * a];
            MPhi[e] = acc; // MPhi[k][row + r*a]
        }
        wave_sync();
        const int K = mixed ? N - 1 : N;
        const bool accumulate = (ct.kind != kCostTarget);
        if (lane < n) {
            // Hessian of the U block: same block-diagonal walk as lmpc_fused.hpp
            const int delta = blk, ic = sub;
            double val[kMaxNu], cross[kMaxNu];
            for (int jc = 0; jc < kMaxNu; ++jc) val[jc] = cross[jc] = 0.0;
            if (mixed) {
                for (int jc = 0; jc < nu; ++jc) {
                    double acc = 0.0;
                    if (delta > 0) {
                        const double* Ya = Y + (delta - 1) * r * nu + r * ic;
                        for (int k = 0; k < r; ++k) acc += (Ya[k] * w[k]) * Nm[k + r * jc];
                    } else {
                        for (int k = 0; k < r; ++k) acc += (Nm[k + r * ic] * w[k]) * Nm[k + r * jc];
                    }
                    cross[jc] = acc;
                }
            }
            for (int b = N - 1; b >= delta; --b) {
                const int a = b - delta, m = K - 1 - b;
                for (int jc = 0; jc < nu; ++jc) {
                    double pterm = 0.0;
                    if (m >= 0) {
                        const double* Ya = Y + (m + delta) * r * nu + r * ic;
                        const double* Yb = Y + m * r * nu + r * jc;
                        for (int k = 0; k < r; ++k) pterm += (Ya[k] * w[k]) * Yb[k];
                    }
                    val[jc] = accumulate ? val[jc] + pterm : pterm;
                    Jq[(a * nu + ic) * ldq + b * nu + jc] += mixed ? val[jc] + cross[jc] : val[jc];
                }
            }
            // E(:, j) and f_j for column j = (b, jc) of the U block, steps in ascending order
            const int b = blk, jc = sub;
            const int k_lo = (ct.kind == kCostTarget) ? N : (mixed ? b : b + 1);
            for (int k = k_lo; k <= K; ++k) {
                const double* tk = (k == b) ? (Nm + r * jc) : (Y + (k - 1 - b) * r * nu + r * jc); // tmp_k(:, j)
                double sf = 0.0;
                for (int q = 0; q < r; ++q) sf += Wf[k * r + q] * tk[q];
                fj += sf;
                for (int a = 0; a < nx; ++a) {
                    double se = 0.0;
                    for (int q = 0; q < r; ++q) se += (MPhi[k * r * nx + q + r * a] * w[q]) * tk[q];
                    Eb[a + nx * lane] += se;
                }
            }
        }
    }
    wave_sync();
    if (P.denseQ >= 0 && lane < n) { // host-evaluated user cost functions: Q += Q_, E += E_, f += f_ (InitialStateLMPC.cpp:80-84)
        const double* Qd = P.params + P.denseQ + (size_t)n * lane;
        for (int i = 0; i <= lane; ++i) Jq[i * ldq + lane] += Qd[i];
        for (int a = 0; a < nx; ++a) Eb[a + nx * lane] += P.params[P.denseE + a + nx * lane];
        fj += P.params[P.densef + lane];
    }
    // ---- assemble [[R + E Q^-1 E', E], [E', Q]] (upper triangle) and [r; f] ----
    if (lane < n) {
        for (int i = 0; i <= lane; ++i) S.J[(nx + i) * ld + nx + lane] = Jq[i * ldq + lane];
        for (int a = 0; a < nx; ++a) S.J[a * ld + nx + lane] = Eb[a + nx * lane];
        S.cvec[nx + lane] = fj;
    }
    if (lane < nx) S.cvec[lane] = P.is_r[lane];
    wave_sync();
    int status = 0;
    {
        // Q = Rq'Rq, Jq = Rq^-1 with the same wave-level routine (its by-products xs / dv are scratch here)
        SolverLds Sq = S;
        Sq.J = Jq;
        Sq.ldj = ldq;
        Sq.cvec = S.ap; // any n zeros: the unconstrained minimiser it also computes is not used
        if (lane < n) S.ap[lane] = 0.0;
        wave_sync();
        status = gi_factorize<0>(Sq, n, nullptr COPRA_FINE_PASS);
        if (status == 0) gi_invert<0>(Sq, n);
        if (status == 0) {
            // T = E Jq (nx x n): lane j holds column j;  top-left = R + T T'
            double Tj[16];
            for (int a = 0; a < nx && a < 16; ++a) {
                double acc = 0.0;
                if (lane < n)
                    for (int i = 0; i <= lane; ++i) acc += Eb[a + nx * i] * Jq[i * ldq + lane];
                Tj[a] = acc;
            }
            for (int a = 0; a < nx; ++a)
                for (int b2 = a; b2 < nx; ++b2) {
                    const double sab = wave_sum((lane < n) ? Tj[a] * Tj[b2] : 0.0);
                    if (lane == 0) S.J[a * ld + b2] = P.is_R[a + nx * b2] + sab;
                }
        }
        wave_sync();
    }
    // parity hook: the dense QP of this instance (LMPC.h:112-127 on the InitialStateLMPC object)
    StageRows<0, 0, 0> base { P, G, Xcur, Xcur, nb, RowDesc {}, 0.0, 0.0 };
    base.inst = inst;
    StageRowsIS rows { P, base, Phi, Xi, 0.0, 0.0 };
    {
        const int lv = (lane < nv) ? lane : nv - 1;
        if (lv < nx) {
            rows.ubv = P.x0ub ? P.x0ub[(size_t)inst * nx + lv] : P.x0[(size_t)inst * nx + lv];
            rows.lbv = P.x0lb ? P.x0lb[(size_t)inst * nx + lv] : P.x0[(size_t)inst * nx + lv];
        } else {
            rows.ubv = base.bound_ub(lv - nx);
            rows.lbv = base.bound_lb(lv - nx);
        }
    }
    for (int i = lane; i < P.mgen; i += kWave) nb[i] = sqrt(rows.norm2(base.load_desc(i)));
    if (inst == P.dump_instance && P.dumpQ) {
        if (lane < nv) {
            for (int i = 0; i < nv; ++i)
                P.dumpQ[(size_t)lane * nv + i] = (i <= lane) ? S.J[i * ld + lane] : S.J[lane * ld + i];
            P.dumpc[lane] = S.cvec[lane];
        }
        for (int i = lane; i < P.mgen; i += kWave) {
            const RowDesc d = base.load_desc(i);
            for (int j = 0; j < nv; ++j) P.dumpA[(size_t)j * P.mgen + i] = rows.coeff(d, j);
            P.dumpb[i] = d.f - base.lhs(d, Xi, nullptr); // z = f - E xi_k (constraints.cpp:80)
        }
    }
    wave_sync();
    if (P.dump_only) return;
    int it_main = 0, it_drop = 0;
    if (status == 0) status = gi_factorize<0>(S, nv, nullptr COPRA_FINE_PASS);
    if (status == 0) status = gi_active_set<0>(S, nv, P.meq, P.mgen, rows, P.vsmall, P.max_iter, it_main, it_drop COPRA_FINE_PASS);
    wave_sync();
    // ---- results (InitialStateLMPC.cpp:124-128) ----
    if (status == 0) {
        rows.refresh_trajectory(S.xs);
        wave_sync();
        for (int e = lane; e < n; e += kWave) P.control[(size_t)inst * n + e] = S.xs[nx + e];
        for (int e = lane; e < X; e += kWave) P.trajectory[(size_t)inst * X + e] = Xcur[e];
        for (int e = lane; e < nx; e += kWave) P.x0_opt[(size_t)inst * nx + e] = S.xs[e];
    } else {
        const double qnan = __builtin_nan("");
        for (int e = lane; e < n; e += kWave) P.control[(size_t)inst * n + e] = qnan;
        for (int e = lane; e < X; e += kWave) P.trajectory[(size_t)inst * X + e] = qnan;
        for (int e = lane; e < nx; e += kWave) P.x0_opt[(size_t)inst * nx + e] = qnan;
    }
    if (lane == 0) {
        P.status[inst] = status;
        P.iter[2 * (size_t)inst] = it_main;
        P.iter[2 * (size_t)inst + 1] = it_drop;
    }
}

} // namespace copra_hip
