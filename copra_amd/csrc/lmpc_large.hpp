// lmpc_large.hpp -- "condense + solve" for problems with more than 64 decision variables (up to 512): one MPC
// instance per WORKGROUP, thread = row of J, persistent grid over the batch.  Covers LMPC (src/LMPC.cpp:79-101) and
// InitialStateLMPC (src/InitialStateLMPC.cpp:52-128), e.g. BASELINE config 5 (xDim 12, uDim 6, 50 steps: 312
// variables) and the 300-step fixtures of the reference's tests (tests/systems.h).
//
// Same construction as lmpc_fused.hpp / islmpc_fused.hpp -- Psi stays implicit (first block column G_k = A^k B in
// LDS), the Hessian is built by prefix sums along its block diagonals, constraints stay implicit rows -- but
//   * the n x n matrices (Hessian -> factor -> R, and J) live in the workgroup's HBM workspace (gi_large.hpp);
//   * Phi_k (and M Phi_k, E, E Jq of the InitialStateLMPC variant) live there too: they are touched once per instance
//     or with workgroup-uniform addresses, LDS keeps what every iteration gathers from (G, the trajectory, vectors);
//   * InitialStateLMPC: Q is built twice (once alone for E Q^-1 E' = (E Jq)(E Jq)', once at its place in the
//     (nx + n)^2 Hessian) instead of being copied inside the workspace.
// Full-size constraint rows are evaluated cooperatively by the workgroup (lhs_cooperative); full-size COST entries run
// as rank-4 updates of the Hessian in the workspace (a completeness path: no structure is exploited there).
#pragma once

#include "gi_large.hpp"
#include "lmpc_fused.hpp"

namespace copra_hip {

// Implicit rows at workgroup level; the coefficient algebra is StageRows<0,0,0>'s (thread-level functions only).
struct LargeRows {
    const FusedPlan& P;
    StageRows<0, 0, 0> base;
    const double* Phi; // HBM: (N+1) blocks nx x nx
    const double* Xi; // LDS
    const double* x0ub; // this instance (InitialStateLMPC)
    const double* x0lb;
    int off; // nx for InitialStateLMPC (decision vector [x0; U]), else 0
    RowDesc mine0, mine1; // descriptors of rows tid and tid + T: the scan does not go back to the plan tables for them
    double* fulls; // LDS: left-hand sides of the (few) full-size rows, evaluated by the whole workgroup
    double* red; // LDS scratch of the workgroup reductions

    COPRA_DEV int nx() const { return P.nx; }
    COPRA_DEV int nvar() const { return off + P.n; }

    COPRA_DEV double coeff(const RowDesc& d, int j) const
    {
        if (j >= off) return base.coeff(d, j - off);
        const int a = j, k = d.k, eo = d.eo, nPhi = nx() * nx(); // x0 part: Y_row = E_row Phi_k
        double v = 0.0;
        if (e_onehot(d.ek)) {
            v = e_sign(d.ek) * Phi[k * nPhi + eo + nx() * a];
        } else if (d.ek == kEDense) {
            for (int c = 0; c < nx(); ++c) v += base.params()[eo + c] * Phi[k * nPhi + c + nx() * a];
        } else if (d.ek == kEFull) {
            for (int s = 0; s <= P.N; ++s)
                for (int c = 0; c < nx(); ++c) v += base.params()[eo + s * nx() + c] * Phi[s * nPhi + c + nx() * a];
        }
        return v;
    }
    COPRA_DEV double norm2(const RowDesc& d) const
    {
        if (off == 0) return base.norm2(d);
        double s = 0.0;
        for (int j = 0; j < nvar(); ++j) {
            const double a = coeff(d, j);
            s += a * a;
        }
        return s;
    }
    // X = Phi x0 + xi + Psi U at the current iterate (x0 = the variable for InitialStateLMPC, folded in Xbar otherwise)
    COPRA_DEV void refresh_trajectory(const double* xs) const
    {
        const int nPhi = nx() * nx(), nu = P.nu, T = bt_size();
        for (int row = bt_tid(); row < P.X; row += T) {
            const int k = row / nx(), comp = row - k * nx();
            double a0;
            if (off) {
                a0 = Xi[row];
                for (int a = 0; a < nx(); ++a) a0 += Phi[k * nPhi + comp + nx() * a] * xs[a];
            } else {
                a0 = base.Xbar[row];
            }
            const double* g = base.G + comp + (k - 1) * nx() * nu;
            for (int jb = 0; jb < k; ++jb)
                for (int jc = 0; jc < nu; ++jc) a0 += g[-jb * nx() * nu + nx() * jc] * xs[off + jb * nu + jc];
            base.Xcur[row] = a0;
        }
    }
    COPRA_DEV static bool is_full(const RowDesc& d) { return d.ek == kEFull || d.gk == kGFull; }
    // E_row . X + G_row . U of a full-size row: a fullXDim (+ fullUDim) long inner product, strided over the workgroup
    COPRA_DEV double lhs_cooperative(const RowDesc& d, const double* Xv, const double* u) const
    {
        const int T = bt_size();
        double part = 0.0;
        if (d.ek == kEFull) {
            for (int r = bt_tid(); r < P.X; r += T) part += base.params()[d.eo + r] * Xv[r];
        } else if (bt_tid() == 0) {
            part += base.lhs(RowDesc { d.k, d.ek, d.eo, kGNone, 0, 0.0 }, Xv, nullptr);
        }
        if (u) {
            if (d.gk == kGFull) {
                for (int j = bt_tid(); j < P.n; j += T) part += base.params()[d.go + j] * u[j];
            } else if (d.gk == kGStep && bt_tid() == 0) {
                for (int c = 0; c < P.nu; ++c) part += base.params()[d.go + c] * u[d.k * P.nu + c];
            }
        }
        return block_sum(part, red);
    }
    COPRA_DEV int full_slot(int i) const
    {
        for (int f = 0; f < kMaxFullRows; ++f)
            if (f < P.n_full_rows && P.full_row[f] == i) return f;
        return -1;
    }
    COPRA_DEV void begin_scan(const double* xs) const
    {
        if (P.any_state_rows || off) refresh_trajectory(xs);
        bt_sync();
        for (int f = 0; f < P.n_full_rows; ++f) { // (n_full_rows < 0: too many, they stay thread-level)
            const RowDesc d = base.load_desc(P.full_row[f]);
            const double ax = lhs_cooperative(d, base.Xcur, xs + off);
            if (bt_tid() == 0) fulls[f] = ax;
        }
        if (P.n_full_rows > 0) bt_sync();
    }
    COPRA_DEV void cache_own_rows()
    {
        const RowDesc none { 0, kENone, 0, kGNone, 0, 0.0 };
        const int i0 = bt_tid(), i1 = bt_tid() + bt_size();
        mine0 = (i0 < P.mgen) ? base.load_desc(i0) : none;
        mine1 = (i1 < P.mgen) ? base.load_desc(i1) : none;
    }
    COPRA_DEV double slack_of(const RowDesc& d, int i, const double* xs) const
    {
        double ax;
        if (P.n_full_rows > 0 && is_full(d))
            ax = fulls[full_slot(i)];
        else
            ax = base.lhs(d, base.Xcur, xs + off);
        return (i < P.meq) ? (ax - d.f) : (d.f - ax);
    }
    COPRA_DEV double slack(int i, const double* xs) const { return slack_of(base.load_desc(i), i, xs); }
    COPRA_DEV double slack_at(int pass, int i, const double* xs) const
    {
        if (pass == 0) return slack_of(mine0, i, xs);
        if (pass == 1) return slack_of(mine1, i, xs);
        return slack(i, xs);
    }
    COPRA_DEV double slack_uniform(int p, const double* xs) const // cooperative (the trajectory is current)
    {
        const int i = uniform_i32(p);
        const RowDesc d = base.load_desc(i);
        if (!is_full(d)) return slack_of(d, i, xs);
        const double ax = lhs_cooperative(d, base.Xcur, xs + off);
        return (i < P.meq) ? (ax - d.f) : (d.f - ax);
    }
    COPRA_DEV double norm(int i) const { return base.nb[i]; }
    // a row at step k involves x0 and u_0 .. u_k only (Psi is block lower triangular): its normal ends there
    COPRA_DEV int normal_extent(int p) const
    {
        const int i = uniform_i32(p);
        if (P.row_ekind[i] == kEFull || P.row_gkind[i] == kGFull) return nvar();
        const int k = P.row_step[i];
        const int blocks = (k + 1 < P.N) ? k + 1 : P.N;
        return off + blocks * P.nu;
    }
    COPRA_DEV double ub(int j) const { return (j < off) ? x0ub[j] : base.bound_ub(j - off); }
    COPRA_DEV double lb(int j) const { return (j < off) ? x0lb[j] : base.bound_lb(j - off); }
    COPRA_DEV void load_normal(int p, double sgn, double* np) const
    {
        const int j = bt_tid();
        if (j >= nvar()) return;
        const RowDesc d = base.load_desc(uniform_i32(p));
        const double v = coeff(d, j);
        np[j] = (p < P.meq) ? sgn * v : -v;
    }
};

// Hessian of the U block (lower triangle, placed at rows / columns qoff.. of F) and, when `linear`, the linear terms:
//   LMPC:              cj += c_j                                   (costFunctions.cpp:78-80, 106, 152-155, 211-213)
//   InitialStateLMPC:  cj += f_j and Ecol[a] += E(a, j)            (InitialStateLMPC.cpp:82-83)
// thread j < n owns column j = (block, component) of the linear terms and walks block diagonal `block` of Q.
COPRA_DEV void large_costs(const FusedPlan& P, int inst, double* lds, double* F, int ld, int qoff, bool linear, const double* G,
    const double* Xbar, const double* Xi, const double* Phi, double* MPhi, double& cj, double* Ecol)
{
    const LargeLayout& L = P.large;
    const int tid = bt_tid(), T = bt_size();
    const int nx = P.nx, nu = P.nu, N = P.N, n = P.n;
    const bool is = P.initial_state != 0;
    const int nPhi = nx * nx;
    double* Y = lds + L.Y;
    double* We = lds + L.We;
    double* Cp = lds + L.Cp;
    const int blk = tid / nu, sub = tid - blk * nu;
    double* Q = F + (size_t)qoff * ld + qoff;
    if (tid < n) {
        double one = 1.0;
        one *= 1e-6; // Q_.setIdentity(); Q_ *= 1e-6;  (LMPC.cpp:228-229)
        Q[(size_t)tid * ld + tid] = one;
    }
    for (int t = 0; t < P.ncost; ++t) {
        const CostTerm& ct = P.cost[t];
        const int r = ct.rows;
        bt_sync();
        if (ct.full) {
            // Full-size entry (costFunctions.cpp:65-71, 141-146, 197-203): tmp = M Psi (+ N), R x n and dense.
            //   Q += tmp' W tmp,  c += (M xbar - p)' W tmp   |   InitialStateLMPC: E += (M Phi)' W tmp, f += (M xi - p)' W tmp
            // Four rows of tmp per pass over the lower triangle of Q (thread j owns row j of Q: coalesced RMW).
            // Completeness path: R n^2 / 8 read-modify-writes per cost, no structure exploited.
            const int X = P.X;
            const double* Mr = (ct.offM >= 0) ? P.params + ct.offM : nullptr; // R x X, row-major
            const double* Nr = (ct.offN >= 0) ? P.params + ct.offN : nullptr; // R x n, row-major
            const double* pp = cost_reference(P, t, inst);
            const double* ww = P.params + ct.offW;
            double* rowbuf = lds + L.sol.stage; // 4 x n
            double* mphi = lds + L.sol.xs; // 4 x (nx + 1): (M Phi)(row, :) and the residual of the row
            const double* xfree = is ? Xi : Xbar;
            const int jb = tid / nu, jc = tid - jb * nu;
            for (int r0 = 0; r0 < r; r0 += 4) {
                bt_sync();
                double tv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rr = r0 + u;
                    double v = 0.0;
                    if (rr < r && tid < n) {
                        if (Mr) { // row rr of M times column tid of Psi: Psi_{s, jb} = G_{s-1-jb} for s > jb
                            const double* mrow = Mr + (size_t)rr * X;
                            for (int s2 = jb + 1; s2 <= N; ++s2) {
                                const double* Gk = G + (s2 - 1 - jb) * nx * nu + nx * jc;
                                for (int c = 0; c < nx; ++c) v += mrow[s2 * nx + c] * Gk[c];
                            }
                        }
                        if (Nr) v += Nr[(size_t)rr * n + tid];
                    }
                    tv[u] = v;
                    if (tid < n) rowbuf[u * n + tid] = v;
                }
                if (linear) { // thread (u, a): a < nx -> (M Phi)(rr, a) (InitialStateLMPC), a == nx -> residual of row rr
                    for (int e = tid; e < 4 * (nx + 1); e += T) {
                        const int u = e / (nx + 1), a = e - u * (nx + 1);
                        const int rr = r0 + u;
                        double acc = 0.0;
                        if (rr < r && Mr) {
                            const double* mrow = Mr + (size_t)rr * X;
                            if (a < nx) {
                                if (is)
                                    for (int s2 = 0; s2 <= N; ++s2)
                                        for (int c = 0; c < nx; ++c) acc += mrow[s2 * nx + c] * Phi[s2 * nPhi + c + nx * a];
                            } else {
                                for (int col = 0; col < X; ++col) acc += mrow[col] * xfree[col];
                            }
                        }
                        mphi[e] = (a < nx) ? acc : ((rr < r) ? acc - pp[rr] : 0.0);
                    }
                }
                bt_sync();
                if (tid < n) {
                    double tw[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) tw[u] = (r0 + u < r) ? tv[u] * ww[r0 + u] : 0.0;
                    for (int i = 0; i <= tid; ++i) { // Q(tid, i), i <= tid: lower triangle, column i
                        double acc = 0.0;
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc += rowbuf[u * n + i] * tw[u];
                        Q[(size_t)i * ld + tid] += acc;
                    }
                    if (linear) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            cj += mphi[u * (nx + 1) + nx] * tw[u];
                            if (is)
                                for (int a = 0; a < nx; ++a) Ecol[a] += mphi[u * (nx + 1) + a] * tw[u];
                        }
                    }
                }
            }
            continue;
        }
        double* Mx = Cp;
        double* Nm = Cp + r * nx;
        double* p = Nm + r * nu;
        double* w = p + r;
        for (int e = tid; e < r * nx; e += T) Mx[e] = (ct.offM >= 0) ? P.params[ct.offM + e] : 0.0;
        for (int e = tid; e < r * nu; e += T) Nm[e] = (ct.offN >= 0) ? P.params[ct.offN + e] : 0.0;
        for (int e = tid; e < r; e += T) {
            p[e] = cost_reference(P, t, inst)[e];
            w[e] = P.params[ct.offW + e];
        }
        bt_sync();
        if (ct.kind == kCostControl) { // ControlCost::update (costFunctions.cpp:148-156): block-diagonal N'WN
            if (tid < n) {
                for (int i2 = 0; i2 < nu; ++i2) {
                    double acc = 0.0;
                    for (int k = 0; k < r; ++k) acc += (Nm[k + r * i2] * w[k]) * Nm[k + r * sub];
                    Q[(size_t)(blk * nu + i2) * ld + tid] += acc;
                }
                if (linear) {
                    double acc = 0.0;
                    for (int k = 0; k < r; ++k) acc += ((-p[k]) * w[k]) * Nm[k + r * sub];
                    cj += acc;
                }
            }
            continue;
        }
        const bool mixed = (ct.kind == kCostMixed);
        for (int e = tid; e < N * r * nu; e += T) { // Y_k = M G_k
            const int k = e / (r * nu), rem = e - k * r * nu;
            const int jc = rem / r, row = rem - jc * r;
            const double* Gk = G + k * nx * nu + nx * jc;
            double acc = 0.0;
            for (int c = 0; c < nx; ++c) acc += Mx[row + r * c] * Gk[c];
            Y[e] = acc;
        }
        if (linear) {
            const double* xfree = is ? Xi : Xbar; // We_k = w .* (M xbar_k - p), InitialStateLMPC: w .* (M xi_k - p)
            for (int e = tid; e < (N + 1) * r; e += T) {
                const int k = e / r, row = e - k * r;
                double acc = 0.0;
                for (int c = 0; c < nx; ++c) acc += Mx[row + r * c] * xfree[k * nx + c];
                We[e] = (acc - p[row]) * w[row];
            }
            if (is) {
                for (int e = tid; e < (N + 1) * r * nx; e += T) { // MPhi_k = M Phi_k (costFunctions.cpp:77)
                    const int k = e / (r * nx), rem = e - k * r * nx;
                    const int a = rem / r, row = rem - a * r;
                    double acc = 0.0;
                    for (int c = 0; c < nx; ++c) acc += Mx[row + r * c] * Phi[k * nPhi + c + nx * a];
                    MPhi[e] = acc;
                }
            }
        }
        bt_sync();
        const int K = mixed ? N - 1 : N; // last state index that enters the sum (target: only K = N)
        const bool accumulate = (ct.kind != kCostTarget);
        if (tid < n) {
            const int delta = blk, ic = sub;
            double val[kMaxNu], cross[kMaxNu];
#pragma unroll
            for (int jc = 0; jc < kMaxNu; ++jc) val[jc] = cross[jc] = 0.0;
            if (mixed) { // step k = b of MixedCost (costFunctions.cpp:207)
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
#pragma unroll
                for (int jc = 0; jc < kMaxNu; ++jc) {
                    if (jc < nu) {
                        double pterm = 0.0;
                        if (m >= 0) {
                            const double* Ya = Y + (m + delta) * r * nu + r * ic;
                            const double* Yb = Y + m * r * nu + r * jc;
                            for (int k = 0; k < r; ++k) pterm += (Ya[k] * w[k]) * Yb[k];
                        }
                        val[jc] = accumulate ? val[jc] + pterm : pterm;
                        Q[(size_t)(a * nu + ic) * ld + b * nu + jc] += mixed ? val[jc] + cross[jc] : val[jc];
                    }
                }
            }
            if (linear) {
                // column j = (b, jc): steps in ascending order; tmp_k(:, j) = N (k == b, mixed) or Y_{k-1-b}
                const int b = blk, jc = sub;
                const int k_lo = (ct.kind == kCostTarget) ? N : (mixed ? b : b + 1);
                for (int k = k_lo; k <= K; ++k) {
                    const double* tk = (k == b) ? (Nm + r * jc) : (Y + (k - 1 - b) * r * nu + r * jc);
                    double sf = 0.0;
                    for (int q = 0; q < r; ++q) sf += We[k * r + q] * tk[q];
                    cj += sf;
                    if (is) {
                        for (int a = 0; a < nx; ++a) {
                            double se = 0.0;
                            for (int q = 0; q < r; ++q) se += (MPhi[k * r * nx + q + r * a] * w[q]) * tk[q];
                            Ecol[a] += se;
                        }
                    }
                }
            }
        }
    }
    bt_sync();
    if (P.denseQ >= 0 && tid < n) { // host-evaluated user cost functions (COPRA_COST_DENSE), lower triangle: row tid
        const double* Qd = P.params + P.denseQ;
        for (int c = 0; c <= tid; ++c) Q[(size_t)c * ld + tid] += Qd[(size_t)n * c + tid];
        if (linear) {
            cj += is ? P.params[P.densef + tid] : P.params[P.densec + tid]; // InitialStateLMPC.cpp:84 / LMPC.cpp:254
            if (is)
                for (int a = 0; a < nx; ++a) Ecol[a] += P.params[P.denseE + a + nx * tid];
        }
    }
    bt_sync();
}

COPRA_DEV void lmpc_large_body(const FusedPlan& P)
{
    double* lds = lds_base();
    const LargeLayout& L = P.large;
    // the parameter blob (E, G of every constraint) is read in every scan: the row algebra reads an LDS copy of it
    const double* prm = nullptr;
    if (L.nparams > 0) {
        double* pl = lds + L.Params;
        for (int e = bt_tid(); e < L.nparams; e += bt_size()) pl[e] = P.params[e];
        prm = pl;
        bt_sync();
    }
    const int tid = bt_tid(), T = bt_size();
    const int nx = P.nx, nu = P.nu, N = P.N, n = P.n, X = P.X;
    const bool is = P.initial_state != 0;
    const int off = is ? nx : 0, nv = n + off;
    const int ld = L.ld;
    double* ws = P.ws + (size_t)instance_id() * L.ws_total;
    double* F = ws + L.wsF;
    double* Jg = ws + L.wsJ;
    double* Phi = ws + L.wsPhi;
    double* MPhi = ws + L.wsMPhi;
    double* Eg = ws + L.wsE;
    double* Tg = ws + L.wsT;
    double* A = lds + L.A;
    double* B = lds + L.B;
    double* D = lds + L.D;
    double* X0 = lds + L.X0;
    double* G = lds + L.G;
    double* Xi = lds + L.Xi;
    double* Xbar = lds + L.Xbar;
    double* Xcur = lds + L.Xcur;
    double* PP = lds + L.PhiPP;
    double* TL = lds + L.TL;
    LargeSolver S = carve_large(lds, L.sol, nv, Jg, F);
    const int nPhi = nx * nx, nG = nx * nu;

    // (second tier of the Riccati interior-point path: the instances it queued, P.from_list)
    const int work = P.from_list ? *P.ovf_count : P.batch;
    for (int witem = P.inst_offset + instance_id(); witem < work; witem += instance_stride()) {
        const int inst = P.from_list ? P.ovf_list[witem] : witem;
        long long stamp[8];
        stamp[0] = cycle_counter();
#ifdef COPRA_FINE_PROFILE
        const long long wall0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
#ifdef COPRA_FINE_PROFILE
        S.fine = P.prof_fine ? P.prof_fine + 32 * (size_t)inst : nullptr;
#endif
        // ---- 0. this instance's system ----
        for (int e = tid; e < nPhi; e += T) A[e] = P.A[(size_t)inst * nPhi + e];
        for (int e = tid; e < nG; e += T) B[e] = P.B[(size_t)inst * nG + e];
        for (int e = tid; e < nx; e += T) {
            D[e] = P.d[(size_t)inst * nx + e];
            X0[e] = P.x0[(size_t)inst * nx + e];
        }
        bt_sync();
        // ---- 1. preview recursion (PreviewSystem.cpp:57-74): G_s = A G_{s-1}, Phi_s = A Phi_{s-1}, xi_s = A xi_{s-1} + d
        for (int e = tid; e < nPhi; e += T) {
            const double v = (e % nx == e / nx) ? 1.0 : 0.0; // Phi_0 = I (:51)
            PP[e] = v;
            Phi[e] = v;
        }
        for (int e = tid; e < nG; e += T) G[e] = B[e]; // Psi_{1,0} = B (:60)
        for (int e = tid; e < nx; e += T) {
            Xi[e] = 0.0;
            if (!is) Xbar[e] = X0[e];
        }
        bt_sync();
        const int per_step = nx * (nx + nu + 1);
        for (int s = 1; s <= N; ++s) {
            const double* Pprev = PP + ((s - 1) & 1) * nPhi;
            double* Pcur = PP + (s & 1) * nPhi;
            for (int e = tid; e < per_step; e += T) {
                const int c = e / nx, r = e - c * nx;
                if (c < nx) { // Phi_s (:59, :64)
                    double acc = 0.0;
                    for (int t = 0; t < nx; ++t) acc += A[r + nx * t] * Pprev[c * nx + t];
                    Pcur[c * nx + r] = acc;
                    Phi[(size_t)s * nPhi + c * nx + r] = acc;
                } else if (c < nx + nu) { // G_s = A G_{s-1} (:65); G_{N} is not needed
                    if (s < N) {
                        const double* src = G + (s - 1) * nG + (c - nx) * nx;
                        double acc = 0.0;
                        for (int t = 0; t < nx; ++t) acc += A[r + nx * t] * src[t];
                        G[s * nG + (c - nx) * nx + r] = acc;
                    }
                } else { // xi_s = A xi_{s-1} + d (:61, :70)
                    const double* src = Xi + (s - 1) * nx;
                    double acc = 0.0;
                    for (int t = 0; t < nx; ++t) acc += A[r + nx * t] * src[t];
                    Xi[s * nx + r] = acc + D[r];
                }
            }
            bt_sync();
            if (!is) {
                for (int r = tid; r < nx; r += T) { // free response  xbar_s = Phi_s x0 + xi_s
                    double acc = 0.0;
                    for (int c = 0; c < nx; ++c) acc += Pcur[r + nx * c] * X0[c];
                    Xbar[s * nx + r] = acc + Xi[s * nx + r];
                }
            }
        }
        bt_sync();
        stamp[1] = cycle_counter();
        // ---- 2. Hessian and linear term (LMPC.cpp:228-230, 252-255; InitialStateLMPC.cpp:77-122) ----
        if (tid < ld)
            for (int c = 0; c < nv; ++c) F[(size_t)c * ld + tid] = 0.0;
        bt_sync();
        double cj = 0.0;
        double Ecol[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) Ecol[a] = 0.0;
        large_costs(P, inst, lds, F, ld, 0, true, G, Xbar, Xi, Phi, MPhi, cj, Ecol);
        int status = 0;
        if (is) {
            // E -> HBM; Q = Lq Lq', Jq = Lq^-T; T = E Jq; top-left = R + T T'
            if (tid < n)
                for (int a = 0; a < nx; ++a) Eg[(size_t)a * n + tid] = Ecol[a];
            bt_sync();
            LargeSolver Sq = S;
            Sq.n = n;
            status = gl_factorize(Sq);
            if (status == 0) {
                gl_invert(Sq);
                for (int a = 0; a < nx; ++a) {
                    if (tid < n) S.np[tid] = Eg[(size_t)a * n + tid];
                    bt_sync();
                    gl_matvec_t(Sq, Jg, S.np, Tg + (size_t)a * n, n);
                    bt_sync();
                }
                double Tj[16];
#pragma unroll
                for (int a = 0; a < 16; ++a) Tj[a] = (a < nx && tid < n) ? Tg[(size_t)a * n + tid] : 0.0;
                for (int a = 0; a < nx; ++a)
                    for (int b2 = a; b2 < nx; ++b2) {
                        const double sab = block_sum(Tj[a] * Tj[b2], S.red);
                        if (tid == 0) TL[a + nx * b2] = P.is_R[a + nx * b2] + sab;
                    }
            }
            bt_sync();
            // the (nx + n)^2 Hessian: Q again at its place, E' below the top-left block
            if (tid < ld)
                for (int c = 0; c < nv; ++c) F[(size_t)c * ld + tid] = 0.0;
            bt_sync();
            if (tid < n)
                for (int a = 0; a < nx; ++a) F[(size_t)a * ld + nx + tid] = Eg[(size_t)a * n + tid]; // row nx + j, column a
            if (tid < nx) // (TL shares LDS with the cost tables: consume it before they are rebuilt)
                for (int b2 = tid; b2 < nx; ++b2) F[(size_t)tid * ld + b2] = TL[tid + nx * b2]; // row b2 >= column tid
            double dummy = 0.0;
            large_costs(P, inst, lds, F, ld, nx, false, G, Xbar, Xi, Phi, MPhi, dummy, Ecol);
            if (tid < nx) S.cv[tid] = P.is_r[tid];
            if (tid < n) S.cv[nx + tid] = cj;
        } else {
            if (tid < n) S.cv[tid] = cj;
        }
        bt_sync();
        stamp[2] = cycle_counter();
        // ---- 3. implicit rows: norms; parity hook ----
        StageRows<0, 0, 0> base { P, G, Xbar, Xcur, S.nb, RowDesc {}, 0.0, 0.0, prm, inst };
        LargeRows rows { P, base, Phi, Xi,
            is ? (P.x0ub ? P.x0ub + (size_t)inst * nx : P.x0 + (size_t)inst * nx) : nullptr,
            is ? (P.x0lb ? P.x0lb + (size_t)inst * nx : P.x0 + (size_t)inst * nx) : nullptr, off, RowDesc {}, RowDesc {},
            lds + L.FullS, S.red };
        rows.cache_own_rows();
        for (int i = tid; i < P.mgen; i += T) {
            const RowDesc d = base.load_desc(i);
            if (P.n_full_rows > 0 && LargeRows::is_full(d)) continue; // below, by the whole workgroup
            S.nb[i] = sqrt(rows.norm2(d));
        }
        for (int f = 0; f < P.n_full_rows; ++f) { // a full-size row: thread j squares its own coefficient
            const RowDesc d = base.load_desc(P.full_row[f]);
            const double a = (tid < nv) ? rows.coeff(d, tid) : 0.0;
            const double s2 = block_sum(a * a, S.red);
            if (tid == 0) S.nb[P.full_row[f]] = sqrt(s2);
        }
        if (inst == P.dump_instance && P.dumpQ) { // LMPC::Q() c() Aeq() ... (LMPC.h:112-127)
            if (tid < nv) {
                for (int i = 0; i < nv; ++i)
                    P.dumpQ[(size_t)tid * nv + i] = (i >= tid) ? F[(size_t)tid * ld + i] : F[(size_t)i * ld + tid];
                P.dumpc[tid] = S.cv[tid];
            }
            for (int i = tid; i < P.mgen; i += T) {
                const RowDesc d = base.load_desc(i);
                for (int j = 0; j < nv; ++j) P.dumpA[(size_t)j * P.mgen + i] = rows.coeff(d, j);
                P.dumpb[i] = d.f - base.lhs(d, is ? Xi : Xbar, nullptr); // b = z - Y x0 (constraints.cpp:80-81)
            }
        }
        bt_sync();
        if (P.dump_only) return; // copra_batch_dump_qp: one workgroup, one instance
        // ---- 4. + 5. solve ----
        stamp[3] = cycle_counter();
        int it_main = 0, it_drop = 0;
        if (status == 0) status = gl_factorize(S);
        stamp[4] = cycle_counter();
        stamp[5] = stamp[4];
        if (status == 0) {
            gl_invert(S);
            gl_unconstrained(S);
            stamp[5] = cycle_counter();
            status = gl_active_set(S, P.meq, P.mgen, rows, P.vsmall, P.max_iter, it_main, it_drop);
        }
        bt_sync();
        stamp[6] = cycle_counter();
        // ---- 6. results (LMPC.cpp:95-97, 282-286; InitialStateLMPC.cpp:124-128) ----
        if (status == 0) {
            rows.refresh_trajectory(S.xs);
            bt_sync();
            for (int e = tid; e < n; e += T) P.control[(size_t)inst * n + e] = S.xs[off + e];
            for (int e = tid; e < X; e += T) P.trajectory[(size_t)inst * X + e] = Xcur[e];
            if (is)
                for (int e = tid; e < nx; e += T) P.x0_opt[(size_t)inst * nx + e] = S.xs[e];
        } else {
            const double qnan = __builtin_nan("");
            for (int e = tid; e < n; e += T) P.control[(size_t)inst * n + e] = qnan;
            for (int e = tid; e < X; e += T) P.trajectory[(size_t)inst * X + e] = qnan;
            if (is)
                for (int e = tid; e < nx; e += T) P.x0_opt[(size_t)inst * nx + e] = qnan;
        }
        if (tid == 0) {
            P.status[inst] = status;
            P.iter[2 * (size_t)inst] = it_main;
            P.iter[2 * (size_t)inst + 1] = it_drop;
#ifdef COPRA_FINE_PROFILE
            if (P.prof_fine) { // where and when did this instance run (co-residency studies)
                long long* pf = P.prof_fine + 32 * (size_t)inst;
                pf[28] = (long long)__builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
                pf[29] = wall0;
                pf[30] = (long long)__builtin_amdgcn_s_memrealtime();
                pf[31] = (long long)instance_id();
            }
#endif
            if (P.prof) { // preview, costs, norms, cholesky, inverse + x0, active set, results, total
                stamp[7] = cycle_counter();
                long long* pr = P.prof + 8 * (size_t)inst;
                for (int k = 0; k < 7; ++k) pr[k] = stamp[k + 1] - stamp[k];
                pr[7] = stamp[7] - stamp[0];
            }
        }
        bt_sync();
    }
}

} // namespace copra_hip
