// lmpc_fused.hpp -- body of the fused "condense + solve" kernel: one MPC instance per 64-lane wavefront.
//
// What one wave does for its instance (reference call stack: LMPC::solve, src/LMPC.cpp:79-101):
//   0. coalesced load of A, B, d, x0 (66 doubles at (6,3)) into LDS;
//   1. PreviewSystem::updateSystem (src/PreviewSystem.cpp:57-74) WITHOUT the O(N^2) Toeplitz fill: only the first
//      block column G_k = A^k B of Psi (Psi_{i,j} = G_{i-1-j}), Phi_k = A^k and xi_k are formed, by the same
//      left-multiplication recursion as the reference; the three recursions are ONE stacked product
//      [Phi_s | G_{s-1} | xi_s] = A [Phi_{s-1} | G_{s-2} | xi_{s-1}] (+ d on the last column), one element per lane;
//   2. cost functions (src/costFunctions.cpp:63-215) summed as LMPC::makeQPForm does (src/LMPC.cpp:228-230,
//      252-255).  The Hessian block (a,b), a <= b, of a per-step cost is the prefix sum
//          Q_{a,b} = sum_{s=0}^{K-1-b} (M G_{s+b-a})' W (M G_s)
//      along block diagonal delta = b-a, so lane (delta, ic) walks its diagonal once: the per-step terms of the
//      reference's loop are added in the same (ascending step) order but its "lot of sums of zero"
//      (costFunctions.cpp:73) is skipped -- 11 k MACs instead of 453 k at (6,3,20);
//   3. constraints (src/constraints.cpp) stay implicit rows (plan.hpp); only their norms are computed;
//   4. QuadProgDenseSolver::SI_solve (src/QuadProgSolver.cpp:54-72): bounds become the 2n implicit rows [I; -I];
//      Goldfarb-Idnani in gi_core.hpp;
//   5. LMPC::updateResults (src/LMPC.cpp:282-286): control = U, trajectory = Phi x0 + Psi U + xi.
//
// Template parameters fix (xDim, uDim, nrUStep) at compile time (0 = take them from the plan at run time): the
// launcher instantiates the BASELINE shapes and falls back to the generic instantiation for everything else.
#pragma once

#include "gi_core.hpp"

#ifndef COPRA_LATE_ROW_CACHE
#define COPRA_LATE_ROW_CACHE 1
#endif
#ifndef COPRA_STATE_GROUP
#define COPRA_STATE_GROUP 5
#endif
#ifndef COPRA_CHAIN_GROUP
#define COPRA_CHAIN_GROUP 2
#endif
#ifndef COPRA_GRAD_UNROLL
#define COPRA_GRAD_UNROLL 4
#endif

namespace copra_hip {

// reference vector p of cost term t for this instance: the controller-wide one from the parameter blob, or the
// per-instance one handed over by copra_batch_set_cost_reference (each instance tracks its own goal)
COPRA_DEV const double* cost_reference(const FusedPlan& P, int t, int inst)
{
    const CostTerm& ct = P.cost[t];
    return P.cost_p[t] ? P.cost_p[t] + (size_t)inst * ct.prows : P.params + ct.offP;
}

// Which instance does workgroup w of a first tier solve?  Without the one-instance-per-lane pass in front (lmpc_lane.hpp): instance w.  Behind
// it: an entry of the list the pass left -- false when there is none for this workgroup.
// Workgroups go to the eight XCDs in turn (each has its own L2), and the gather of the stage records reads 64-byte sectors that eight
// neighbouring instances share: entry (w % 8) per + w / 8 of the list gives every XCD a CONTIGUOUS eighth of it, so that the neighbours
// run on one XCD at about the same time and seven of their eight reads hit its L2 (HBM fetches of the tier: 1.32 GB -> 0.18 GB).
COPRA_DEV bool tier_instance(const FusedPlan& P, int w, int& inst, bool& lane_failed)
{
    inst = P.inst_offset + w;
    lane_failed = false;
    if (!P.lane_from_list) return true;
    if (P.lane_cap > 0) { // (a grid sized from the last solves' lists: entry w, no dealing in eighths -- what lies beyond the grid is the second launch's)
        if (w >= *P.lane_count || w >= P.lane_cap) return false;
        inst = P.lane_list[w];
        lane_failed = inst < 0;
        inst &= 0x7fffffff;
        return true;
    }
    const int cnt = *P.lane_count, per = (cnt + 7) >> 3;
    const int idx = (w & 7) * per + (w >> 3);
    if ((w >> 3) >= per || idx >= cnt) return false;
    inst = P.lane_list[idx];
    lane_failed = inst < 0; // (top bit: the pass's factorisation failed -- status 2)
    inst &= 0x7fffffff;
    return true;
}

// One constraint row of the plan (see plan.hpp), held in registers.
struct RowDesc {
    int k, ek, eo, gk, go;
    double f;
};

// Implicit constraint rows of an LMPC problem.
template <int NX_, int NU_, int NH_>
struct StageRows {
    const FusedPlan& P;
    const double* G; // LDS
    const double* Xbar; // LDS
    double* Xcur; // LDS
    const double* nb; // LDS
    RowDesc mine; // descriptor of row `lane` (rows 0..63 are scanned as i == lane): no table look-ups in the loop
    double ub_mine, lb_mine; // XU_lane, XL_lane
    const double* prm = nullptr; // optional LDS copy of the parameter blob (workgroup-per-instance kernel)
    int inst = 0; // instance this wave / workgroup works on (per-instance right-hand sides and bounds)
    const double* zero = nullptr; // an LDS double that holds 0.0 (compile-time shapes: switched-off terms read it)
    // Riccati-factor tier: the state trajectory AT THE UNCONSTRAINED MINIMISER (a by-product of its roll-out), nx (N + 1)
    // doubles.  The first scan of the active-set iteration happens at exactly that iterate, so a one-hot state row reads its
    // left-hand side there instead of summing over the blocks G (3 k of the 6.5 k cycles of a first scan).
    double* xu = nullptr;
    mutable int scans = 0; // begin_scan() calls so far
    // ... and with `xi` set the tier MAINTAINS that trajectory: z = R^-1 v is a closed-loop recursion whose states ARE the
    // response Psi z of the trajectory to the step direction (ric_factor.hpp writes them to xi), so a step x += t z is
    // followed by xu += t xi (moved()) -- every scan and the results read the trajectory, nothing sums over G any more.
    const double* xi = nullptr;
    COPRA_DEV void moved(double t) const
    {
        if (!xi) return;
        for (int e = lane_id(); e < xdim(); e += kWave) xu[e] += t * xi[e];
    }

    COPRA_DEV const double* params() const { return prm ? prm : P.params; }

    COPRA_DEV int nx() const { return NX_ ? NX_ : P.nx; }
    COPRA_DEV int nu() const { return NU_ ? NU_ : P.nu; }
    COPRA_DEV int nh() const { return NH_ ? NH_ : P.N; }
    COPRA_DEV int nvar() const { return (NU_ && NH_) ? NU_ * NH_ : P.n; }
    COPRA_DEV int xdim() const { return (NX_ && NH_) ? NX_ * (NH_ + 1) : P.X; }

    COPRA_DEV double bound_ub(int j) const { return P.ub_inst ? P.ub_inst[(size_t)inst * nvar() + j] : P.ub[j]; }
    COPRA_DEV double bound_lb(int j) const { return P.lb_inst ? P.lb_inst[(size_t)inst * nvar() + j] : P.lb[j]; }
    COPRA_DEV RowDesc load_desc(int i) const
    {
        RowDesc d;
        d.k = P.row_step[i];
        d.ek = P.row_ekind[i];
        d.eo = P.row_eoff[i];
        d.gk = P.row_gkind[i];
        d.go = P.row_goff[i];
        d.f = P.row_f_inst ? P.row_f_inst[(size_t)inst * P.mgen + i] : P.row_f[i];
        return d;
    }
    COPRA_DEV void cache_own_row()
    {
        const int i = lane_id();
        if (i < P.mgen)
            mine = load_desc(i);
        else
            mine = RowDesc { 0, kENone, 0, kGNone, 0, 0.0 };
        const int j = (i < nvar()) ? i : nvar() - 1;
        ub_mine = bound_ub(j);
        lb_mine = bound_lb(j);
    }
    COPRA_DEV RowDesc desc(int i) const { return (i < kWave) ? mine : load_desc(i); }
    // descriptor of a wave-uniform row: rows 0..63 sit in the registers of their lane (no trip to the plan tables)
    COPRA_DEV RowDesc desc_uniform(int p) const
    {
        p = uniform_i32(p);
        if (p >= kWave) return load_desc(p);
        RowDesc d;
        d.k = bcast_i32(mine.k, p);
        d.ek = bcast_i32(mine.ek, p);
        d.eo = bcast_i32(mine.eo, p);
        d.gk = bcast_i32(mine.gk, p);
        d.go = bcast_i32(mine.go, p);
        d.f = bcast_f64(mine.f, p);
        return d;
    }

    // coefficient j of a row in the reference's orientation (row of Aeq / Aineq)
    COPRA_DEV double coeff(const RowDesc& d, int j) const
    {
        const int k = d.k, eo = d.eo, go = d.go;
        const int jb = j / nu(), jc = j - jb * nu();
        double v = 0.0;
        if (e_onehot(d.ek)) {
            if (jb < k) v = e_sign(d.ek) * G[(k - 1 - jb) * nx() * nu() + nx() * jc + eo];
        } else if (d.ek == kEDense) {
            if (jb < k) {
                const double* Gk = G + (k - 1 - jb) * nx() * nu() + nx() * jc;
                for (int c = 0; c < nx(); ++c) v += params()[eo + c] * Gk[c];
            }
        } else if (d.ek == kEFull) {
            for (int s = jb + 1; s <= nh(); ++s) {
                const double* Gk = G + (s - 1 - jb) * nx() * nu() + nx() * jc;
                for (int c = 0; c < nx(); ++c) v += params()[eo + s * nx() + c] * Gk[c];
            }
        }
        if (d.gk == kGStep) {
            if (jb == k) v += params()[go + jc];
        } else if (d.gk == kGFull) {
            v += params()[go + j];
        }
        return v;
    }

    // squared norm of a row: sum_j coeff(j)^2
    COPRA_DEV double norm2(const RowDesc& d) const
    {
        if (e_onehot(d.ek) && d.gk == kGNone) { // rows of +-Psi: sum over the blocks G_0 .. G_{k-1}
            double s0 = 0.0;
            const double* g = G + d.eo;
            for (int t = 0; t < d.k; ++t) {
                for (int jc = 0; jc < nu(); ++jc) {
                    const double a = g[t * nx() * nu() + nx() * jc];
                    s0 += a * a;
                }
            }
            return s0;
        }
        double s = 0.0;
        for (int j = 0; j < nvar(); ++j) {
            const double a = coeff(d, j);
            s += a * a;
        }
        return s;
    }

    // E_row . Xv (+ G_row . u when u != nullptr)
    COPRA_DEV double lhs(const RowDesc& d, const double* Xv, const double* u) const
    {
        const int k = d.k, eo = d.eo, go = d.go;
        double ax = 0.0;
        if (e_onehot(d.ek)) {
            ax = e_sign(d.ek) * Xv[k * nx() + eo];
        } else if (d.ek == kEDense) {
            for (int c = 0; c < nx(); ++c) ax += params()[eo + c] * Xv[k * nx() + c];
        } else if (d.ek == kEFull) {
            for (int r = 0; r < xdim(); ++r) ax += params()[eo + r] * Xv[r];
        }
        if (u) {
            if (d.gk == kGStep) {
                for (int c = 0; c < nu(); ++c) ax += params()[go + c] * u[k * nu() + c];
            } else if (d.gk == kGFull) {
                for (int j = 0; j < nvar(); ++j) ax += params()[go + j] * u[j];
            }
        }
        return ax;
    }

    // trajectory at the current iterate: Xcur = Xbar + Psi U  (Psi implicit)
    COPRA_DEV void refresh_trajectory(const double* xs) const
    {
        if constexpr (NH_ > 0 && NU_ > 0 && NX_ > 0) { // (the same sums, unrolled: see state_component)
            for (int row = lane_id(); row < xdim(); row += kWave) {
                const int k = row / NX_, comp = row - k * NX_;
                Xcur[row] = state_component(k, comp, xs);
            }
            return;
        }
        for (int row = lane_id(); row < xdim(); row += kWave) {
            const int k = row / nx(), comp = row - k * nx();
            double a0 = 0.0;
            const double* g = G + comp + (k - 1) * nx() * nu(); // block G_{k-1-jb}
            for (int jb = 0; jb < k; ++jb) {
                for (int jc = 0; jc < nu(); ++jc) a0 += g[-jb * nx() * nu() + nx() * jc] * xs[jb * nu() + jc];
            }
            Xcur[row] = Xbar[row] + a0;
        }
    }

    // component eo of state k at the iterate xs, straight from G (FusedPlan::rows_direct): the same sum in the same
    // order as refresh_trajectory().  With a compile-time horizon every lane runs all steps (the blocks past its own
    // step enter with a zero factor), so the loop unrolls and its LDS reads are issued back to back.
    COPRA_DEV double state_component(int k, int eo, const double* xs) const
    {
        double a0 = 0.0;
        if constexpr (NH_ > 0 && NU_ > 0 && NX_ > 0) {
            const int zoff = (int)(zero - G);
            // groups of COPRA_STATE_GROUP blocks: all operands of a group are read before its first multiply-add (left
            // to itself the compiler waits for every single LDS read -- 60 exposed latencies -- to save registers).
            // The loop over the groups stays rolled: unrolled, all 120 reads are hoisted to the front and ~50 registers
            // spill (705 MB of scratch traffic per launch of 65536 instances, and slower).
            constexpr int GB = COPRA_STATE_GROUP;
#pragma clang loop unroll(disable)
            for (int j0 = 0; j0 < NH_; j0 += GB) {
                double gv[GB][NU_], xv[GB][NU_];
#pragma unroll
                for (int u = 0; u < GB; ++u) {
                    const int jb = j0 + u;
                    if (jb < NH_) {
                        const int t = k - 1 - jb;
                        const bool on = t >= 0;
                        // a block past the lane's own step reads the zero slot: one address select, no value select
                        const int off = on ? eo + t * NX_ * NU_ : zoff; // (an index relative to G: 32 bits)
                        const int st = on ? NX_ : 0;
#pragma unroll
                        for (int jc = 0; jc < NU_; ++jc) {
                            gv[u][jc] = G[off + st * jc];
                            xv[u][jc] = xs[jb * NU_ + jc];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < GB; ++u)
#pragma unroll
                    for (int jc = 0; jc < NU_; ++jc)
                        if (j0 + u < NH_) a0 += gv[u][jc] * xv[u][jc];
            }
        } else {
            const double* g = G + eo + (k - 1) * nx() * nu();
            for (int jb = 0; jb < k; ++jb)
                for (int jc = 0; jc < nu(); ++jc) a0 += g[-jb * nx() * nu() + nx() * jc] * xs[jb * nu() + jc];
        }
        return Xbar[k * nx() + eo] + a0;
    }

    // E_row . X(xs) + G_row . xs at the current iterate
    COPRA_DEV double lhs_now(const RowDesc& d, const double* xs) const
    {
        if (!P.rows_direct) return lhs(d, Xcur, xs);
        double ax = 0.0;
        if (e_onehot(d.ek)) ax = e_sign(d.ek) * ((xu && (scans <= 1 || xi)) ? xu[d.k * nx() + d.eo] : state_component(d.k, d.eo, xs));
        if (d.gk == kGStep) {
            for (int c = 0; c < nu(); ++c) ax += params()[d.go + c] * xs[d.k * nu() + c];
        } else if (d.gk == kGFull) {
            for (int j = 0; j < nvar(); ++j) ax += params()[d.go + j] * xs[j];
        }
        return ax;
    }

    COPRA_DEV void begin_scan(const double* xs) const
    {
        scans += 1;
        if (P.any_state_rows && !P.rows_direct) refresh_trajectory(xs);
        wave_sync();
    }

    // i == base + lane in the scan (rows < 64 come from registers), or a wave-uniform row index otherwise
    COPRA_DEV double slack(int i, const double* xs) const
    {
        const RowDesc d = desc(i);
        const double ax = lhs_now(d, xs);
        return (i < P.meq) ? (ax - d.f) : (d.f - ax);
    }
    COPRA_DEV double slack_uniform(int p, const double* xs) const
    {
        const RowDesc d = desc_uniform(p);
        const double ax = lhs_now(d, xs);
        return (p < P.meq) ? (ax - d.f) : (d.f - ax);
    }

    // (Riccati-factor tier, compact variant: the norm of row `lane` sits in a register, nb holds the rows from 64 on)
    bool nb_split = false;
    double nb_mine = 0.0;
    COPRA_DEV double norm(int i) const { return nb_split ? (i < kWave ? nb_mine : nb[i - kWave]) : nb[i]; }
    COPRA_DEV double ub(int) const { return ub_mine; } // asked for j == min(lane, n-1) only
    COPRA_DEV double lb(int) const { return lb_mine; }

    // Riccati-factor tier: with g_dead the blocks G are gone (lmpc_fused_ric.hpp, compact variant) and the state part of a row --
    // one component of one state -- is handed back as a unit injection (ric_factor.hpp) instead of being spread over ap;
    // without g_dead this is load_normal
    bool g_dead = false;
    COPRA_DEV void load_normal_split(int p, double sgn, double* ap, int& inj_stage, int& inj_comp, double& inj_val) const
    {
        inj_stage = 0;
        if (!g_dead) {
            load_normal(p, sgn, ap);
            return;
        }
        const int j = lane_id();
        const RowDesc d = desc_uniform(p); // (every lane takes part in the broadcast)
        const double sign = (p < P.meq) ? sgn : -1.0;
        if (e_onehot(d.ek) && d.k > 0) {
            inj_stage = d.k;
            inj_comp = d.eo;
            inj_val = sign * e_sign(d.ek);
        }
        if (j >= nvar()) return;
        const int jb = j / nu(), jc = j - jb * nu();
        double v = 0.0;
        if (d.gk == kGStep) {
            if (jb == d.k) v = params()[d.go + jc];
        } else if (d.gk == kGFull) {
            v = params()[d.go + j];
        }
        ap[j] = sign * v;
    }

    COPRA_DEV void load_normal(int p, double sgn, double* ap) const
    {
        const int j = lane_id();
        const RowDesc d = desc_uniform(p); // (every lane takes part in the broadcast)
        if (j >= nvar()) return;
        const double v = coeff(d, j);
        ap[j] = (p < P.meq) ? sgn * v : -v;
    }
};

// ------------------------------------------------------------------------------------------------
// Full-size cost entry (costFunctions.cpp:65-71 TrajectoryCost, :141-146 ControlCost, :197-203 MixedCost):
//     tmp = M Psi (+ N)   (R x n, dense),   Q += tmp' W tmp,   c += tmp' W (M xbar - p).
// This is the dense Psi' W Psi contraction of the reference, and BOTH products run on the matrix cores
// (v_mfma_f64_16x16x4_f64), sixteen rows of tmp at a time:
//   * tmp(16 x 64) = M(16 x X) Psi(X x 64): K = 4 columns of M per instruction, A operand = M straight from HBM / L2 (the
//     matrix is shared by the batch), B operand = Psi built on the fly from the blocks G_k in LDS (Psi_{s,j} = G_{s-1-j}
//     for s > j, else 0); a K-step whose four rows of Psi are zero for a whole 16-column tile is skipped (Psi is block
//     lower triangular: about 45 % of the tile products);
//   * the tile goes to LDS once (16 x 64 doubles) and from there feeds (a) the gradient c_j += sum_r We_r tmp(r, j), lane =
//     column, and (b) the ten upper 16x16 tiles of Q += (tmp' W) tmp, four K-steps of four rows, weights on the A operand
//     as Eigen evaluates it.
// Round 1 formed tmp row by row on the VALU (0.83 M solves/s at the headline shape, the matrix cores 0.7 % busy).
// ------------------------------------------------------------------------------------------------
template <int NX_, int NU_, int NH_, bool TRI_ = false>
COPRA_DEV void full_size_cost_term(const FusedPlan& P, const CostTerm& ct, const double* p, const double* G,
    const double* Xbar, double* Q, int ld, double* scratch, double& cj)
{
    const int lane = lane_id();
    const int nx = NX_ ? NX_ : P.nx, nu = NU_ ? NU_ : P.nu, N = NH_ ? NH_ : P.N;
    const int n = nu * N, X = nx * (N + 1);
    const int R = ct.rows;
    const double* Mr = (ct.offM >= 0) ? P.params + ct.offM : nullptr; // R x X, row-major
    const double* Nr = (ct.offN >= 0) ? P.params + ct.offN : nullptr; // R x n, row-major
    const double* w = P.params + ct.offW;
    double* We = scratch; // R weighted residuals (only where they cannot ride in the product: ControlCost, or n == 64)
    // The residual  M xbar - p  RIDES in the product when the 64-column tile has a spare column (n < 64: the headline has 60):
    // column n of the B operand is xbar, so tmp(r, n) = M_r . xbar comes out of the same instructions.  (Round 3 walked every row of
    // M once more on the vector ALU for it, one lane per row: 2 x 126 dependent trips to the L2 per instance.)
    const bool xcol = Mr != nullptr && n < kWave;
    if (!xcol) { // weighted residuals  We_r = (M_r . xbar - p_r) w_r   (ControlCost: -p_r w_r)
        for (int r = lane; r < R; r += kWave) {
            double acc = 0.0;
            if (Mr)
                for (int col = 0; col < X; ++col) acc += Mr[(size_t)r * X + col] * Xbar[col];
            We[r] = (acc - p[r]) * w[r];
        }
        wave_sync();
    }
    mfma_acc acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t].v[0] = acc[t].v[1] = acc[t].v[2] = acc[t].v[3] = 0.0;
    const int kk = lane >> 4, col = lane & 15;
    // this lane's column of every 16-column tile of Psi: j = 16 tj + col = (jb, jc)
    int jbv[4], goff[4];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
        const int j = 16 * tj + col;
        jbv[tj] = (j < n) ? j / nu : (1 << 20); // (a column beyond n never sees s > jb)
        goff[tj] = (j < n) ? nx * (j - (j / nu) * nu) - (j / nu) * nx * nu : 0; // G[(s-1-jb) nx nu + c + goff]
    }
    const int txc = n >> 4; // the tile that holds column n (the residual column)
    const bool mine_x = xcol && col == (n & 15);
    // K-steps of four columns of M, eight at a time: their A operands -- M(r0 + col, 4 ks + kk), straight from the L2, where the
    // matrix sits for the whole batch -- are ALL requested before the first product (round 3 asked for one, waited, multiplied: 32
    // exposed trips per row block).  K-steps in which these sixteen rows of M are zero are never visited: the plan builder marks the
    // non-zero ones (CostTerm::offMask, one 64-bit word per row block and 64 K-steps) -- a block-diagonal M, which is what AutoSpan
    // produces and the only full-size cost the reference's users write, has 5 of 32.
    constexpr int CH = 8;
    const int KS = (X + 3) >> 2, NW = (KS + 63) >> 6;
    for (int r0 = 0; r0 < R; r0 += 16) {
        mfma_acc tt[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) tt[t].v[0] = tt[t].v[1] = tt[t].v[2] = tt[t].v[3] = 0.0;
        // (the N part of the tile, for the staging below: requested now, used after the products)
        double nrv[4][4];
        if (Nr) {
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int r = r0 + kk + 4 * reg, j = 16 * tj + col;
                    nrv[tj][reg] = (r < R && j < n) ? Nr[(size_t)r * n + j] : 0.0;
                }
        }
        if (Mr) {
            const int ra = r0 + col; // A operand: M(ra, 4 ks + kk)
            const double* mrow = Mr + (size_t)(ra < R ? ra : 0) * X;
            for (int wi = 0; wi < NW; ++wi) {
                unsigned long long m = ~0ull;
                if (ct.offMask >= 0) {
                    const int mo = ct.offMask + 2 * ((r0 >> 4) * NW + wi);
                    m = (unsigned long long)(unsigned)uniform_load(P.params, mo) | ((unsigned long long)(unsigned)uniform_load(P.params, mo + 1) << 32);
                }
                if (KS - 64 * wi < 64) m &= (1ull << (KS - 64 * wi)) - 1ull;
                while (m) { // (wave-uniform: the mask is the same for every lane)
                    int ksv[CH];
#pragma unroll
                    for (int q = 0; q < CH; ++q) {
                        const bool on = m != 0ull;
                        ksv[q] = on ? 64 * wi + (int)__builtin_ctzll(on ? m : 1ull) : -1;
                        m = on ? (m & (m - 1ull)) : 0ull;
                    }
                    double am[CH];
#pragma unroll
                    for (int q = 0; q < CH; ++q) {
                        const int k = 4 * ksv[q] + kk;
                        am[q] = (ksv[q] >= 0 && ra < R && k < X) ? mrow[k] : 0.0;
                    }
#pragma unroll
                    for (int q = 0; q < CH; ++q) {
                        if (ksv[q] < 0) continue;
                        const int k0 = 4 * ksv[q], k = k0 + kk;
                        const int s = k / nx, c = k - s * nx; // row k of Psi = (step s, component c)
                        const int smax = (k0 + 3 < X ? k0 + 3 : X - 1) / nx; // wave-uniform: the last step this K-step touches
                        const double xb = (mine_x && k < X) ? Xbar[k] : 0.0;
#pragma unroll
                        for (int tj = 0; tj < 4; ++tj) {
                            const bool rides = xcol && tj == txc;
                            if (smax <= (16 * tj) / nu && !rides) continue; // the whole tile of Psi is zero (s <= jb for every column)
                            double b = (k < X && s > jbv[tj]) ? G[(s - 1) * nx * nu + c + goff[tj]] : 0.0;
                            if (rides) b = mine_x ? xb : b;
                            mfma_f64_16x16x4(am[q], b, tt[tj]);
                        }
                    }
                }
            }
        }
        // Element `reg` of tile tj in this lane is tmp(r0 + kk + 4 reg, 16 tj + col) -- and the A / B operands of the second product,
        // (tmp' W) tmp over four K-steps of four rows, are tmp(r0 + 4 ks + kk, 16 t + col): THE SAME LANE's element ks.  The tile never
        // leaves the registers (round 3 staged it through 8 KB of LDS per instance: sixteen writes, forty-eight reads and two syncs per
        // row block, and an LDS footprint that kept the kernel at one wave per SIMD).
        if (Nr) {
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) tt[tj].v[reg] += nrv[tj][reg];
        }
        // this lane's four rows: weights, and the weighted residuals We_r = (M_r . xbar - p_r) w_r -- M_r . xbar from the lane of the
        // same row group that holds column n of the tile
        double wv[4], we[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = r0 + kk + 4 * reg, rc = r < R ? r : 0;
            wv[reg] = (r < R) ? w[rc] : 0.0;
            if (xcol) {
                double mx = tt[0].v[reg];
#pragma unroll
                for (int tj = 1; tj < 4; ++tj) mx = (tj == txc) ? tt[tj].v[reg] : mx;
                mx = shfl_f64(mx, (lane & 48) | (n & 15));
                we[reg] = (r < R) ? (mx - p[rc]) * wv[reg] : 0.0;
            } else {
                we[reg] = (r < R) ? We[rc] : 0.0;
            }
        }
        // c += (resid' W) tmp: four rows per lane, then over the four row groups; lane j = 16 tj + col takes column j
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            double part = 0.0;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) part += we[reg] * tt[tj].v[reg];
            part += shfl_xor_f64(part, 16);
            part += shfl_xor_f64(part, 32);
            cj += (kk == tj && lane < n) ? part : 0.0;
        }
        // Q += (tmp' W) tmp : four K-steps of four rows
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            int idx = 0;
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                for (int tj = ti; tj < 4; ++tj) {
                    mfma_f64_16x16x4(tt[ti].v[ks] * wv[ks], tt[tj].v[ks], acc[idx]);
                    ++idx;
                }
        }
    }
    wave_sync();
    int idx = 0;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = ti; tj < 4; ++tj) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int i = 16 * ti + kk + 4 * reg, j = 16 * tj + col;
                if (i < n && j < n && (!TRI_ || i <= j)) Q[fidx<TRI_>(i, j, ld)] += acc[idx].v[reg]; // (factor-only: the packed upper triangle)
            }
            ++idx;
        }
    wave_sync();
}

// RP_ > 0: every cost term is padded to RP_ rows (zero M / N / p / w rows add exact zeros), so the inner products
// over the cost rows unroll; RP_ == 0 uses the run-time row count of each term.
// TRI_: factor-only layout (LdsLayout::tri; gi_core.hpp) -- the Hessian is built straight into the packed upper triangle
// (each entry has ONE writer: the lanes of a diagonal block skip its lower half), plans without full-size costs only.
template <int NX_, int NU_, int NH_, int RP_, bool TRI_ = false, int QR_ = 0>
COPRA_DEV void lmpc_fused_body(const FusedPlan& P, int inst)
{
    double* lds = lds_base();
    const LdsLayout& L = P.lds;
    const int lane = lane_id();
    const int nx = NX_ ? NX_ : P.nx, nu = NU_ ? NU_ : P.nu, N = NH_ ? NH_ : P.N;
    const int n = nu * N, X = nx * (N + 1);
    constexpr int NV = NU_ * NH_;
    double* A = lds + L.A;
    double* B = lds + L.B;
    double* D = lds + L.D;
    double* X0 = lds + L.X0;
    double* G = lds + L.G;
    double* Xbar = lds + L.Xbar;
    double* Xcur = lds + L.Xcur;
    double* nb = lds + L.nb;
    SolverLds S = carve_solver(lds, L);
    const int ld = NV ? (NV | 1) : S.ldj;

    long long stamp[8];
    COPRA_FINE_DECL;
    stamp[0] = cycle_counter();
    // plan look-ups this lane needs much later (its constraint row, its bounds): issue the global loads now so that
    // their latency hides under the preview / cost phases
    StageRows<NX_, NU_, NH_> rows { P, G, Xbar, Xcur, nb, RowDesc {}, 0.0, 0.0 };
    rows.inst = inst;
    rows.zero = S.scal; // (written after the cost phase: its tables borrow the solver vectors in the compact layouts)
    if (!COPRA_LATE_ROW_CACHE) rows.cache_own_row();
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
        // stacked step: element (r, c) of [Phi_s | G_{s-1} | xi_s], c in [0, nx + nu + 1)
        const int per_step = nx * (nx + nu + 1);
        if constexpr (NX_ > 0 && NU_ > 0 && NX_ * (NX_ + NU_ + 1) <= kWave) {
            // one element per lane for the whole recursion: keep row r of A in registers, only 'src' is re-read
            const int e = (lane < per_step) ? lane : 0;
            const int c = e / NX_, r = e - c * NX_;
            double ar[NX_];
#pragma unroll
            for (int t = 0; t < NX_; ++t) ar[t] = A[r + NX_ * t];
            const double add = (c == NX_ + NU_) ? D[r] : 0.0;
            for (int s = 2; s <= N; ++s) {
                wave_sync();
                const double* src = (c < NX_) ? Phi + (s - 1) * nPhi + c * NX_
                                              : (c < NX_ + NU_) ? G + (s - 2) * nG + (c - NX_) * NX_ : Xi + (s - 1) * NX_;
                double* dst = (c < NX_) ? Phi + s * nPhi + c * NX_ + r
                                        : (c < NX_ + NU_) ? G + (s - 1) * nG + (c - NX_) * NX_ + r : Xi + s * NX_ + r;
                double sv[NX_];
#pragma unroll
                for (int t = 0; t < NX_; ++t) sv[t] = src[t];
                double acc = 0.0;
#pragma unroll
                for (int t = 0; t < NX_; ++t) acc += ar[t] * sv[t];
                if (lane < per_step) *dst = acc + add;
            }
        } else
        for (int s = 2; s <= N; ++s) {
            wave_sync();
            for (int e = lane; e < per_step; e += kWave) {
                const int c = e / nx, r = e - c * nx;
                const double* src;
                double* dst;
                double add = 0.0;
                if (c < nx) { // Phi_s = A Phi_{s-1} (:64)
                    src = Phi + (s - 1) * nPhi + c * nx;
                    dst = Phi + s * nPhi + c * nx + r;
                } else if (c < nx + nu) { // Psi_{s,0} = A Psi_{s-1,0}, i.e. G_{s-1} = A G_{s-2} (:65)
                    src = G + (s - 2) * nG + (c - nx) * nx;
                    dst = G + (s - 1) * nG + (c - nx) * nx + r;
                } else { // xi_s = A xi_{s-1} + d (:70)
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
        // free response  Xbar = Phi x0 + xi
        for (int row = lane; row < X; row += kWave) {
            const int k = row / nx, r = row - k * nx;
            const double* Pk = Phi + k * nPhi + r;
            double acc = 0.0;
            for (int c = 0; c < nx; ++c) acc += Pk[nx * c] * X0[c];
            Xbar[row] = acc + Xi[row];
        }
        wave_sync(); // the compact layout lends the J region to the preview tables: done with them before Q is built
    }
    stamp[1] = cycle_counter();
    // ---- 2. Hessian and gradient: Q = 1e-6 I + sum Q_k, c = sum c_k (LMPC.cpp:228-230, 252-255) ----
    {
        double* Q = S.J;
        COPRA_FINE("costs:start");
        // Compile-time shapes whose first cost is a per-step state cost: its block-diagonal walk below writes every
        // entry of the upper triangle, so it stores 1e-6 I + Q_0 outright instead of clearing Q and adding to it.
        const bool fresh0 = RP_ > 0 && NU_ > 0 && NH_ > 0 && P.ncost > 0 && P.cost[0].kind != kCostControl;
        if (!fresh0) { // the J region starts 16-byte aligned (plan_builder.hpp: even offsets): clear two doubles per store
            f64x2* Q2 = reinterpret_cast<f64x2*>(Q);
            const int n2 = TRI_ ? (n * (n + 1) / 2 + 1) / 2 : (n * ld + 1) / 2;
#pragma unroll 8
            for (int e = lane; e < n2; e += kWave) Q2[e] = f64x2 { 0.0, 0.0 };
        }
        wave_sync();
        COPRA_FINE("costs:zeroed");
        if (lane < n && !fresh0) {
            double one = 1.0;
            one *= 1e-6; // Q_.setIdentity(); Q_ *= 1e-6;
            Q[fidx<TRI_>(lane, lane, ld)] = one;
        }
        double cj = 0.0; // lane j accumulates c_j
        double* Ybuf = lds + L.BldY;
        double* We = lds + L.BldWe;
        double* Cp = lds + L.BldCp; // this cost's parameters: M (r x nx) | N (r x nu) | p (r) | w (r)
        const int blk = lane / nu, sub = lane - blk * nu; // lane as (block, component)
        for (int t = 0; t < P.ncost; ++t) {
            const CostTerm& ct = P.cost[t];
            if constexpr (RP_ == 0) { // plans with full-size entries run the instantiations without padded cost rows (plan.hpp)
                if (ct.full) {
                    wave_sync();
                    full_size_cost_term<NX_, NU_, NH_, TRI_>(P, ct, cost_reference(P, t, inst), G, Xbar, Q, ld, lds + L.BldFull, cj);
                    continue;
                }
            }
            const int rc = ct.rows; // rows of this term as given
            const int r = RP_ ? RP_ : rc; // rows it is processed with (zero padded)
            wave_sync();
            double* Mx = Cp;
            double* Nm = Cp + r * nx;
            double* p = Nm + r * nu;
            double* w = p + r;
            for (int e = lane; e < r * nx; e += kWave) {
                const int row = e % r, c = e / r;
                Mx[e] = (ct.offM >= 0 && row < rc) ? P.params[ct.offM + row + rc * c] : 0.0;
            }
            for (int e = lane; e < r * nu; e += kWave) {
                const int row = e % r, c = e / r;
                Nm[e] = (ct.offN >= 0 && row < rc) ? P.params[ct.offN + row + rc * c] : 0.0;
            }
            const double* pref = cost_reference(P, t, inst);
            for (int e = lane; e < r; e += kWave) {
                p[e] = (e < rc) ? pref[e] : 0.0;
                w[e] = (e < rc) ? P.params[ct.offW + e] : 0.0;
            }
            wave_sync();
            COPRA_FINE("cost:params");
            if (ct.kind == kCostControl) {
                // ControlCost::update (costFunctions.cpp:148-156): block-diagonal N'WN, c = -p'WN
                if (lane < n) {
                    for (int i2 = 0; i2 < (TRI_ ? sub + 1 : nu); ++i2) {
                        double acc = 0.0;
                        for (int k = 0; k < r; ++k) acc += (Nm[k + r * i2] * w[k]) * Nm[k + r * sub];
                        Q[fidx<TRI_>(blk * nu + i2, lane, ld)] += acc;
                    }
                    double acc = 0.0;
                    for (int k = 0; k < r; ++k) // (reference trajectory: the reference of this lane's step)
                        acc += ((-(ct.pstride ? (k < rc ? pref[blk * ct.pstride + k] : 0.0) : p[k])) * w[k]) * Nm[k + r * sub];
                    cj += acc;
                }
                continue;
            }
            const bool mixed = (ct.kind == kCostMixed);
            // Y_k = M G_k (the block  M * Psi_{i, j}  with k = i-1-j), We_k = w .* (M xbar_k - p)
            const bool ident = ct.ident && r == nx; // M = I: the blocks G_k themselves (same layout when r == nx)
            const double* Y = ident ? G : Ybuf;
            if (!ident)
            for (int e = lane; e < N * r * nu; e += kWave) {
                const int k = e / (r * nu), rem = e - k * r * nu;
                const int jc = rem / r, row = rem - jc * r;
                const double* Gk = G + k * nx * nu + nx * jc;
                double acc = 0.0;
                for (int c = 0; c < nx; ++c) acc += Mx[row + r * c] * Gk[c];
                Ybuf[e] = acc; // Y[k][row + r*jc]
            }
            for (int e = lane; e < (N + 1) * r; e += kWave) {
                const int k = e / r, row = e - k * r;
                double acc = 0.0;
                for (int c = 0; c < nx; ++c) acc += Mx[row + r * c] * Xbar[k * nx + c];
                We[e] = (acc - (ct.pstride ? ((row < rc && (k + 1) * ct.pstride <= ct.prows) ? pref[k * ct.pstride + row] : 0.0) : p[row])) * w[row]; // (reference trajectory: p_k; MixedCost has none for x_N)
            }
            wave_sync();
            COPRA_FINE("cost:YWe");
            // last state index K that enters the sum: trajectory K = N, mixed K = N-1, target only K = N
            const int K = mixed ? N - 1 : N;
            const bool accumulate = (ct.kind != kCostTarget);
            if (lane < n) {
                // lane = (delta, ic): walk block diagonal delta from the bottom-right corner upwards
                const int delta = blk, ic = sub;
                double val[kMaxNu], cross[kMaxNu];
#pragma unroll
                for (int jc = 0; jc < kMaxNu; ++jc) {
                    val[jc] = 0.0;
                    cross[jc] = 0.0;
                }
                if (mixed) { // step k = b of MixedCost (costFunctions.cpp:207): (M G_{delta-1})' W N  or  N' W N
#pragma unroll
                    for (int jc = 0; jc < kMaxNu; ++jc) {
                        if (jc < nu) {
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
                }
                if constexpr (RP_ > 0 && NU_ > 0 && NH_ > 0) {
                    // Compile-time shape: the walk runs in groups of four steps.  The products P_{m+delta,m} of a group
                    // are formed first (no store in between, so the LDS reads of the group are issued back to back),
                    // then the running sums are stored.  Every lane runs all NH_ steps; the ones past the end of its
                    // diagonal are switched off.  Same additions in the same order as the step-by-step loop below.
                    double wr[RP_];
#pragma unroll
                    for (int k = 0; k < RP_; ++k) wr[k] = w[k];
                    const bool fresh = fresh0 && t == 0;
                    const int last = NH_ - 1 - delta; // this lane's steps: i = N-1-b = 0 .. last
                    const int shift = NH_ - K; // m = i - shift
                    constexpr int CH = COPRA_CHAIN_GROUP;
#pragma unroll
                    for (int i0 = 0; i0 < NH_; i0 += CH) {
                        double pt[CH][NU_], qv[CH][NU_];
#pragma unroll
                        for (int u = 0; u < CH; ++u) {
                            const int i = i0 + u;
                            const int m = i - shift;
                            const bool on = (i <= last) && (m >= 0);
                            const int ma = on ? m + delta : 0, mb = on ? m : 0;
                            double ya[RP_];
#pragma unroll
                            for (int k = 0; k < RP_; ++k) ya[k] = Y[ma * RP_ * NU_ + RP_ * ic + k] * wr[k];
#pragma unroll
                            for (int jc = 0; jc < NU_; ++jc) {
                                double pterm = 0.0;
#pragma unroll
                                for (int k = 0; k < RP_; ++k) pterm += ya[k] * Y[mb * RP_ * NU_ + RP_ * jc + k];
                                pt[u][jc] = on ? pterm : 0.0;
                                const int b = (i <= last) ? NH_ - 1 - i : delta; // (an idle step re-reads the lane's last block)
                                qv[u][jc] = fresh ? ((delta == 0 && jc == ic) ? 1e-6 : 0.0)
                                                  : Q[fidx<TRI_>((b - delta) * NU_ + ic, b * NU_ + jc, ld)];
                            }
                        }
#pragma unroll
                        for (int u = 0; u < CH; ++u) {
                            const int i = i0 + u;
                            if (i <= last) {
                                const int b = NH_ - 1 - i, a = b - delta;
#pragma unroll
                                for (int jc = 0; jc < NU_; ++jc) {
                                    val[jc] = accumulate ? val[jc] + pt[u][jc] : pt[u][jc];
                                    if (!TRI_ || delta > 0 || jc >= ic)
                                        Q[fidx<TRI_>(a * NU_ + ic, b * NU_ + jc, ld)]
                                            = qv[u][jc] + (mixed ? val[jc] + cross[jc] : val[jc]);
                                }
                            }
                        }
                    }
                } else if constexpr (RP_ > 0 && NU_ > 0) {
                    // compile-time rows: every operand of a step is loaded before the first FMA so the LDS reads batch
                    double wr[RP_];
#pragma unroll
                    for (int k = 0; k < RP_; ++k) wr[k] = w[k];
                    for (int b = N - 1; b >= delta; --b) {
                        const int a = b - delta;
                        const int m = K - 1 - b; // P_{m+delta, m}
                        const int ma = (m >= 0) ? m + delta : 0, mb = (m >= 0) ? m : 0;
                        double ya[RP_], yb[NU_][RP_], qv[NU_];
#pragma unroll
                        for (int k = 0; k < RP_; ++k) ya[k] = Y[ma * RP_ * NU_ + RP_ * ic + k] * wr[k];
#pragma unroll
                        for (int jc = 0; jc < NU_; ++jc)
#pragma unroll
                            for (int k = 0; k < RP_; ++k) yb[jc][k] = Y[mb * RP_ * NU_ + RP_ * jc + k];
#pragma unroll
                        for (int jc = 0; jc < NU_; ++jc) qv[jc] = Q[fidx<TRI_>(a * NU_ + ic, b * NU_ + jc, ld)];
#pragma unroll
                        for (int jc = 0; jc < NU_; ++jc) {
                            double pterm = 0.0;
#pragma unroll
                            for (int k = 0; k < RP_; ++k) pterm += ya[k] * yb[jc][k];
                            if (m < 0) pterm = 0.0;
                            val[jc] = accumulate ? val[jc] + pterm : pterm;
                            if (!TRI_ || delta > 0 || jc >= ic)
                                Q[fidx<TRI_>(a * NU_ + ic, b * NU_ + jc, ld)] = qv[jc] + (mixed ? val[jc] + cross[jc] : val[jc]);
                        }
                    }
                } else {
                    for (int b = N - 1; b >= delta; --b) {
                        const int a = b - delta;
                        const int m = K - 1 - b; // P_{m+delta, m}
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
                                if (!TRI_ || delta > 0 || jc >= ic)
                                    Q[fidx<TRI_>(a * nu + ic, b * nu + jc, ld)] += mixed ? val[jc] + cross[jc] : val[jc];
                            }
                        }
                    }
                }
                COPRA_FINE("cost:chain");
                // gradient: c_j = sum_k tmp_k(:,j)' We_k  (ascending step order, costFunctions.cpp:78,80 / :106 / :211)
                const int b = blk, jc = sub;
                double acc = 0.0;
                if (ct.kind == kCostTarget) {
                    const double* Yb = Y + (N - 1 - b) * r * nu + r * jc;
                    for (int k = 0; k < r; ++k) acc += We[N * r + k] * Yb[k];
                } else {
                    if (mixed) { // step k = b of MixedCost: tmp_b(:, j) = N
                        double s0 = 0.0;
                        for (int k = 0; k < r; ++k) s0 += We[b * r + k] * Nm[k + r * jc];
                        acc += s0;
                    }
                    if constexpr (RP_ > 0 && NU_ > 0 && NH_ > 0) {
                        // every lane runs all steps (the ones outside b < s <= K add an exact zero): the loop unrolls
#pragma unroll COPRA_GRAD_UNROLL
                        for (int s = 1; s <= NH_; ++s) {
                            const bool on = (s > b) && (s <= K);
                            const int yi = on ? s - 1 - b : 0;
                            double yv[RP_], wv[RP_];
#pragma unroll
                            for (int k = 0; k < RP_; ++k) {
                                yv[k] = Y[yi * RP_ * NU_ + RP_ * jc + k];
                                wv[k] = We[s * RP_ + k];
                            }
                            double sk = 0.0;
#pragma unroll
                            for (int k = 0; k < RP_; ++k) sk += wv[k] * yv[k];
                            acc += on ? sk : 0.0;
                        }
                    } else if constexpr (RP_ > 0 && NU_ > 0) {
                        for (int s = b + 1; s <= K; ++s) {
                            double yv[RP_], wv[RP_];
#pragma unroll
                            for (int k = 0; k < RP_; ++k) {
                                yv[k] = Y[(s - 1 - b) * RP_ * NU_ + RP_ * jc + k];
                                wv[k] = We[s * RP_ + k];
                            }
                            double sk = 0.0;
#pragma unroll
                            for (int k = 0; k < RP_; ++k) sk += wv[k] * yv[k];
                            acc += sk;
                        }
                    } else {
                        for (int s = b + 1; s <= K; ++s) {
                            const double* Yb = Y + (s - 1 - b) * r * nu + r * jc;
                            const double* Ws = We + s * r;
                            double sk = 0.0;
                            for (int k = 0; k < r; ++k) sk += Ws[k] * Yb[k];
                            acc += sk;
                        }
                    }
                }
                cj += acc;
            }
        }
        COPRA_FINE("costs:grad");
        // this lane's row descriptor and bounds: first needed by the norms below; fetched here rather than at kernel
        // start so that eleven registers are not carried (and spilled) across the preview and cost phases
        if (COPRA_LATE_ROW_CACHE) rows.cache_own_row();
        wave_sync();
        if (P.denseQ >= 0 && lane < n) { // host-evaluated user cost functions (COPRA_COST_DENSE): Q += Q_, c += c_ (LMPC.cpp:252-255)
            const double* Qd = P.params + P.denseQ + (size_t)n * lane;
            for (int i = 0; i <= lane; ++i) Q[fidx<TRI_>(i, lane, ld)] += Qd[i];
            cj += P.params[P.densec + lane];
        }
        if (lane < n) S.cvec[lane] = cj;
        wave_sync();
        if (inst == P.dump_instance && P.dumpQ) { // parity hook (LMPC::Q(), LMPC::c(); LMPC.h:113-115)
            if (lane < n) {
                for (int i = 0; i < n; ++i) {
                    const double v = (i <= lane) ? Q[fidx<TRI_>(i, lane, ld)] : Q[fidx<TRI_>(lane, i, ld)];
                    P.dumpQ[(size_t)lane * n + i] = v;
                }
                P.dumpc[lane] = cj;
            }
        }
    }
    stamp[2] = cycle_counter();
    if (lane == 0) S.scal[0] = 0.0; // the zero slot of StageRows::state_component (synchronised by the norms phase's barrier)
    // ---- 3. implicit rows: norms (qpgen2: column norms of amat) ----
    for (int i = lane; i < P.mgen; i += kWave) nb[i] = sqrt(rows.norm2(rows.desc(i)));
    if (inst == P.dump_instance && P.dumpA) { // parity hook (LMPC::Aeq/beq/Aineq/bineq; LMPC.h:116-123)
        for (int i = lane; i < P.mgen; i += kWave) {
            const RowDesc d = rows.desc(i);
            for (int j = 0; j < n; ++j) P.dumpA[(size_t)j * P.mgen + i] = rows.coeff(d, j);
            P.dumpb[i] = d.f - rows.lhs(d, Xbar, nullptr); // b = z - Y x0 (constraints.cpp:81)
        }
    }
    wave_sync();
    if (P.dump_only) return;
    stamp[3] = cycle_counter();
    // ---- 4. + 5. solve ----
    stamp[4] = stamp[3];
    int status = gi_factorize<NV, TRI_>(S, n, &stamp[4] COPRA_FINE_PASS);
    if (!TRI_ && P.model_out) { // "prepare" launch of the shared-model fast path (lmpc_shared.hpp); runs with the full layout
        if (inst != P.dump_instance) return;
        const ModelLayout m = model_layout(nx, nu, N, n, X, ld, P.mgen);
        double* out = P.model_out;
        if (lane < n) { // the factor as the factor-only tier wants it: packed columns, and the reciprocal pivots
            for (int i = 0; i <= lane; ++i) out[m.Rtri + rcol(lane) + i] = S.J[i * ld + lane];
            out[m.rinv + lane] = S.coef[lane];
        }
        wave_sync();
        if (status == 0) gi_invert<NV>(S, n);
        wave_sync();
        if (lane == 0) out[m.status] = (double)status;
        for (int e = lane; e < n * ld; e += kWave) out[m.J + e] = S.J[e];
        if (lane < n) { // Qinv(i, lane) = sum_{k >= max(i, lane)} J(i, k) J(lane, k)   (J upper triangular)
            for (int i = 0; i < n; ++i) {
                double acc = 0.0;
                for (int k = (i > lane ? i : lane); k < n; ++k) acc += S.J[i * ld + k] * S.J[lane * ld + k];
                out[m.Qinv + (size_t)i * ld + lane] = acc;
            }
        }
        for (int e = lane; e < N * nx * nu; e += kWave) out[m.G + e] = G[e];
        for (int e = lane; e < (N + 1) * nx * nx; e += kWave) out[m.Phi + e] = Phi[e];
        for (int e = lane; e < X; e += kWave) out[m.Xi + e] = Xi[e];
        for (int e = lane; e < P.mgen; e += kWave) out[m.nb + e] = nb[e];
        return;
    }
    stamp[5] = cycle_counter();
    int it_main = 0, it_drop = 0;
    if (status == 0)
        status = gi_active_set<NV, TRI_, QR_>(S, n, P.meq, P.mgen, rows, P.vsmall, P.max_iter, it_main, it_drop COPRA_FINE_PASS);
    wave_sync();
    stamp[6] = cycle_counter();
    if (status == 4) { // R outgrew the compact layout: queue for the second (full-layout) launch, write nothing else
        if (lane == 0) {
            const int slot = atomic_append(P.ovf_count);
            P.ovf_list[slot] = inst;
            P.status[inst] = 4;
        }
        return;
    }
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
#ifdef COPRA_FINE_PROFILE
        if (P.prof_fine) {
            long long* pf = P.prof_fine + 32 * (size_t)inst;
            for (int k = 0; k < 32; ++k) pf[k] = (k < copra_fine_n) ? copra_fine[k] - stamp[0] : -1;
        }
#endif
        if (P.prof) {
            stamp[7] = cycle_counter();
            long long* pr = P.prof + 8 * (size_t)inst;
            for (int k = 0; k < 7; ++k) pr[k] = stamp[k + 1] - stamp[k]; // preview, costs, norms, chol, inv, set, out
            pr[7] = stamp[7] - stamp[0];
        }
    }
}

} // namespace copra_hip
