// LMPC::solve() for one instance per wavefront, factor-only tier with the factor in RICCATI form.
//
// Same call stack as lmpc_fused.hpp (LMPC.cpp:79-101: updateSystem, makeQPForm, SI_solve, updateResults) and the same
// Goldfarb-Idnani iteration (gi_core.hpp, TRI), but the condensed Hessian  Q = 1e-6 I + sum Q_k  (LMPC.cpp:228-230,
// 252-255) is never formed and never factorised as an n x n matrix: for a controller whose costs are all per-step entries
// (TrajectoryCost / TargetCost / ControlCost / MixedCost without full-size weights, costFunctions.cpp:63-215) Q is the
// Hessian of a stage-wise LQ problem, and ONE backward Riccati sweep over the N stages yields a factor of Q^-1 whose two
// products are closed-loop recursions (ric_factor.hpp).  O(N (nx+nu)^3) = 15 k multiply-adds at the headline shape
// instead of 11 k (cost phase) + 72 k (Cholesky) + 7 k (two substitutions), and -- what matters on a latency-bound wave --
// twenty stages of small chained block products instead of the 35 k + 32 k + 8 k cycles those three phases take.
// Every matrix product of the body (sweep, preview recursion, roll-out, the two recursions of the active-set iteration) runs
// on v_mfma_f64_4x4x4 with results chained as the B operands of the next product; the trajectory is MAINTAINED (the closed-loop
// states of a step direction are its effect on the trajectory), so nothing sums over the blocks G after the row norms.
//     stage k < N:   l_k(x, u) = 1/2 [x; u]' Hin [x; u] + hin' [x; u],   Hin = sum_t [M_t N_t]' W_t [M_t N_t] (+ 1e-6 I on u),
//                                                                       hin = -sum_t [M_t N_t]' W_t p_t
//     stage N:       l_N(x) = 1/2 x' HN x + hN' x                        (TrajectoryCost and TargetCost terms)
// The unconstrained minimiser -Q^-1 c (qpgen2's starting point) is the LQ roll-out u_k = K_k x_k + kv_k from x_0.
// Constraint rows, bounds, status codes, iteration counts, results: exactly the fused body's (StageRows, gi_active_set).
// Shapes: compile-time (NX, NU, NH) with NX <= 7, 2 <= NU <= 3, NU NH <= 64 (the static_asserts below); instantiated for the CoM
// shape (6, 3) at N = 10, 15, 20.  Shared-model mode: FusedPlan::ric_model (the records come from one prepare run).
#pragma once
#include "lmpc_fused.hpp"

namespace copra_hip {

// SREFS: the build for controllers with reference trajectories (FusedPlan::stage_refs): the stage-varying affine term of the sweep in its
// own instantiation (in one build for both, the headline's kernel ran 3 % slower: 281 -> 290 us, twice the scalar-register spills)
template <int NX, int NU, int NH, int RP, int QR, bool SREFS = false>
COPRA_DEV void lmpc_fused_ric_body(const FusedPlan& P, int inst, bool lane_failed = false)
{
    // NH == 0: the horizon is a RUN-TIME value (P.N, with NU P.N <= 64): the builds the library ships for every horizon of a shape --
    // round-3 verdict: every shape but the BASELINE ones ran 2.5 x slower unless the USER's box had hipcc for copra_batch_specialise.
    // NH > 0: compile-time horizon (BASELINE shapes and run-time-compiled kernels): constant trip counts, exact register arrays.
    constexpr int NZ = NX + NU, NHM = NH ? NH : kWave / NU; // (NHM: the largest horizon this build may see)
    const int nh = NH ? NH : P.N;
    const int NV = NU * nh, X = NX * (nh + 1);
    constexpr int NVM = NU * NHM;
    constexpr int nxx = NX * (NX + 1) / 2, nux = NU * NX, nuu = NU * (NU + 1) / 2;
    static_assert(NX * (NZ + 1) <= kWave && nxx + nux + nuu + NZ <= kWave && NVM <= kWave, "one element per lane");
    static_assert(nxx + nux + nuu >= NX * (NX + 1), "affine lanes of the stage cost and of the terminal cost are disjoint (ric_tab)");
    using RR = RicRec<NX, NU>;
    double* lds = lds_base();
    const LdsLayout& L = P.lds;
    const int lane = lane_id();
    double* A = lds + L.A;
    double* B = lds + L.B;
    double* D = lds + L.D;
    double* X0 = lds + L.X0;
    double* G = lds + L.G;
    double* Xbar = lds + L.Xbar;
    double* Xcur = lds + L.Xcur;
    double* nb = lds + L.nb;
    SolverLds S = carve_solver(lds, L);
    double* F = S.J; // NH stage records
    double* const KV = lds + L.ricKv; // the feed-forward terms kv_k of the unconstrained minimiser, NU per stage (RicRec: not part of the records)
    // Where the trajectory lives once the roll-out has produced it.  Compact variant (LdsLayout::ricC: a row has a state term --
    // one component of one state -- or a control term, not both): the blocks G are NEVER STORED.  Their block-row norms, which is
    // all the row norms need, are taken by the preview steps; the normal of a state row enters w = R^-T n as a unit injection
    // (ric_factor.hpp); scans and results read the maintained trajectory.  `G` is then the trajectory's place, L.Xbar (behind
    // it) that of the closed-loop states of z = R^-1 v.  General variant: G as in lmpc_fused.hpp, the trajectory over A | B | d |
    // x0 | ricX.
    const bool compact = L.ricC != 0;
    double* XU = compact ? G : A;
    const bool xu_ok = compact || (P.rows_direct && (L.ricX + kWave - 2 - L.A) >= X && L.ricX > L.X0);
    // scratch of the sweep (aliases the solver vectors, which are written after it)
    double* Pm = lds + L.ricS; // NX x NX cost-to-go Hessian (symmetric, both halves)
    double* pv = Pm + NX * NX; // NX
    double* T = pv + ((NX + 1) & ~1); // NU x 12: the rows u of M (A operand of the update of P); 12 doubles of hand-over before the sweep
    double* Zs = T + ((NU * 12 + 1) & ~1); // [0]: holds 0.0 during the sweep (operands that are structurally zero); [1]: write-only spare

    long long stamp[8];
    COPRA_FINE_DECL;
    stamp[0] = cycle_counter();
    StageRows<NX, NU, NH> rows { P, G, Xbar, Xcur, nb, RowDesc {}, 0.0, 0.0 };
    rows.inst = inst;
    rows.zero = S.scal;
    int status = 0;
    // shared-model mode: the records, bkd, G (and later the row norms) of the whole batch come from the prepare launch
    int mBk = 0, mG = 0, mNb = 0;
    ric_model_offsets(NX, NU, nh, P.mgen, mBk, mG, mNb);
    const bool from_model = P.ric_model != nullptr;
    // behind the one-instance-per-lane pass (lmpc_lane.hpp) the sweep has been done already: K | kv | Lam^-1 of every stage and the running
    // block-row norms sit in its lane-major workspace, element (k, e) of this instance at lane_ws[(k WR + e) lane_bp + inst] -- a gather
    // of NH WR doubles (eleven loads per lane at the headline shape) and Acl = A + B K instead of the 32 k cycles of the sweep.
    // (Warming the L2 for the instance that gathers 176 / 352 list entries later -- the same loads issued once more, unused -- was
    //  measured and dropped: 0.49 -> 0.53 / 0.54 ms per step.)
    const bool from_lane = !from_model && compact && P.lane_from_list && P.lane_handover && !P.ric_model_out;
    if (lane_failed) status = 2; // (its factorisation met a control block that is not positive definite)
    // the unconstrained minimiser and its trajectory came from the SHARED-MODEL pass in front: no roll-out below.  (Behind the per-instance pass
    // the tier rolls out for itself since round 5 -- 2 k cycles --: where that pass has taken the first step of the iteration speculatively,
    // what it left in `control` / `trajectory` is that iterate, not the minimiser: lmpc_lane.hpp.)
    const bool have_ux = from_model && compact && P.lane_from_list && P.lane_handover && !P.lane_spec; // (a speculating pass leaves its deepest iterate, not the minimiser)
    if (from_lane) {
        // K from the pass's lane-major workspace (rows k (NU NX + NU) + e, e < NU NX: one 64-byte sector per value -- what is left of a gather
        // that round 4 measured at 16.5 k of the 44 k cycles of a tier instance, bound by the CU's sector requests), Lam^-1, kv and the norm
        // sums from this instance's hand-over block (FusedPlan::lane_ws2: contiguous).
        constexpr int KWl = NU * NX, WRK = KWl + NU, NL = NU * (NU + 1) / 2, NLU = NL + NU; // (plan.hpp: lane_ws_rows, lane_ws2_doubles)
        const double sysA = lane < NX * NX ? P.A[(size_t)inst * NX * NX + lane] : 0.0;
        const double sysB = lane < NX * NU ? P.B[(size_t)inst * NX * NU + lane] : 0.0;
        const double sysD = lane < NX ? P.d[(size_t)inst * NX + lane] : 0.0;
        const double sysX = lane < NX ? P.x0[(size_t)inst * NX + lane] : 0.0;
        constexpr int NGK = (NHM * KWl + kWave - 1) / kWave, NG2 = (NHM * (NLU + NX) + kWave - 1) / kWave;
        double gk[NGK], g2[NG2];
        const double* const wsb = P.lane_ws + (size_t)inst;
#pragma unroll
        for (int u = 0; u < NGK; ++u) {
            const int idx = lane + kWave * u, k = idx / KWl, e = idx - k * KWl;
            gk[u] = wsb[(size_t)(idx < nh * KWl ? k * WRK + e : 0) * (size_t)P.lane_bp];
        }
        const int t2 = nh * (NLU + NX);
        const double* const w2 = P.lane_ws2 + (size_t)inst * t2;
#pragma unroll
        for (int u = 0; u < NG2; ++u) g2[u] = w2[lane + kWave * u < t2 ? lane + kWave * u : 0];
        rows.cache_own_row();
        if (lane < NX * NX) A[lane] = sysA;
        if (lane < NX * NU) B[lane] = sysB;
        if (lane < NX) {
            D[lane] = sysD;
            X0[lane] = sysX;
        }
#pragma unroll
        for (int u = 0; u < NGK; ++u) {
            const int idx = lane + kWave * u, k = idx / KWl, e = idx - k * KWl;
            if (idx < nh * KWl) F[k * RR::SZ + RR::oK + e] = gk[u];
        }
#pragma unroll
        for (int u = 0; u < NG2; ++u) {
            const int idx = lane + kWave * u;
            if (idx < nh * NLU) {
                const int k = idx / NLU, e = idx - k * NLU;
                *(e < NL ? F + k * RR::SZ + RR::oLi + e : KV + k * NU + (e - NL)) = g2[u];
            } else if (idx < t2) {
                Xbar[idx - nh * NLU] = g2[u]; // (NB2: the running block-row norms, [stage][NX])
            }
        }
        wave_sync();
        // Acl_k = A + B K_k, every stage: element (i, j) of stage k
        for (int idx = lane; idx < nh * NX * NX; idx += kWave) {
            const int k = idx / (NX * NX), r = idx - k * (NX * NX), i = r % NX, j = r / NX;
            double acc = A[r];
#pragma unroll
            for (int c = 0; c < NU; ++c) acc += B[i + NX * c] * F[k * RR::SZ + RR::oK + c + NU * j];
            F[k * RR::SZ + RR::oAcl + r] = acc;
        }
        // the constant block behind the records (as after the sweep below)
        if (lane < NX * NU) F[nh * RR::SZ + RR::cB + lane] = B[lane];
        if (lane < NX) F[nh * RR::SZ + RR::cD + lane] = D[lane];
        if (lane == 0) F[nh * RR::SZ + RR::cZ] = 0.0;
        if (lane == 1) F[nh * RR::SZ + RR::cO] = 1.0;
        wave_sync();
        stamp[1] = cycle_counter();
    } else if (from_model) {
        if (lane < NX) X0[lane] = P.x0[(size_t)inst * NX + lane];
        rows.cache_own_row();
        for (int e = lane; e < nh * RR::SZ + RR::CST; e += kWave) F[e] = P.ric_model[e];
        if (!have_ux) { // (behind the shared-model lane pass in its plain form there is no roll-out here)
            // (an instance with references of its own: the pass has left the DELTA of its feed-forward terms, lane-major -- lmpc_lane_shared_body)
            bool own_refs = false;
            for (int t = 0; t < P.ncost; ++t) own_refs = own_refs || P.cost_p[t] != nullptr;
            own_refs = own_refs && P.lane_from_list && P.lane_ws != nullptr;
            for (int e = lane; e < NV; e += kWave)
                KV[e] = P.ric_model[mBk + e] + (own_refs ? P.lane_ws[(size_t)e * (size_t)P.lane_bp + inst] : 0.0);
        }
        if (!compact) // (compact variant: the blocks G are not kept at all -- the row norms come from the model too)
            for (int e = lane; e < nh * NX * NU; e += kWave) G[e] = P.ric_model[mG + e];
        if (have_ux) { // (behind the shared-model lane pass, lmpc_lane_shared_body: U and its trajectory are there already)
            if (lane < NV) S.xs[lane] = P.control[(size_t)inst * NV + lane];
            for (int e = lane; e < X; e += kWave) XU[e] = P.trajectory[(size_t)inst * X + e];
        }
        if (!xu_ok) {
            // rows that go through the free response of the preview (a dense state row: StageRows::refresh_trajectory reads Xbar, which the
            // preview steps of the per-instance path leave there): xbar_0 = x0, xbar_{k+1} = A xbar_k + d from THIS instance's x0 -- lane i owns
            // row i of the model's A.  (Round 4 had taken general rows out of this mode: the random differential test of the engine's modes
            // found statuses and U off -- Xbar was simply never written here, every such row's slack was read from stale LDS.)
            const int li = lane < NX ? lane : 0;
            const double* const Am = P.ric_model + ric_model_A(NX, NU, nh, P.mgen);
            double arow[NX];
#pragma unroll
            for (int j = 0; j < NX; ++j) arow[j] = Am[li + NX * j];
            wave_sync(); // (the constant block of the records and X0 are in place)
            const double di = F[nh * RR::SZ + RR::cD + li];
            if (lane < NX) Xbar[lane] = X0[lane];
            wave_sync();
            for (int k = 0; k < nh; ++k) {
                double acc = di;
#pragma unroll
                for (int j = 0; j < NX; ++j) acc += arow[j] * Xbar[k * NX + j];
                if (lane < NX) Xbar[(k + 1) * NX + lane] = acc;
                wave_sync();
            }
        }
        stamp[1] = cycle_counter();
    } else {
    // ---- 0. coalesced loads of this instance's system: into registers now, into LDS after the loads of the cost tables
    //      below have been issued as well (one trip to memory for both, not two) ----
    static_assert(NX * NX <= kWave, "one element of A per lane");
    const double sysA = lane < NX * NX ? P.A[(size_t)inst * NX * NX + lane] : 0.0;
    const double sysB = lane < NX * NU ? P.B[(size_t)inst * NX * NU + lane] : 0.0;
    const double sysD = lane < NX ? P.d[(size_t)inst * NX + lane] : 0.0;
    const double sysX = lane < NX ? P.x0[(size_t)inst * NX + lane] : 0.0;
    // ---- 0b. stage costs (they do not depend on the system: their loads overlap the ones above) ----
    // lane -> entry (a, b) of M = Hin + [A B]' P+ [A B] that it owns in the sweep:  x-x upper triangle | u-x | u-u upper
    // triangle | the affine column (b == NZ)
    // (the sweep takes Hin in accumulator layout from the plan builder's tables; of this lane layout only the affine lanes
    //  are still decoded: lane base + a owns hin(a), which may depend on per-instance references)
    const bool aff_lane = lane >= nxx + nux + nuu && lane < nxx + nux + nuu + NZ;
    const int ma = lane - (nxx + nux + nuu);
    // The plan builder has evaluated them per lane (plan_builder.hpp, build_ric_tables): entry of Hin / HN, and for the
    // affine lanes the coefficients of the references p_t, which may be per-instance -- 2 + RP coalesced loads per cost.
    const int pi = lane % NX, pj = lane / NX; // element (pi, pj) of an NX x (NZ + 1) table: [A B d] layout
    double hreg, term; // Hin(ma, mb) or hin(ma) in the affine column | lanes e < NX NX: HN(e % NX, e / NX), the next NX: hN
    double Hacc[3]; // Hin in the accumulator layout of the MFMA sweep (row blocks 0, 1, 2): fetched here, with the other tables
    {
        const double* tabq = P.params + P.ric_tab + kWave * (2 + kRicMaxCosts * RP);
#pragma unroll
        for (int I = 0; I < 3; ++I) Hacc[I] = tabq[kWave * I + lane];
    }
    {
        const int ti = pi, tj = pj; // tj == NX: hN
        const double* tab = P.params + P.ric_tab;
        hreg = tab[lane];
        term = tab[kWave + lane];
        double cw[kRicMaxCosts][RP], pr[kRicMaxCosts][RP];
#pragma unroll
        for (int t = 0; t < kRicMaxCosts; ++t) {
            const bool live = t < P.ncost;
            const int tt = live ? t : 0;
            const int rc = P.cost[tt].rows;
            // (a reference trajectory, CostTerm::pstride: the terminal lanes take the reference of the last step here, the affine terms of the
            //  stages are formed per stage below)
            const double* pref = cost_reference(P, tt, inst) + ((SREFS && P.cost[tt].pstride) ? P.cost[tt].prows - P.cost[tt].pstride : 0);
#pragma unroll
            for (int r = 0; r < RP; ++r) {
                cw[t][r] = tab[kWave * (2 + tt * RP + r) + lane];
                pr[t][r] = pref[r < rc ? r : (rc > 0 ? rc - 1 : 0)]; // (rows past the term's have zero coefficients; a cost that is not there reads
                                                      //  cost 0's -- every load unconditional: no branches in this prologue, whose
                                                      //  loads are all in flight together -- and is left out of the sum below)
            }
        }
        double aff = 0.0;
#pragma unroll
        for (int t = 0; t < kRicMaxCosts; ++t) {
            double at = 0.0;
#pragma unroll
            for (int r = 0; r < RP; ++r) at += cw[t][r] * pr[t][r];
            aff += (t < P.ncost) ? at : 0.0;
        }
        // (no lane owns an affine entry of the stage AND one of the terminal cost: the table holds whichever it has)
        hreg += (aff_lane && !SREFS) ? aff : 0.0;
        term += (tj == NX) ? aff : 0.0;
        if (tj < NX)
            Pm[lane] = term;
        else if (tj == NX)
            pv[ti] = term;
    }
    if (lane < NX * NX) A[lane] = sysA;
    if (lane < NX * NU) B[lane] = sysB;
    if (lane < NX) {
        D[lane] = sysD;
        X0[lane] = sysX;
    }
    if (P.ric_model_out && inst == P.dump_instance && lane < NX * NX) // prepare launch of the shared-model mode: the system's A (ric_model_A)
        P.ric_model_out[ric_model_A(NX, NU, nh, P.mgen) + lane] = sysA;
    // Reference trajectories (FusedPlan::stage_refs): the affine term of the stage cost changes along the horizon,
    //     h_k(a) = sum_t sum_r c_t(r, a) p_t[k r_t + r]     (c_t: the coefficients of the affine lanes in the table above),
    // NH NZ values, each formed once: entry a of stage k waits in the place of record k (where Acl_k goes at the END of stage k of the
    // sweep, which reads it at its start).
    if constexpr (SREFS) {
        const double* tab = P.params + P.ric_tab;
        for (int e = lane; e < nh * NZ; e += kWave) {
            const int k = e / NZ, a = e - k * NZ;
            double acc = 0.0;
#pragma unroll
            for (int t = 0; t < kRicMaxCosts; ++t) {
                const bool live = t < P.ncost;
                const int tt = live ? t : 0;
                const CostTerm& ct = P.cost[tt];
                const int rc = ct.rows, ps = ct.pstride;
                const int kk = (ps && (k + 1) * ps > ct.prows) ? ct.prows / ps - 1 : k; // (a cost without a step k has zero coefficients there)
                const double* pref = cost_reference(P, tt, inst) + kk * ps;
                double at = 0.0;
#pragma unroll
                for (int r = 0; r < RP; ++r)
                    at += tab[kWave * (2 + tt * RP + r) + nxx + nux + nuu + a] * pref[r < rc ? r : (rc > 0 ? rc - 1 : 0)];
                acc += live ? at : 0.0;
            }
            F[k * RR::SZ + RR::oAcl + a] = acc;
        }
    }
    COPRA_FINE("ric:costs");
    // ---- 1. preview: [G_s | xbar_s] = A [G_{s-1} | xbar_{s-1}] + [0 | d]  (G_0 = B, xbar_0 = x0; PreviewSystem.cpp:57-74
    //      applied to x0).  One step per stage of the sweep below, on v_mfma_f64_4x4x4: the NU + 1 columns are the four
    //      columns of a block, the two row blocks (x_0..3 | x_4..) are two accumulators, and an accumulator is laid out like
    //      the B operand of the next step -- no LDS reads, no waiting: the recursion runs in the shadow of the sweep. ----
    static_assert(NU + 1 <= 4 && NX <= 8, "preview recursion on 4 x 4 blocks");
    wave_sync();
    const int q4 = lane >> 4, r4 = lane & 3; // lane = 16 q + 4 b + r
    double pa[2][2] = { { 0.0, 0.0 }, { 0.0, 0.0 } }; // A operand (row 4 I + r, column 4 K + q) of A
    if (!compact) { // (compact variant: the recursion rides in the sweep's products, which read A themselves)
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int K = 0; K < 2; ++K) pa[I][K] = (4 * I + r4 < NX && 4 * K + q4 < NX) ? A[(4 * I + r4) + NX * (4 * K + q4)] : 0.0;
    }
    const bool pw = ((lane >> 2) & 3) == 0 && q4 < NX && (compact ? r4 < NU : r4 <= NU); // lanes of block 0 store, rows q < NX (compact variant: not the
                                                                              // free response, which nobody reads -- its place is inside G)
    const bool pgc = r4 < NU; // a column of G (else xbar)
    double pc[2], px[2]; // [0 | d] and the state, rows q and 4 + q, column r
#pragma unroll
    for (int I = 0; I < 2; ++I) {
        const int row = 4 * I + q4, rw = row < NX ? row : 0;
        // (compact variant: no free response -- its lanes carry zeros, so that the squares below need no mask)
        pc[I] = 0.0;
        px[I] = (row < NX && pgc) ? B[rw + NX * (pgc ? r4 : 0)] : 0.0;
        if (!compact) {
            pc[I] = (row < NX && r4 == NU) ? D[rw] : 0.0;
            px[I] = (row < NX && r4 == NU) ? X0[rw] : px[I];
        }
    }
    double* const pdst = pgc ? G + NX * r4 + q4 : Xbar + q4; // row q of step 0 (row 4 + q: + 4)
    const int pst = pgc ? NX * NU : NX;
    if (!compact) {
        if (lane < NX * NU) G[lane] = B[lane];
        if (lane < NX) Xbar[lane] = X0[lane];
    }
    // compact variant: the blocks G are not stored -- only |row i of G_s|^2, s = 0 .. NH-1, for the row norms (NB2 = Xbar's place:
    // free until z = R^-1 v first leaves its states there); lane r = 0 of block 0 writes rows q and 4 + q, the others a spare
    const bool nb2w = compact && ((lane >> 2) & 3) == 3 && r4 == 0; // (hardware block 3: where the preview rides, see the sweep)
    double* nb2p = nb2w ? Xbar + q4 : Zs + 1;
    double* nb2q = (nb2w && 4 + q4 < NX) ? Xbar + 4 + q4 : Zs + 1;
    const int nb2st = nb2w ? NX : 0, nb2qst = (nb2w && 4 + q4 < NX) ? NX : 0;
    double ncum0 = 0.0, ncum1 = 0.0; // running sums over the steps: a row at step k needs sum_{t < k}, ONE read later
    if (compact) {
        ncum0 = quad_sum(px[0] * px[0]), ncum1 = quad_sum(px[1] * px[1]);
        *nb2p = ncum0; // (block 0: G_0 = B)
        *nb2q = ncum1;
    }
    stamp[1] = cycle_counter();
    // ---- 2. backward Riccati sweep: stage records into F ----
    // Every matrix product of a stage runs on v_mfma_f64_4x4x4 in a stacked index space of three blocks of four,
    //     sigma:  0 .. NU-1 = u (block 0) | 4 .. 4+NX-1 = x (blocks 1, 2) | 4+NX = the affine column (d, p, h) | padding,
    // hardware block b of the instruction = COLUMN block b of the result, one accumulator per ROW block; lane 16 q + 4 b + r
    // then holds element (row q of the row block, column r of column block b) -- the layout of the B operand of the next
    // product, so T and M and K chain through registers.  A operands that are results themselves (P+ as the left factor of T,
    // M_ux' as the left factor of the update of P) go through LDS, which replicates them over the hardware blocks for free.
    //     T_I  = P+_{I,.} [B A d]  (+ p+ in the affine column)                 I = 1, 2       4 MFMAs   (A operand: P+ from LDS)
    //     M_I  = Hin_I + [B A]'_{I,.} T                                          I = 0, 1, 2    6 MFMAs   (B operand: T)
    //     M_uu by v_readlane, adjugate / determinant, each lane keeps its element of -M_uu^-1
    //     K    = -M_uu^-1 M_0                                                                    1 MFMA    (B operand: M_0)
    //     P_I  = M_I + M_{0,I}' K                                                I = 1, 2       2 MFMAs   (A operand: M_0 from LDS)
    //     [Acl | bkd]_I = [A | d]_I + B_I K                                      I = 1, 2       2 MFMAs   (off the chain)
    // Two LDS hand-overs per stage (P+, M_0) instead of three, ~60 vector-ALU instructions instead of ~110.
    // Lam^-1 (the Cholesky factor of M_uu, inverted) and Bt = B Lam^-T are only used by the active-set iteration: they are
    // formed after the sweep for all stages at once (lane = stage), not twenty times inside it.
    static_assert(NX <= 7 && NU <= 3, "stacked blocks of the MFMA sweep: u + pad | x_0..3 | x_4.., affine");
    // DECOUPLED AXES (FusedPlan::lane_axes: costs and rows couple none; here: neither does this instance's system): the recursion is NU
    // independent ones of NX / NU states and one control -- lane a runs axis a in scalar arithmetic, the sums of the pass's sweep
    // (lmpc_lane.hpp: stage_core) with the terms between axes left out: ~ 45 dependent multiply-adds per stage instead of fifteen
    // matrix instructions and two LDS hand-overs (30 k of an instance's cycles on the headline).  Stage costs from the pass's table
    // (controller-wide references only); the block-row norms of the compact variant by lane shuffles.
    bool axes_sweep = false;
    if constexpr (!SREFS && NU > 1 && NX % NU == 0) {
        bool own_refs = false;
        for (int t = 0; t < P.ncost; ++t) own_refs = own_refs || P.cost_p[t] != nullptr;
        if (P.lane_axes && P.lane_tab >= 0 && compact && !own_refs && !P.ric_model_out) {
            const int ei = lane % NX, ej = (lane / NX) % NX, bc = (lane / NX) % NU;
            const bool strayA = lane < NX * NX && ei % NU != ej % NU && !(sysA == 0.0);
            const bool strayB = lane < NX * NU && ei % NU != bc && !(sysB == 0.0);
            axes_sweep = !wave_any(strayA || strayB);
        }
    }
    if (axes_sweep) {
        if constexpr (!SREFS && NU > 1 && NX % NU == 0) {
            constexpr int NA = NX / NU, NZA = NA + 1; // states of an axis; its z = (states, control)
            for (int e = lane; e < nh * RR::SZ; e += kWave) F[e] = 0.0; // (Acl, K and Lam^-1 between axes: zero)
            const int ax = lane < NU ? lane : 0;
            auto zi = [&](int m) -> int { return m < NA ? ax + NU * m : NX + ax; }; // z-index of the axis's m-th entry
            const double* const tab = P.params + P.lane_tab;
            int oh_, oHN_, ohN_, oRows_;
            lane_tab_offsets(NX, NU, oh_, oHN_, ohN_, oRows_);
            double Aa[NA][NA], Ba[NA], da[NA], Ha[NZA][NZA], ha[NZA], Pa[NA][NA], pa_[NA];
#pragma unroll
            for (int m = 0; m < NA; ++m) {
#pragma unroll
                for (int n = 0; n < NA; ++n) {
                    Aa[m][n] = A[zi(m) + NX * zi(n)];
                    Pa[m][n] = Pm[zi(m) + NX * zi(n)];
                }
                Ba[m] = B[zi(m) + NX * ax];
                da[m] = D[zi(m)];
                pa_[m] = pv[zi(m)];
            }
#pragma unroll
            for (int m = 0; m < NZA; ++m) {
#pragma unroll
                for (int n = 0; n < NZA; ++n) Ha[m][n] = tab[zi(m) + (NX + NU) * zi(n)];
                ha[m] = tab[oh_ + zi(m)];
            }
            auto ABa = [&](int l, int z) -> double { return z < NA ? Aa[l][z] : Ba[l]; };
            wave_sync(); // (the records are zero)
            bool bad = false;
            for (int k = nh - 1; k >= 0; --k) {
                double tq[NA], mz[NZA], M[NZA][NZA];
#pragma unroll
                for (int l = 0; l < NA; ++l) {
                    double sacc = pa_[l];
#pragma unroll
                    for (int i = 0; i < NA; ++i) sacc += Pa[l][i] * da[i];
                    tq[l] = sacc;
                }
#pragma unroll
                for (int z = 0; z < NZA; ++z) {
                    double sacc = ha[z];
#pragma unroll
                    for (int l = 0; l < NA; ++l) sacc += ABa(l, z) * tq[l];
                    mz[z] = sacc;
                }
#pragma unroll
                for (int b2 = 0; b2 < NZA; ++b2) {
                    double Tb[NA];
#pragma unroll
                    for (int l = 0; l < NA; ++l) {
                        double sacc = 0.0;
#pragma unroll
                        for (int i = 0; i < NA; ++i) sacc += Pa[l][i] * ABa(i, b2);
                        Tb[l] = sacc;
                    }
#pragma unroll
                    for (int a2 = 0; a2 <= b2; ++a2) {
                        double sacc = Ha[a2][b2];
#pragma unroll
                        for (int l = 0; l < NA; ++l) sacc += ABa(l, a2) * Tb[l];
                        M[a2][b2] = sacc;
                    }
                }
                const double muu = M[NA][NA];
                bad = bad || (lane < NU && !(muu > 0.0));
                const double ni = -ric_rcp(muu); // (v_rcp_f64 + two Newton steps, as the matrix-instruction sweep below: the division is a third of the stage's chain)
                double Kk[NA];
#pragma unroll
                for (int j = 0; j < NA; ++j) Kk[j] = ni * M[j][NA];
                const double kvk = ni * mz[NA];
#pragma unroll
                for (int j = 0; j < NA; ++j)
#pragma unroll
                    for (int i = 0; i <= j; ++i) {
                        const double v = M[i][j] + M[i][NA] * Kk[j];
                        Pa[i][j] = v;
                        Pa[j][i] = v;
                    }
#pragma unroll
                for (int i = 0; i < NA; ++i) pa_[i] = mz[i] + M[i][NA] * kvk;
                if (lane < NU) {
                    double* Fk = F + k * RR::SZ;
#pragma unroll
                    for (int j = 0; j < NA; ++j) {
                        Fk[RR::oK + ax + NU * zi(j)] = Kk[j];
#pragma unroll
                        for (int i = 0; i < NA; ++i) Fk[RR::oAcl + zi(i) + NX * zi(j)] = Aa[i][j] + Ba[i] * Kk[j];
                    }
                    Fk[RR::oLi + ax * (ax + 1) / 2 + ax] = fast_rsqrt(muu);
                    KV[k * NU + ax] = kvk;
                }
            }
            if (wave_any(bad)) status = 2; // "Problems with the decomposition of Q" (QuadProgSolver.h:25)
            rows.cache_own_row();
            // the constant block behind the records: B (it takes the place of every Bt_k, ric_factor.hpp), d and a zero
            if (lane < NX * NU) F[nh * RR::SZ + RR::cB + lane] = B[lane];
            if (lane < NX) F[nh * RR::SZ + RR::cD + lane] = D[lane];
            if (lane == 0) F[nh * RR::SZ + RR::cZ] = 0.0;
            if (lane == 1) F[nh * RR::SZ + RR::cO] = 1.0;
            // the running block-row norms NB2[s][i] = sum_{t <= s} |row i of G_t|^2, G_t = A^t B: row i of G_t has ONE entry that is not zero,
            // g_i = G_t(i, i % NU), and g_i(t + 1) = sum over the states j of its axis of A(i, j) g_j(t) -- lane i < NX carries g_i
            {
                const int gi = lane < NX ? lane : 0;
                double aax[NA];
#pragma unroll
                for (int m = 0; m < NA; ++m) aax[m] = A[gi + NX * (gi % NU + NU * m)];
                double g = B[gi + NX * (gi % NU)], ncum = 0.0;
                for (int st = 0; st < nh; ++st) {
                    ncum += g * g;
                    if (lane < NX) Xbar[st * NX + lane] = ncum;
                    double acc = 0.0;
#pragma unroll
                    for (int m = 0; m < NA; ++m) acc += aax[m] * shfl_f64(g, gi % NU + NU * m);
                    g = acc;
                }
            }
        }
    } else {
        constexpr int SAFF = 4 + NX; // stacked index of the affine column
        const int q = lane >> 4, hb = (lane >> 2) & 3, r = lane & 3;
        const int scol = 4 * hb + r; // stacked column of this lane's results
        const bool col_x = scol >= 4 && scol < 4 + NX, col_aff = scol == SAFF;
        double* const Mu2 = T; // NU x 12: rows u of M (every stacked column), the A operand of the update of P
        double* const dummy = Zs + 1; // (lanes with nothing to store write here: no branches in the loop)
        // Hin in result layout: the quadratic part straight from the plan builder's table (second view), the affine column from
        // the lanes that hold hin(a) in `hreg`, through NZ doubles of LDS
        {
            double* AF = T; // [stacked row]: 12 doubles
            if (lane < 12) AF[lane] = 0.0;
            wave_sync();
            if (aff_lane) AF[(ma < NX) ? 4 + ma : ma - NX] = hreg;
            wave_sync();
            if (col_aff) {
#pragma unroll
                for (int I = 0; I < 3; ++I) Hacc[I] += AF[4 * I + q];
            }
            wave_sync();
        }
        if (lane < 2) Zs[lane] = 0.0;
        wave_sync(); // (the zero is read below, by other lanes)
        // row t of [B A d] at stacked column sc
        // (ONE unconditional read through a selected address -- LDS addresses are 32 bits -- instead of three masked reads and
        //  64-bit selects: this is straight-line set-up that every instance pays)
        auto abd = [&](int t, int sc) -> double {
            const bool tv = t >= 0 && t < NX;
            const double* p = Zs;
            p = (tv && sc < NU) ? B + t + NX * sc : p;
            p = (tv && sc >= 4 && sc < 4 + NX) ? A + t + NX * (sc - 4) : p;
            p = (tv && sc == SAFF) ? D + t : p;
            return *p;
        };
        double bK[3], aM[3][3], aB[3]; // (index 0 unused: K-blocks / row blocks 1 and 2 are the x blocks)
        const double* pA[3][3];
        const double* pC[3];
        const double* muA[3];
        // Compact variant: the preview recursion G_s = A G_{s-1} RIDES in hardware block 3 of the products T_I -- its columns (12..)
        // are padding there, and T_I = sum_K P+_{I,K} [B A d]_K has the shape of G_I = sum_K A_{I,K} G_K: the lanes of block 3 hand
        // in A instead of P+ and the previous G instead of [B A d], and take the new G out of T (four MFMAs less per stage).
        const bool ride = compact && hb == 3;
#pragma unroll
        for (int K = 1; K <= 2; ++K) {
            bK[K] = ride ? px[K - 1] : abd(4 * (K - 1) + q, scol); // B operand (row x_t of K-block K, column scol); also C of the closed loop
#pragma unroll
            for (int I = 0; I < 3; ++I) aM[I][K] = (4 * I + r == SAFF) ? 0.0 : abd(4 * (K - 1) + q, 4 * I + r); // A operand of M
        }
#pragma unroll
        for (int I = 1; I <= 2; ++I) {
            const int ir = 4 * (I - 1) + r, iq = 4 * (I - 1) + q; // x row as A-operand row / as result row
            aB[I] = *((ir < NX && q < NU) ? B + ir + NX * q : Zs);
#pragma unroll
            for (int K = 1; K <= 2; ++K) {
                const int kq = 4 * (K - 1) + q;
                pA[I][K] = (ir < NX && kq < NX) ? (ride ? A : Pm) + ir + NX * kq : Zs;
            }
            pC[I] = (col_aff && iq < NX) ? pv + iq : Zs;
            muA[I] = (ir < NX && q < NU) ? Mu2 + q + NU * (4 + ir) : Zs;
        }
        // where results go
        double* const wMu = (q < NU && hb < 3) ? Mu2 + q + NU * scol : dummy;
        const bool uu_on = q < NU && hb == 0 && r <= q;
        const int uu_off = RR::oLi + q * (q + 1) / 2 + r;
        const bool k_on = q < NU && (col_x || col_aff);
        // (K_k into record k, kv_k into the block of its own: one pointer per lane that walks down with the stages)
        double* wK = !k_on ? dummy : col_x ? F + (nh - 1) * RR::SZ + RR::oK + q + NU * (scol - 4) : KV + (nh - 1) * NU + q;
        const int wKst = !k_on ? 0 : col_x ? RR::SZ : NU;
        double* wP[3];
        double* wA[3];
        int wAst[3];
#pragma unroll
        for (int I = 1; I <= 2; ++I) {
            const int iq = 4 * (I - 1) + q;
            const bool on = iq < NX && (col_x || col_aff);
            wP[I] = !on ? dummy : col_x ? Pm + iq + NX * (scol - 4) : pv + iq;
            wAst[I] = (on && col_x) ? RR::SZ : 0;
            // (the affine column, d + B kv, is formed by the roll-out; the pointer walks down with the stages)
            wA[I] = (on && col_x) ? F + RR::oAcl + iq + NX * (scol - 4) + (nh - 1) * RR::SZ : dummy;
        }
        const bool my_minv = r < NU && q < NU; // A operand of K: element (row r, k = q) of -M_uu^-1
        // (NU == 3) this lane's cofactor of M_uu = x1 x2 - x3 x4, operands by their offsets in the rows-u buffer (M_uu(a, b) at a + NU b)
        const double *adj1 = Zs, *adj2 = Zs, *adj3 = Zs, *adj4 = Zs;
        if (NU == 3 && my_minv) {
            const int lo = r < q ? r : q, hi = r < q ? q : r, pr = 3 * lo + hi; // (symmetric: pair (lo, hi))
            // pairs: 00 -> 0, 01 -> 1, 02 -> 2, 11 -> 4, 12 -> 5, 22 -> 8;  entries (a, b) as a + 3 b
            const int e1 = pr == 0 ? 4 : pr == 1 ? 2 : pr == 2 ? 1 : pr == 4 ? 0 : pr == 5 ? 1 : 0;
            const int e2 = pr == 0 ? 8 : pr == 1 ? 5 : pr == 2 ? 5 : pr == 4 ? 8 : pr == 5 ? 2 : 4;
            const int e3 = pr == 0 ? 5 : pr == 1 ? 1 : pr == 2 ? 2 : pr == 4 ? 2 : pr == 5 ? 0 : 1;
            const int e4 = pr == 0 ? 5 : pr == 1 ? 8 : pr == 2 ? 4 : pr == 4 ? 2 : pr == 5 ? 5 : 1;
            adj1 = Mu2 + e1, adj2 = Mu2 + e2, adj3 = Mu2 + e3, adj4 = Mu2 + e4;
        }
        // reference trajectories: the affine column of Hin of stage k, from the place of record k (see above); everybody else reads a zero
        constexpr bool srefs = SREFS;
        const double* hkp[3];
        int hkst[3];
#pragma unroll
        for (int I = 0; I < 3; ++I) {
            const int sr = 4 * I + q, a = sr < NU ? NX + sr : (sr >= 4 && sr < 4 + NX) ? sr - 4 : -1; // stacked row -> index in z = (x, u)
            const bool on = srefs && col_aff && a >= 0;
            hkp[I] = on ? F + (nh - 1) * RR::SZ + RR::oAcl + a : Zs;
            hkst[I] = on ? RR::SZ : 0;
        }
        bool bad = false;
        wave_sync();
        for (int k = nh - 1; k >= 0; --k) {
            double* Fk = F + k * RR::SZ;
            double Hk[3] = { Hacc[0], Hacc[1], Hacc[2] };
            if constexpr (SREFS) {
#pragma unroll
                for (int I = 0; I < 3; ++I) {
                    Hk[I] += *hkp[I];
                    hkp[I] -= hkst[I];
                }
            }
            if (!compact) { // preview step s = NH - k
                const int s = nh - k;
                double n0 = mfma_f64_4x4x4(pa[0][0], px[0], pc[0]);
                double n1 = mfma_f64_4x4x4(pa[1][0], px[0], pc[1]);
                if (NX > 4) {
                    n0 = mfma_f64_4x4x4(pa[0][1], px[1], n0);
                    n1 = mfma_f64_4x4x4(pa[1][1], px[1], n1);
                }
                px[0] = n0;
                px[1] = n1;
                if (pw && (s < nh || !pgc)) {
                    pdst[s * pst] = n0;
                    if (4 + q4 < NX) pdst[s * pst + 4] = n1;
                }
            }
            // T_I = P+_{I,.} [B A d] + [0 | p+]
            double T1 = mfma_f64_4x4x4(*pA[1][1], bK[1], *pC[1]);
            double T2 = mfma_f64_4x4x4(*pA[2][1], bK[1], *pC[2]);
            if (NX > 4) {
                T1 = mfma_f64_4x4x4(*pA[1][2], bK[2], T1);
                T2 = mfma_f64_4x4x4(*pA[2][2], bK[2], T2);
            }
            // M_I = Hin_I + [B A]'_{I,.} T
            double M0 = mfma_f64_4x4x4(aM[0][1], T1, Hk[0]);
            double M1 = mfma_f64_4x4x4(aM[1][1], T1, Hk[1]);
            double M2 = mfma_f64_4x4x4(aM[2][1], T1, Hk[2]);
            if (NX > 4) {
                M0 = mfma_f64_4x4x4(aM[0][2], T2, M0);
                M1 = mfma_f64_4x4x4(aM[1][2], T2, M1);
                M2 = mfma_f64_4x4x4(aM[2][2], T2, M2);
            }
            // (The compact variant's use of T stands BEHIND the products M on purpose.  Between T and M it was a wave-uniform branch, and on
            //  its taken side -- the variant with general rows -- the compiler (ROCm 7.2, gfx950) left ONE instruction between the
            //  v_mfma_f64_4x4x4 that writes T1 and the one that reads it as its B operand: the hazard recogniser had counted the wait states
            //  of the fall-through side only.  The hardware then multiplies with whatever the register held before; which instantiations
            //  were hit was a matter of scheduling -- <4, 2> and <2, 1> with Q1 in LDS, found by the random differential test on the device,
            //  tests/random_controllers.py::make_integrator.  Dependent matrix instructions are kept in straight-line code from here on;
            //  tools/mfma_hazard_lint.py checks the compiled kernels for the pattern.)
            if (compact) { // block 3 of T: G_s, s = NH - k -- its block-row norms are all that is kept (the other blocks' sums go nowhere)
                if (k > 0) {
                    ncum0 += quad_sum(T1 * T1);
                    ncum1 += quad_sum(T2 * T2);
                    nb2p += nb2st;
                    nb2q += nb2qst;
                    *nb2p = ncum0;
                    *nb2q = ncum1;
                }
                bK[1] = ride ? T1 : bK[1];
                bK[2] = ride ? T2 : bK[2];
            }
            mfma_settle(); // (the branch above merges here, M0 is stored next)
            *wMu = M0;
            if (uu_on) Fk[uu_off] = M0; // (packed upper triangle; replaced by Lam^-1 after the sweep)
            wave_sync();
            const double mu1 = *muA[1], mu2 = *muA[2]; // (read back now: the latency hides under the inversion of M_uu below)
            // M_uu(c, c') sits in lane 16 c + c' of M0
            double mine = 0.0; // element (r, q) of -M_uu^-1
            if constexpr (NU == 3) {
                // 3 x 3: adjugate over determinant -- ONE reciprocal.  The loop is bound by instruction issue, so the block is
                // taken from LDS (the rows u of M were written there above for the update of P): six reads at wave-uniform
                // addresses for the determinant, four at this lane's own addresses for ITS cofactor (element (r, q) of the
                // adjugate) -- instead of six v_readlane pairs and a chain of selects.  Positive definite <=> the trailing
                // minors m22, C00, det are positive (Sylvester)
                const double m00 = Mu2[0], m01 = Mu2[1], m02 = Mu2[2], m11 = Mu2[1 + NU], m12 = Mu2[2 + NU], m22 = Mu2[2 + 2 * NU];
                const double x1 = *adj1, x2 = *adj2, x3 = *adj3, x4 = *adj4;
                const double c00 = m11 * m22 - m12 * m12, c01 = m02 * m12 - m01 * m22, c02 = m01 * m12 - m02 * m11;
                const double det = m00 * c00 + (m01 * c01 + m02 * c02);
                bad = bad || !(m22 > 0.0) || !(c00 > 0.0) || !(det > 0.0);
                const double nrdet = -ric_rcp(det);
                mine = (x1 * x2 - x3 * x4) * nrdet; // (lanes outside the block read zeros)
            } else {
                double lm[NU][NU], rd[NU], li[NU][NU];
#pragma unroll
                for (int cb = 0; cb < NU; ++cb)
#pragma unroll
                    for (int ca = 0; ca <= cb; ++ca) lm[cb][ca] = bcast_f64(M0, 16 * cb + ca);
#pragma unroll
                for (int c = 0; c < NU; ++c) {
                    double dl[NU];
#pragma unroll
                    for (int t = 0; t < c; ++t) dl[t] = lm[c][t];
                    double dc = lm[c][c];
#pragma unroll
                    for (int t = 0; t < c; ++t) {
                        const double lct = dl[t] * rd[t];
                        dc -= lct * dl[t];
                        lm[c][t] = lct;
                    }
                    bad = bad || !(dc > 0.0);
                    rd[c] = ric_rcp(dc);
#pragma unroll
                    for (int r2 = c + 1; r2 < NU; ++r2) {
                        double v = lm[r2][c];
#pragma unroll
                        for (int t = 0; t < c; ++t) v -= lm[r2][t] * lm[c][t];
                        lm[r2][c] = v;
                    }
                }
#pragma unroll
                for (int c = 0; c < NU; ++c)
#pragma unroll
                    for (int r2 = c; r2 < NU; ++r2) {
                        double v = (r2 == c) ? 1.0 : 0.0;
#pragma unroll
                        for (int t = c; t < r2; ++t) v -= lm[r2][t] * li[t][c];
                        li[r2][c] = v;
                    }
#pragma unroll
                for (int r2 = 0; r2 < NU; ++r2)
#pragma unroll
                    for (int c = r2; c < NU; ++c) {
                        double v = 0.0;
#pragma unroll
                        for (int t = c; t < NU; ++t) v += (li[t][r2] * rd[t]) * li[t][c];
                        mine = ((r == r2 && q == c) || (r == c && q == r2)) ? -v : mine;
                    }
                mine = my_minv ? mine : 0.0;
            }
            // K = -M_uu^-1 M_0  (rows u: K | kv in the affine column)
            const double Kr = mfma_f64_4x4x4(mine, M0, 0.0);
            *wK = Kr;
            wK -= wKst;
            // P_I = M_I + M_{0,I}' K   (rows / columns x: the new cost-to-go; affine column: p)
            const double P1 = mfma_f64_4x4x4(mu1, Kr, M1);
            const double P2 = mfma_f64_4x4x4(mu2, Kr, M2);
            *wP[1] = P1;
            if (NX > 4) *wP[2] = P2;
            // [Acl | bkd]_I = [A | d]_I + B_I K
            const double A1 = mfma_f64_4x4x4(aB[1], Kr, bK[1]);
            const double A2 = mfma_f64_4x4x4(aB[2], Kr, bK[2]);
            *wA[1] = A1;
            if (NX > 4) *wA[2] = A2;
            wA[1] -= wAst[1];
            wA[2] -= wAst[2];
            wave_sync();
            if (k == nh - 2) COPRA_FINE("sweep:stage");
            if (k == nh - 1) COPRA_FINE("sweep:first");
        }
        COPRA_FINE("sweep:loop");
        if (bad) status = 2; // "Problems with the decomposition of Q" (QuadProgSolver.h:25)
        rows.cache_own_row(); // (global loads of this lane's row descriptor and bounds: their latency hides under the passes below)
        // Lam^-1 of every stage (lane = stage): Cholesky of the stored M_uu, inverted in place of it
        if (lane < nh) {
            double* Fk = F + lane * RR::SZ;
            double lm[NU][NU], rd[NU], li[NU][NU];
#pragma unroll
            for (int cb = 0; cb < NU; ++cb)
#pragma unroll
                for (int ca = 0; ca <= cb; ++ca) lm[cb][ca] = Fk[RR::oLi + cb * (cb + 1) / 2 + ca];
#pragma unroll
            for (int c = 0; c < NU; ++c) {
                double dpp = lm[c][c];
#pragma unroll
                for (int q = 0; q < c; ++q) dpp -= lm[c][q] * lm[c][q];
                rd[c] = fast_rsqrt(dpp);
#pragma unroll
                for (int r2 = c + 1; r2 < NU; ++r2) {
                    double v = lm[r2][c];
#pragma unroll
                    for (int q = 0; q < c; ++q) v -= lm[r2][q] * lm[c][q];
                    lm[r2][c] = v * rd[c];
                }
            }
#pragma unroll
            for (int c = 0; c < NU; ++c)
#pragma unroll
                for (int r2 = 0; r2 < NU; ++r2) {
                    if (r2 < c) {
                        li[r2][c] = 0.0;
                    } else if (r2 == c) {
                        li[r2][c] = rd[c];
                    } else {
                        double v = 0.0;
#pragma unroll
                        for (int q = c; q < r2; ++q) v -= lm[r2][q] * li[q][c];
                        li[r2][c] = v * rd[r2];
                    }
                }
#pragma unroll
            for (int c = 0; c < NU; ++c)
#pragma unroll
                for (int r2 = c; r2 < NU; ++r2) Fk[RR::oLi + r2 * (r2 + 1) / 2 + c] = li[r2][c];
        }
        wave_sync();
        COPRA_FINE("sweep:Li");
        // the constant block behind the records: B (it takes the place of every Bt_k, ric_factor.hpp), d and a zero
        if (lane < NX * NU) F[nh * RR::SZ + RR::cB + lane] = B[lane];
        if (lane < NX) F[nh * RR::SZ + RR::cD + lane] = D[lane];
        if (lane == 0) F[nh * RR::SZ + RR::cZ] = 0.0;
        if (lane == 1) F[nh * RR::SZ + RR::cO] = 1.0;
    }
    } // (!from_model)
    wave_sync();
    if (P.ric_model_out && inst == P.dump_instance) { // prepare launch of the shared-model mode, first half
        for (int e = lane; e < nh * RR::SZ + RR::CST; e += kWave) P.ric_model_out[e] = F[e];
        for (int e = lane; e < NV; e += kWave) P.ric_model_out[mBk + e] = KV[e];
        if (!compact)
            for (int e = lane; e < nh * NX * NU; e += kWave) P.ric_model_out[mG + e] = G[e];
    }
    COPRA_FINE("sweep:Bt");
    stamp[2] = cycle_counter();
    // ---- 3. implicit rows: norms (qpgen2: column norms of amat).  Before the roll-out: in the compact variant the trajectory
    //      will take the place of the blocks G. ----
    // A row of Psi (one component of one state, no control term: TrajectoryBoundConstraint) has the squared norm
    // sum_{t < k} |row eo of G_t|^2: the NH NX block-row norms once (two per lane), then at most NH additions per row --
    // instead of 3 k multiply-adds behind as many dependent LDS reads for the row of the last step.  The block-row norms sit
    // at Xbar's place (free: the preview's free response is not used); in the compact variant the preview steps wrote them.
    // (compact variant: the norm of row `lane` stays in a register -- StageRows::nb_mine --, only rows 64.. go to LDS)
    const double* const X0r = compact ? XU : X0; // (compact variant: the norms of rows 64.. overwrite the system's slots)
    if (compact && !have_ux) { // (have_ux: the whole trajectory is there already)
        if (lane < NX) XU[lane] = X0[lane];
        wave_sync();
    }
    rows.nb_split = compact;
    auto put_norm = [&](int i, double v) {
        if (!compact)
            nb[i] = v;
        else if (i < kWave)
            rows.nb_mine = v;
        else
            nb[i - kWave] = v;
    };
    if (from_model) {
        for (int i = lane; i < P.mgen; i += kWave) put_norm(i, P.ric_model[mNb + i]);
    } else {
        const bool fast = xu_ok;
        for (int i = lane; i < P.mgen; i += kWave) {
            const RowDesc d = rows.desc(i);
            if (!(fast && e_onehot(d.ek) && d.gk == kGNone)) put_norm(i, sqrt(rows.norm2(d)));
        }
        if (fast) {
            double* NB2 = Xbar;
            const int tst = NX; // stride between the blocks
            wave_sync();
            if (!compact) { // (compact variant: the preview steps have left the block-row norms there)
            double s2[(NHM * NX + kWave - 1) / kWave];
#pragma unroll
            for (int u = 0; u < (NHM * NX + kWave - 1) / kWave; ++u) {
                const int e = lane + kWave * u, t = e / NX, comp = e - t * NX;
                s2[u] = 0.0;
                if (e < nh * NX) {
#pragma unroll
                    for (int c = 0; c < NU; ++c) {
                        const double a = G[t * NX * NU + comp + NX * c];
                        s2[u] += a * a;
                    }
                }
            }
            wave_sync();
#pragma unroll
            for (int u = 0; u < (NHM * NX + kWave - 1) / kWave; ++u) {
                const int e = lane + kWave * u, t = e / NX, comp = e - t * NX;
                if (e < nh * NX) NB2[t * tst + comp] = s2[u];
            }
            wave_sync();
            }
            for (int i = lane; i < P.mgen; i += kWave) {
                const RowDesc d = rows.desc(i);
                if (compact && e_onehot(d.ek) && d.gk == kGNone) { // (the preview steps left RUNNING sums)
                    put_norm(i, d.k > 0 ? sqrt(NB2[(d.k - 1) * tst + d.eo]) : 0.0);
                } else if (e_onehot(d.ek) && d.gk == kGNone) {
                    double acc = 0.0;
                    if constexpr (NH > 0) {
                        double part[NHM];
#pragma unroll
                        for (int t = 0; t < NHM; ++t) part[t] = NB2[(t < d.k ? t : 0) * tst + d.eo];
#pragma unroll
                        for (int t = 0; t < NHM; ++t) acc += (t < d.k) ? part[t] : 0.0;
                    } else {
                        for (int t = 0; t < d.k && t < nh; ++t) acc += NB2[t * tst + d.eo]; // (the same order of summation)
                    }
                    put_norm(i, sqrt(acc));
                }
            }
        }
    }
    wave_sync();
    if (P.ric_model_out) { // prepare launch, second half: the row norms; nothing is solved
        if (inst == P.dump_instance)
            for (int i = lane; i < P.mgen; i += kWave) P.ric_model_out[mNb + i] = rows.norm(i);
        return;
    }
    rows.g_dead = compact; // (from here on nothing reads the blocks G: StageRows::load_normal_split)
    COPRA_FINE("norms");
    stamp[3] = cycle_counter();
    // ---- 4. unconstrained minimiser: roll-out [x_{k+1}; u_k] = [Acl_k B d; K_k I 0] [x_k; kv_k; 1] from x_0, on the matrix
    //      cores like the recursions of ric_factor.hpp (v_mfma_f64_4x4x4: block b of the lane = rows 4b .. 4b+3 of the
    //      stacked matrix, the state handed on by a DPP row broadcast) ----
    if (!have_ux) { // (have_ux: U and its trajectory came from the pass in front)
        const int q = lane >> 4, b4 = (lane >> 2) & 3, r = lane & 3, row = 4 * b4 + r;
        int off[2], km[2];
#pragma unroll
        for (int J = 0; J < 2; ++J) off[J] = ric_stack_offset<NX, NU, NH>(row, 4 * J + q, km[J], nh);
        // K-block 2: the stacked input is (kv_k, 1) -- lane row q holds kv_k(q), lane row NU the constant 1 -- and the matrix
        // [B d] for the state rows (B kv + d = bkd_k: never stored), the identity for the rows u: the same at every stage
        const double a2 = (row < NX && q <= NU) ? F[nh * RR::SZ + (q < NU ? RR::cB + row + NX * q : RR::cD + row)]
                                                : (b4 == 2 && r == q && q < NU) ? 1.0 : 0.0;
        // (kv_k from its block -- in the compact variant the tail of XU: the state x_{k+1} this loop stores at NX (k + 1) .. stays below the
        //  kv_{k+2} .. still to be read because NX >= NU, and kv_{k+1} is in a register by then: layout_lds_ric)
        const double* kvp = (q < NU) ? KV + q : F + nh * RR::SZ + (q == NU ? RR::cO : RR::cZ);
        const int kvst = (q < NU) ? NU : 0;
        const bool writer = q < NU && b4 == 2 && r == 0;
        // the states of the roll-out (rows 0 .. NX-1 of the stacked result: blocks 0 and 1) are the trajectory at the
        // unconstrained minimiser: kept for the first scan and, if that finds nothing violated, for the results
        const int yrow = 4 * b4 + q; // (the RESULT of lane 16 q + 4 b + r is row 4 b + q of the stacked product)
        const bool xwriter = xu_ok && b4 < 2 && r == 0 && yrow < NX;
        const double x0r = X0r[yrow < NX ? yrow : 0];
        // (one store per stage: the lanes that own an output row or a state row; the others write to a spare double -- no
        //  branches in the loop; operand pointers with per-lane strides, fetched right after their last use: ric_apply_mfma4)
        double* sp = writer ? S.xs + q : xwriter ? XU + NX + yrow : S.ricd;
        const int sst = writer ? NU : xwriter ? NX : 0;
        double s0 = X0r[q < NX ? q : 0], s1 = (4 + q < NX) ? X0r[4 + q < NX ? 4 + q : 0] : 0.0;
        const double* p0 = F + off[0];
        const double* p1 = F + off[1];
        double a0 = *p0, a1 = *p1, kv = *kvp; // (stage 0)
        wave_sync(); // (every lane has read x0: XU overwrites the system's slots)
        if (xwriter) XU[yrow] = x0r;
#pragma unroll COPRA_RIC_UNROLL
        for (int k = 0; k < nh; ++k) {
            double y = mfma_f64_4x4x4(a2, kv, 0.0);
            kvp += kvst; // (behind the last stage: one record past the end, unused)
            kv = *kvp;
            y = mfma_f64_4x4x4(a0, s0, y);
            p0 += km[0];
            a0 = *p0;
            y = mfma_f64_4x4x4(a1, s1, y);
            p1 += km[1];
            a1 = *p1;
            *sp = y;
            sp += sst;
            s0 = row_bcast_f64<0>(y);
            s1 = row_bcast_f64<4>(y);
        }
    }
    if (xu_ok) {
        rows.xu = XU;
        rows.xi = Xbar; // the trajectory is maintained from here on (StageRows::moved): the free response is not needed any more
        S.ricxi = Xbar;
    }
    COPRA_FINE("rollout");
    if (lane == 0) S.scal[0] = 0.0; // the zero slot of StageRows::state_component
    wave_sync();
    stamp[4] = cycle_counter();
    stamp[5] = stamp[4];
    // ---- 5. active set ----
    int it_main = 0, it_drop = 0;
    if (status == 0)
        status = gi_active_set<NU * NH, true, QR, NX, NU>(S, NV, P.meq, P.mgen, rows, P.vsmall, P.max_iter, it_main, it_drop COPRA_FINE_PASS);
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
        const double* Xres = Xcur;
        if (rows.xu && (it_main == 1 || rows.xi)) { // the trajectory at the final iterate is there already (nothing was violated at the
                                                      // unconstrained minimiser, or every step has been applied to it)
            Xres = rows.xu;
        } else {
            rows.refresh_trajectory(S.xs);
            wave_sync();
        }
        for (int e = lane; e < NV; e += kWave) P.control[(size_t)inst * NV + e] = S.xs[e];
        for (int e = lane; e < X; e += kWave) P.trajectory[(size_t)inst * X + e] = Xres[e];
    } else {
        const double qnan = __builtin_nan("");
        for (int e = lane; e < NV; e += kWave) P.control[(size_t)inst * NV + e] = qnan;
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
        if (P.prof) { // set-up | Riccati sweep (+ preview) | row norms | roll-out | - | active set | results | total
            stamp[7] = cycle_counter();
            long long* pr = P.prof + 8 * (size_t)inst;
            for (int k = 0; k < 7; ++k) pr[k] = stamp[k + 1] - stamp[k];
            pr[7] = stamp[7] - stamp[0];
        }
    }
}

} // namespace copra_hip
