// LMPC::solve() with ONE (INSTANCE, AXIS) PER LANE: the whole solve -- LQ sweep, roll-out AND the active-set iteration -- for
// controllers whose axes are decoupled (round 6).
//
// The CoM model of BASELINE configs[2], the point masses, the reference's falling mass: state i belongs to axis i % nu (x = (p, v)) or to
// axis i / nxa (x = (p_x, v_x, p_y, ..): FusedPlan::axis_order, seen from the first system a controller is given), control c to
// axis c, and neither A, B, the costs (costFunctions.cpp:63-215) nor any constraint row (constraints.cpp:66-315) couple two axes.  Then
// the condensed QP of LMPC::makeQPForm (LMPC.cpp:250-280) is block diagonal -- nu independent QPs over one chain each (nxa = nx / nu
// states, ONE control) -- and qpgen2's run on the whole problem (QuadProgSolver.cpp:45-72) is an INTERLEAVING of the axes' own runs: a
// step of one axis moves no iterate and no multiplier of another, the pick within an axis is the same whenever it is taken, the
// counters add up (tools/exp/axis_proto.py replays this against the oracle: status, both counters, U).  So every (instance, axis) gets a
// lane, and the lane does everything:
//   * backward Riccati sweep of its chain (the recursion of lmpc_lane.hpp with nu = 1): K_k, 1 / M_uu,k and kv_k stay in REGISTERS
//     -- no workspace in memory, nothing written but the results;
//   * roll-out = qpgen2's starting point -Q^-1 c, with the first scan inside;
//   * the Goldfarb-Idnani iteration in RANGE-SPACE form on the Riccati factor.  With N the active normals, n+ the pick's:
//         y = Q^-1 n+                        one backward + one forward closed-loop recursion over the stages (ric_factor.hpp, scalar)
//         g = N' y,   S = N' Q^-1 N          (kept explicitly: QMAX x QMAX, padded with the identity),   r = S^-1 g  (Cholesky, in registers)
//         z = Q^-1 (n+ - N r)                the same two recursions once more, the new iterate U += t z and the next scan inside
//         t1 = min lambda_i / r_i (r_i > 0),   t2 = -s / (n+' Q^-1 n+ - g' r)
//     -- qpgen2's d = J' n+, z = J2 d2, r = R^-1 d1 in another basis: the same pick rule (most violated normalised by the row norm, lowest index
//     wins), the same step lengths, the same iterates up to rounding.  A lane's active set is at most QMAX constraints.
// Anything the lane cannot decide exactly as qpgen2 would -- an active set that outgrows QMAX, a pick without a step in primal space
// (|z|^2 <= vsmall: the infeasible and the degenerate cases), a failed factorisation, systems whose axes ARE coupled -- sends the
// INSTANCE to the list of the first tier (lmpc_fused_ric.hpp), which solves it from scratch, as it does behind lmpc_lane.hpp.
//
// What a controller may bring besides (each checked where it is set, copra_hip.hip: axis_solver_wanted): chains of two or three states per control;
// a reference per instance (copra_batch_set_cost_reference) and reference trajectories -- the lane rebuilds the affine terms of its axis from the
// plan's coefficients (FusedPlan::axis_cref), stage by stage for a trajectory --; limits per instance (copra_batch_set_control_bounds,
// _set_constraint_rhs) where they are the same along the horizon; one model for the batch (written out per instance: copra_batch_set_shared_system).
//
// Lane mapping: lane = nu * (instance of the wave) + axis, 64 / nu instances per wave; nothing crosses lanes inside the iteration (a
// wave leaves the loop when its last lane has).  Per lane in LDS: one sparse array over its constraints (N controls + (N + 1) rpa
// rows) -- coefficients of the combined normal on the way into a recursion, responses n_i' y on the way out.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "lmpc_fused.hpp"

namespace copra_hip {

#ifndef COPRA_AXIS_ITER_CAP
#ifndef COPRA_AXIS_ZN_REL
#define COPRA_AXIS_ZN_REL 1e-13 // z'n+ below this share of n+' Q^-1 n+: the direction counts as zero (checked against |z|^2 behind the step)
#endif
#define COPRA_AXIS_ITER_CAP 64 // picks + drops of one lane; beyond: the tier
#endif

// the largest double below x > 0:  s < -axis_pred(x)  <=>  s <= -x  (qpgen2's test "|s| < vsmall counts as zero, a negative slack is a violation")
COPRA_DEV double axis_pred(double x)
{
    long long b;
    std::memcpy(&b, &x, sizeof b);
    b -= 1;
    std::memcpy(&x, &b, sizeof b);
    return x;
}

// EXACT: the horizon IS NMAX (the builds of the BASELINE horizons: no stage of the sweep and of the scan is guarded)
// CT: the tables are the same at every step -- every row of the controller a pure state row present at all N + 1 steps with one E and f, the
//     bounds of a control the same along the horizon (FusedPlan::axis_const: what TrajectoryBoundConstraint and ControlBoundConstraint produce
//     from per-step entries, constraints.cpp:284-315, 359-367): they live in registers, nothing but the lane's sparse array is read from LDS
// RPA_: constraint rows per axis and step the build has registers for (FusedPlan::axis_rpa <= RPA_)
// LIST: the SECOND CHANCE of the instances the first launch listed (FusedPlan::axis_list_in): the same solver with room for more active
//       constraints per lane (QMAX > 8: S, its factor and the small vectors in the lane's LDS, loops with run-time trip counts), the wave's
//       instances taken from the list, loaded and stored lane by lane; what it cannot finish either goes on to the first tier's list
template <int NXA, int NU, int NMAX, int QMAX, bool EXACT = false, bool CT = false, int RPA_ = kAxisMaxRpa, bool LIST = false>
COPRA_DEV void lmpc_axis_body(const FusedPlan& P, int group)
{
    constexpr int NX = NXA * NU, NZ = NXA + 1, RW = NXA + 3, RPA = RPA_;
    constexpr int IPW = kWave / NU; // instances per wave
    static_assert(NXA >= 1 && NXA <= 3 && QMAX >= 1 && QMAX <= 16 && NMAX <= 31, "registers; the stage masks are 32 bits");
    const int lane = lane_id();
    // Lanes: NU consecutive lanes per instance, IPW instances per wave.  Where NU does not divide 64 the SPARE lanes (one per wave for three
    // axes) take the axes of the instances behind the last wave's, NU spare lanes -- of NU consecutive waves -- per instance: 65 536 instances
    // x 3 axes are exactly 3072 full waves, three rounds on the chip's 1024 SIMDs (with 21 instances per wave and an idle lane: 3121 waves,
    // four rounds).  Such an instance's counters meet in a word of FusedPlan::axis_acc (below).
    constexpr int SP = LIST ? 0 : kWave - IPW * NU; // spare lanes per wave
    const int il = lane / NU;
    const bool lane_on = il < IPW;
    const int nwaves = P.axis_waves;
    const int list_n = LIST ? *P.axis_list_count : 0; // (second chance: entries of the first launch's list)
    const int nreg = LIST ? list_n : (IPW * nwaves < P.batch ? IPW * nwaves : P.batch); // instances on regular lanes
    const int spare = SP > 0 ? group * SP + (lane - IPW * NU) : 0; // this spare lane's number
    const int orph_e = SP > 0 ? spare / NU : 0; // ... the instance it works for, counted from nreg
    const bool orphan = SP > 0 && !lane_on && nreg + orph_e < P.batch;
    const int c = lane_on ? lane - il * NU : spare - orph_e * NU; // axis
    const bool valid = (lane_on && group * IPW + il < nreg) || orphan;
    // (lanes without an instance compute on a copy of another one)
    const int slot = orphan ? nreg + orph_e : (valid ? group * IPW + il : (group * IPW < nreg ? group * IPW : 0));
    const int inst = LIST ? (list_n > 0 ? (P.axis_list_in[slot] & 0x7fffffff) : 0) : slot;
    const int NH = EXACT ? NMAX : P.N, rpa = P.axis_rpa;
    const double vsmall = P.vsmall, thr = -axis_pred(vsmall);
    double* const lds = lds_base();
    int oBnd_, oRC_, rcs_;
    (void)axis_lds_doubles(NX, NU, NH, rpa, QMAX, oBnd_, oRC_, rcs_);
    if (!LIST && P.axis_count2 && group == 0 && lane == 0) *P.axis_count2 = 0; // (the second chance's own list: empty before that launch appends to it)
    const int RCS = rcs_;
    int oh_, oHN_, ohN_, oRows_;
    axis_tab_offsets(NXA, oh_, oHN_, ohN_, oRows_);
    const int oh = oh_, oHN = oHN_, ohN = ohN_, oRows = oRows_;
    const int TA = axis_tab_doubles(NXA, NH, rpa);
    // where the states of this lane's axis sit in the system's state vector (FusedPlan::axis_order): state i of axis c at c + NU i (x = (p, v), what
    // the benchmark's CoM model uses) or at c NXA + i (x = (p_x, v_x, p_y, v_y, ..)) -- z0 + zs i either way
    const bool zord = P.axis_order != 0;
    const int zs = zord ? 1 : NU, z0 = zord ? c * NXA : c;
    bool own_refs = false;
    for (int t = 0; t < P.ncost; ++t) own_refs = own_refs || P.cost_p[t] != nullptr;
    own_refs = own_refs && P.axis_cref >= 0;
    // REFERENCE TRAJECTORIES (FusedPlan::stage_refs, CostTerm::pstride: a TrajectoryCost given as a full-size entry whose reference changes along
    // the horizon -- the only form the reference's API has for it, costFunctions.cpp:63-82 with AutoSpan): h differs from stage to stage.  The
    // builds with a run-time horizon rebuild it stage by stage into the lane's (still idle) sparse array before the sweep.
    const bool srefs = !EXACT && P.stage_refs != 0 && P.axis_cref >= 0;
    if (!LIST && P.lane_zero && group == 0 && lane == 0) P.lane_zero[0] = P.lane_zero[2] = 0; // (the NEXT solve's counters: nobody reads them now)
    long long stamp[6];
    stamp[0] = P.prof ? cycle_counter() : 0;

    // ---- 0. this lane's system; the tables of its axis ----
    // One wave per SIMD: nothing overlaps a wave's trips to memory, and the waves of a round start together -- their 34 MB of systems are one
    // burst the SIMDs sit through (measured: 20 k ticks per wave alone on its CU, 37 k with every SIMD busy).  So a wave also TOUCHES the systems
    // of the wave that will follow it on the chip (FusedPlan::axis_pf waves further on: one load per 64 bytes, summed up and looked at once,
    // at the very end): that wave finds them in the L2 / memory-side cache, and the touch travels while this one computes.
    constexpr int LA = (IPW * NX * NX + 7) / 8, LB = (IPW * NX * NU + 7) / 8, LV = (IPW * NX + 7) / 8; // 64-byte pieces of the wave's arrays
    constexpr int NPF = (LA + kWave - 1) / kWave + (LB + kWave - 1) / kWave + 2 * ((LV + kWave - 1) / kWave);
    double pft[NPF];
#pragma unroll
    for (int j = 0; j < NPF; ++j) pft[j] = 0.0;
    if (!CT) { // the tables and bounds of every axis: read stage by stage from LDS
        const double* const src = P.params + P.axis_tab;
        for (int e = lane; e < NU * TA; e += kWave) lds[e] = src[e];
        for (int e = lane; e < NU * NH; e += kWave) { // bounds: [axis][ub (N) | lb (N)]
            const int k = e / NU, a = e - k * NU;
            lds[oBnd_ + a * 2 * NH + k] = P.ub[e];
            lds[oBnd_ + a * 2 * NH + NH + k] = P.lb[e];
        }
    }
    // (the recursions below run inside the iteration's loop and read the tables stage by stage: behind an opaque copy of the OFFSET the
    //  reads stay where they are -- hoisted out of the loop the tables of all stages would sit in registers, 150 doubles of them -- and stay
    //  LDS reads: an opaque POINTER would lose its address space)
    auto opaque = [](int v) __attribute__((always_inline)) -> int {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(v));
#endif
        return v;
    };
    const int oT = c * TA, oUB = oBnd_ + c * 2 * NH; // this lane's axis in LDS (!CT): tables | bounds (ub_k at [k], lb_k at [NH + k])
    const double* const Tg = P.params + P.axis_tab + c * TA; // ... and in memory: what is read once
    // Per-instance references / reference trajectories (below, 1.): the cost rows that look at this axis -- at most kAxisMaxRef, FusedPlan::axis_cref --,
    // where their references lie, their coefficients.  Requested HERE, in front of the systems, and the references themselves right behind the
    // systems: three dependent trips to memory (entry -> cost -> reference) in front of the sweep cost a wave 8 us -- 0.120 instead of 0.086 ms per
    // 65 536 solves with a goal per instance.
    constexpr int MR = kAxisMaxRef, EW = 5 + NZ + NXA;
    const double* bj[MR];
    int psj[MR], lastj[MR];
    double ch[MR][NZ], cN[MR][NXA], pl[MR];
#pragma unroll
    for (int j = 0; j < MR; ++j) {
        bj[j] = P.params;
        psj[j] = lastj[j] = 0;
        pl[j] = 0.0;
#pragma unroll
        for (int a = 0; a < NZ; ++a) ch[j][a] = 0.0;
#pragma unroll
        for (int i = 0; i < NXA; ++i) cN[j][i] = 0.0;
    }
    if (own_refs || srefs) {
        const double* const cf = P.params + P.axis_cref + (size_t)c * (1 + MR * EW);
        const int nref = (int)cf[0];
#pragma unroll
        for (int j = 0; j < MR; ++j) {
            const double* const e = cf + 1 + j * EW;
            const bool on = j < nref;
            const int t = on ? (int)e[0] : -1, r = on ? (int)e[1] : 0, prows = on ? (int)e[2] : 0, ps = on ? (int)e[3] : 0, offP = on ? (int)e[4] : 0;
#pragma unroll
            for (int a = 0; a < NZ; ++a) ch[j][a] = on ? e[5 + a] : 0.0;
#pragma unroll
            for (int i = 0; i < NXA; ++i) cN[j][i] = on ? e[5 + NZ + i] : 0.0;
            const double* own = nullptr; // (this cost's per-instance references, if it has them: the pointers are the launch's arguments)
#pragma unroll
            for (int tt = 0; tt < kMaxCosts; ++tt) own = (t == tt) ? P.cost_p[tt] : own;
            bj[j] = (own ? own + (size_t)inst * prows : P.params + offP) + r;
            psj[j] = ps;
            lastj[j] = ps ? prows / ps - 1 : 0; // (the last step the reference has)
        }
    }
    double A[NXA][NXA], B[NXA], d[NXA], x0[NXA]; // A[i][j]: entry (i, j) of the axis' block
    bool giveup = false; // this lane cannot finish its instance: the first tier solves it from scratch
#if !defined(__HIP_DEVICE_COMPILE__)
    int why = 0; // (emulator, COPRA_EMU_AXIS_REPORT: which test sent it there)
#define AX_WHY(bit, cond) do { if (cond) why |= (bit); } while (0)
#else
#define AX_WHY(bit, cond) do { } while (0)
#endif
    {
        // The columns of A and B that belong to this axis, straight from memory (the three lanes of an instance read neighbouring columns of
        // the same 288 + 144 bytes: one trip, every load requested before the first is used): the axis' own block, and the entries that would
        // couple it to another axis -- those must be zero (each entry of A and B between two axes is looked at by exactly one lane of the instance).
        const double* const sA = P.A + (size_t)inst * NX * NX;
        const double* const sB = P.B + (size_t)inst * NX * NU;
        const double* const sd = P.d + (size_t)inst * NX;
        const double* const sx = P.x0 + (size_t)inst * NX;
        double vA[NXA][NXA][NU], vB[NXA][NU];
#pragma unroll
        for (int jj = 0; jj < NXA; ++jj)
#pragma unroll
            for (int ii = 0; ii < NXA; ++ii)
#pragma unroll
                for (int a = 0; a < NU; ++a) {
                    int ax = c + a; // a == 0: this axis
                    ax = ax >= NU ? ax - NU : ax;
                    vA[jj][ii][a] = sA[((zord ? ax * NXA : ax) + zs * ii) + NX * (z0 + zs * jj)];
                }
#pragma unroll
        for (int ii = 0; ii < NXA; ++ii) {
#pragma unroll
            for (int a = 0; a < NU; ++a) {
                int ax = c + a;
                ax = ax >= NU ? ax - NU : ax;
                vB[ii][a] = sB[((zord ? ax * NXA : ax) + zs * ii) + NX * c];
            }
            d[ii] = sd[z0 + zs * ii];
            x0[ii] = sx[z0 + zs * ii];
        }
        if (own_refs || srefs) { // (requested behind the systems, looked at in front of the sweep)
#pragma unroll
            for (int j = 0; j < MR; ++j) pl[j] = bj[j][lastj[j] * psj[j]];
        }
        double stray = 0.0;
#pragma unroll
        for (int jj = 0; jj < NXA; ++jj)
#pragma unroll
            for (int ii = 0; ii < NXA; ++ii) {
                A[ii][jj] = vA[jj][ii][0];
#pragma unroll
                for (int a = 1; a < NU; ++a) stray += fabs(vA[jj][ii][a]);
            }
#pragma unroll
        for (int ii = 0; ii < NXA; ++ii) {
            B[ii] = vB[ii][0];
#pragma unroll
            for (int a = 1; a < NU; ++a) stray += fabs(vB[ii][a]);
        }
        giveup = !(stray == 0.0); // (a NaN couples)
    }
    // CT: the one row set and the one pair of bounds of this axis
    double ubc = 0.0, lbc = 0.0, re[RPA][NXA], rf[RPA];
    int ridx0[RPA], ridxd[RPA]; // index of row j in the stacked order: ridx0 + step x ridxd  (-1: the slot is empty)
#pragma unroll
    for (int j = 0; j < RPA; ++j) {
        rf[j] = 0.0;
        ridx0[j] = -1;
        ridxd[j] = 0;
#pragma unroll
        for (int i = 0; i < NXA; ++i) re[j][i] = 0.0;
    }
    if (CT) {
        ubc = P.ub[c];
        lbc = P.lb[c];
#pragma unroll
        for (int j = 0; j < RPA; ++j) {
            if (j < rpa) {
                const double* const r0 = Tg + oRows + j * RW;
                const double* const r1 = Tg + oRows + (rpa + j) * RW;
#pragma unroll
                for (int i = 0; i < NXA; ++i) re[j][i] = r0[i];
                rf[j] = r0[NXA + 1];
                ridx0[j] = (int)r0[NXA + 2];
                ridxd[j] = (int)r1[NXA + 2] - ridx0[j];
            }
        }
        // Per-instance limits (copra_batch_set_control_bounds, copra_batch_set_constraint_rhs: every robot its own actuator and velocity limits):
        // this lane's own values where they are the same at every step of the horizon -- what these builds keep in registers --; an instance
        // whose limits change along the horizon goes to the tier.
        if (P.ub_inst) {
            const double* const up = P.ub_inst + (size_t)inst * P.n + c;
            const double* const lp = P.lb_inst + (size_t)inst * P.n + c;
            ubc = up[0];
            lbc = lp[0];
            bool same = true;
            for (int k = 1; k < NH; ++k) same = same & (up[k * NU] == ubc) & (lp[k * NU] == lbc);
            giveup = giveup | !same;
        }
        if (P.row_f_inst) {
            const double* const fp = P.row_f_inst + (size_t)inst * P.mgen;
#pragma unroll
            for (int j = 0; j < RPA; ++j) {
                if (j < rpa && ridx0[j] >= 0) {
                    const double f0 = fp[ridx0[j]];
                    bool same = true;
                    for (int k = 1; k <= NH; ++k) same = same & (fp[ridx0[j] + k * ridxd[j]] == f0);
                    rf[j] = f0;
                    giveup = giveup | !same;
                }
            }
        }
    }
    wave_sync(); // (!CT: the tables are in LDS)
    // this lane's sparse array: [0, NH) the controls, NH + k rpa + j row j of step k, the last entry a spare (what empty slots point to)
    double* const RC = lds + oRC_ + lane * RCS;
    const int posSpare = NH + (NH + 1) * rpa;
    if (!srefs)
        for (int e = 0; e <= posSpare; ++e) RC[e] = 0.0;
    stamp[1] = P.prof ? cycle_counter() : 0;

    // ---- 1. backward Riccati sweep of the chain: K_k, 1 / M_uu,k in registers, kv_k parked in U[k] ----
    double K[NMAX][NXA], MI[NMAX], U[NMAX], Tv[NMAX];
    bool bad = false;
    {
        double H[NZ][NZ], h[NZ]; // (upper triangle used)
#pragma unroll
        for (int a = 0; a < NZ; ++a) {
#pragma unroll
            for (int b = 0; b < NZ; ++b) H[a][b] = Tg[a + NZ * b];
            h[a] = Tg[oh + a];
        }
        double Pm[NXA][NXA], pv[NXA]; // cost-to-go (symmetric: both halves kept, NXA <= 3)
#pragma unroll
        for (int i = 0; i < NXA; ++i) {
#pragma unroll
            for (int j = 0; j < NXA; ++j) Pm[i][j] = Tg[oHN + i + NXA * j];
            pv[i] = Tg[ohN + i];
        }
        // Per-instance cost references (copra_batch_set_cost_reference: every instance tracks its own goal -- one TrajectoryCost(M, p_b) per
        // LMPC in the reference, costFunctions.cpp:63-82): the affine terms h = -sum_t [M N]_t' W_t p_t and hN of this lane's axis are rebuilt
        // from the plan builder's coefficients (FusedPlan::axis_cref; lmpc_lane.hpp does the same for the whole instance), with this instance's
        // references where a cost has them and the controller-wide ones elsewhere.  (A row of another axis has zero coefficients here.)
        // Reference trajectories: hN takes the reference of the last step, h of stage k the reference of step k -- NZ doubles per stage, parked
        // in the lane's sparse array until the sweep has read them (a cost whose reference has no step k -- the last step of a reference over
        // N steps -- takes its last).
        if (own_refs || srefs) {
#pragma unroll
            for (int a = 0; a < NZ; ++a) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < MR; ++j) s += ch[j][a] * pl[j];
                h[a] = s;
            }
#pragma unroll
            for (int i = 0; i < NXA; ++i) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < MR; ++j) s += cN[j][i] * pl[j];
                pv[i] = s;
            }
            if (srefs) { // four stages a turn: their references are requested together
                constexpr int KC = 4;
                for (int k0 = 0; k0 < NH; k0 += KC) {
                    double pk[KC][MR];
#pragma unroll
                    for (int q = 0; q < KC; ++q) {
                        const int k = k0 + q < NH ? k0 + q : NH - 1;
#pragma unroll
                        for (int j = 0; j < MR; ++j) pk[q][j] = bj[j][(k < lastj[j] ? k : lastj[j]) * psj[j]];
                    }
#pragma unroll
                    for (int q = 0; q < KC; ++q) {
                        if (k0 + q < NH) {
#pragma unroll
                            for (int a = 0; a < NZ; ++a) {
                                double s = 0.0;
#pragma unroll
                                for (int j = 0; j < MR; ++j) s += ch[j][a] * pk[q][j];
                                RC[(k0 + q) * NZ + a] = s;
                            }
                        }
                    }
                }
            }
        }
        auto AB = [&](int l, int a) __attribute__((always_inline)) -> double { return a < NXA ? A[l][a] : B[l]; };
        sched_fence(); // (everything this wave reads for ITSELF has been requested: memory operations return in order -- the touches go last)
        {
            // (no branch around the loads: where a branch with a load on one side joins, everything in flight is waited for.  Without a wave to
            //  touch for, a wave touches its own systems again.)
            const int gp0 = group + P.axis_pf, gp = (!LIST && P.axis_pf > 0 && gp0 < nwaves) ? gp0 : (LIST ? 0 : group);
            const size_t i0 = (size_t)gp * IPW;
            const int ni = (int)i0 + IPW <= P.batch ? IPW : (P.batch > (int)i0 ? P.batch - (int)i0 : 1);
            const double* const pa = P.A + i0 * NX * NX;
            const double* const pb = P.B + i0 * NX * NU;
            const double* const pd = P.d + i0 * NX;
            const double* const px = P.x0 + i0 * NX;
            int nt = 0;
#pragma unroll
            for (int j = 0; j < (LA + kWave - 1) / kWave; ++j) {
                const int e = 8 * (j * kWave + lane);
                pft[nt++] = pa[e < ni * NX * NX ? e : 0];
            }
#pragma unroll
            for (int j = 0; j < (LB + kWave - 1) / kWave; ++j) {
                const int e = 8 * (j * kWave + lane);
                pft[nt++] = pb[e < ni * NX * NU ? e : 0];
            }
#pragma unroll
            for (int j = 0; j < (LV + kWave - 1) / kWave; ++j) {
                const int e = 8 * (j * kWave + lane);
                pft[nt++] = pd[e < ni * NX ? e : 0];
                pft[nt++] = px[e < ni * NX ? e : 0];
            }
        }
        sched_fence();
#pragma unroll
        for (int k = NMAX - 1; k >= 0; --k) {
            if (EXACT || k < NH) {
                double tq[NXA], mz[NZ], M[NZ][NZ];
#pragma unroll
                for (int l = 0; l < NXA; ++l) {
                    double s = pv[l];
#pragma unroll
                    for (int i = 0; i < NXA; ++i) s += Pm[l][i] * d[i];
                    tq[l] = s;
                }
                double hk[NZ];
#pragma unroll
                for (int a = 0; a < NZ; ++a) hk[a] = h[a];
                if (!EXACT && srefs) {
                    const int o = opaque(k * NZ);
#pragma unroll
                    for (int a = 0; a < NZ; ++a) hk[a] = RC[o + a];
                }
#pragma unroll
                for (int a = 0; a < NZ; ++a) {
                    double s = hk[a];
#pragma unroll
                    for (int l = 0; l < NXA; ++l) s += AB(l, a) * tq[l];
                    mz[a] = s;
                }
#pragma unroll
                for (int b = 0; b < NZ; ++b) {
                    double Tb[NXA]; // column b of P+ [A B]
#pragma unroll
                    for (int l = 0; l < NXA; ++l) {
                        double s = 0.0;
#pragma unroll
                        for (int i = 0; i < NXA; ++i) s += Pm[l][i] * AB(i, b);
                        Tb[l] = s;
                    }
#pragma unroll
                    for (int a = 0; a <= b; ++a) {
                        double s = H[a][b];
#pragma unroll
                        for (int l = 0; l < NXA; ++l) s += AB(l, a) * Tb[l];
                        M[a][b] = s;
                    }
                }
                const double muu = M[NXA][NXA];
                bad = bad | !(muu > 0.0);
                const double mi = ric_rcp(muu); // (v_rcp_f64 + two Newton steps)
                MI[k] = mi;
#pragma unroll
                for (int j = 0; j < NXA; ++j) K[k][j] = -mi * M[j][NXA];
                const double kvk = -mi * mz[NXA];
                U[k] = kvk;
#pragma unroll
                for (int j = 0; j < NXA; ++j)
#pragma unroll
                    for (int i = 0; i <= j; ++i) {
                        const double s = M[i][j] + M[i][NXA] * K[k][j];
                        Pm[i][j] = s;
                        Pm[j][i] = s;
                    }
#pragma unroll
                for (int i = 0; i < NXA; ++i) pv[i] = mz[i] + M[i][NXA] * kvk;
                sched_fence(); // (nothing of the next stage moves up here: in one basic block of NMAX stages the scheduler would start them all at once)
            } else {
                MI[k] = 0.0;
                U[k] = 0.0;
#pragma unroll
                for (int j = 0; j < NXA; ++j) K[k][j] = 0.0;
            }
            Tv[k] = 0.0;
        }
    }
    if (srefs) // (the sweep has read the stages' h from it)
        for (int e = 0; e <= posSpare; ++e) RC[e] = 0.0;
    giveup = giveup | bad;
    stamp[2] = P.prof ? cycle_counter() : 0;

    // ---- the active set of this lane ----
    // slot a < q: where the constraint lives, packed (aloc: position in RC | step << 8 | kind << 13, kind 0: upper bound of u_step, 1: lower
    // bound, 2 + j: row j of the step; n+ = sg x (the row as written), sg = +1 for a lower bound, -1 otherwise), its multiplier.
    // S = N' Q^-1 N, lower triangle, the identity beyond q.
    constexpr int kEmpty = 1 << 16;
    auto loc_pos = [](int loc) __attribute__((always_inline)) -> int { return loc & 0xff; };
    auto loc_step = [](int loc) __attribute__((always_inline)) -> int { return (loc >> 8) & 31; };
    auto loc_kind = [](int loc) __attribute__((always_inline)) -> int { return (loc >> 13) & 3; };
    auto loc_sg = [](int loc) __attribute__((always_inline)) -> double { return ((loc >> 13) & 3) == 1 ? 1.0 : -1.0; };
    int q = 0;
    int aloc[QMAX];
    // (S and the multipliers: this lane's LDS, behind its sparse array -- 66 registers the recursions have better use for; entry (a, b), b <= a, of S
    //  at a (a + 1) / 2 + b)
    constexpr bool BIG = QMAX > 8;
    double* const Sl = RC + posSpare + 1;
    double* const alam = Sl + (BIG ? 0 : QMAX * (QMAX + 1) / 2);
    // (QMAX > 8: not S but its FACTOR is kept -- L below the diagonal, 1 / L(i, i) on it: it grows by a row when a constraint joins and is
    //  brought back to a triangle by plane rotations when one leaves --, and g = N' y and r = S^-1 g live in the lane's LDS too)
    double* const Ll = alam + QMAX;
    double* const gl = Ll + QMAX * (QMAX + 1) / 2;
    double* const rl = gl + QMAX;
#pragma unroll
    for (int a = 0; a < QMAX; ++a) {
        aloc[a] = posSpare | kEmpty;
        alam[a] = 0.0;
#pragma unroll
        for (int b = 0; b <= a; ++b) (BIG ? Ll : Sl)[a * (a + 1) / 2 + b] = (a == b) ? 1.0 : 0.0; // (S, or its factor: the identity's)
    }
    unsigned mact[2 + RPA]; // [kind] bit k: the upper bound | the lower bound of u_k | row j of step k is active
#pragma unroll
    for (int t = 0; t < 2 + RPA; ++t) mact[t] = 0u;
    // the pick: where it lives, its slack, its multiplier so far; qpgen2's counters of this lane
    bool have = false;
    int ploc = posSpare | kEmpty;
    double psl = 0.0, plam = 0.0;
    int it_main = 0, it_drop = 0, nviol = 0;
    const bool count_viol = P.lane_hist != nullptr;

    // the highest STAGE a set of constraints reaches: a bound of u_k stage k, a row of step k stage min(k, NH - 1) (its state part joins the adjoint
    // that enters stage k - 1, its control part sits in stage k); -1: none
    auto top_stage = [&](unsigned mbounds, unsigned mrows) __attribute__((always_inline)) -> int {
        const unsigned m = mbounds | (mrows & ~(1u << NH)) | (((mrows >> NH) & 1u) << (NH - 1));
        return m ? 31 - __builtin_clz(m) : -1;
    };
    auto wave_top = [&](int k) __attribute__((always_inline)) -> int { return (int)wave_max((double)k); };

    // One forward pass over the stages: the roll-out (FIRST: u_k = K_k x_k + kv_k) or the step (U += tt z with z = Q^-1 of the combined normal
    // whose backward recursion left t_k in Tv), then the trajectory of the new iterate and qpgen2's scan on it: the most violated constraint,
    // normalised by its row norm, the lowest index among equals.  Bounds have unit norms: the worst upper and the worst lower bound are kept
    // as plain minima (upper bounds come first in the stacked order: the upper one wins a tie), the worst row by cross-multiplied comparison
    // (s / |a| < s' / |a'|  <=>  s^2 |a'|^2 > s'^2 |a|^2 for negative slacks), and the three meet at the end.
    // -> bfound: something is violated; bloc, bs: where it lives and its slack
    bool bfound = false;
    int bloc = posSpare | kEmpty;
    double bs = 0.0, zz = 0.0;
    auto forward_scan = [&](auto first_tag, double tt) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const double* const T = lds + opaque(oT);
        const double* const UB = lds + opaque(oUB);
        zz = 0.0;
        double sU = thr, sL = thr, sR = 0.0, nR = 1.0; // slack of the worst upper bound | lower bound | row, and that row's squared norm
        int kU = -1, kL = -1, cR = -1, lR = posSpare | kEmpty; // their stages | the row's index in the stacked order and where it lives
        const unsigned mb = mact[0] | mact[1]; // (the twin of an active bound is its negative: never a candidate -- boxes are not empty here, see below)
        double x[NXA], xi[NXA], G[NXA], W[NXA][NXA], nrow[RPA]; // iterate's state | the step's state | A^k B | sum_{t < k} G_t G_t' (row norms; CT: nrow, per row)
#pragma unroll
        for (int i = 0; i < NXA; ++i) {
            x[i] = x0[i];
            xi[i] = 0.0;
            G[i] = B[i];
#pragma unroll
            for (int j = 0; j < NXA; ++j) W[i][j] = 0.0;
        }
#pragma unroll
        for (int j = 0; j < RPA; ++j) nrow[j] = 0.0;
        auto rows_of = [&](int k, double uk, bool with_u) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < RPA; ++j) {
                if (j < rpa) {
                    double e[NXA], gq = 0.0, f, ax = 0.0, n2 = 0.0;
                    int cid;
                    if (CT) {
#pragma unroll
                        for (int i = 0; i < NXA; ++i) e[i] = re[j][i];
                        f = rf[j];
                        cid = ridx0[j] + k * ridxd[j];
                    } else {
                        const double* const rw = T + oRows + (k * rpa + j) * RW;
#pragma unroll
                        for (int i = 0; i < NXA; ++i) e[i] = rw[i];
                        gq = with_u ? rw[NXA] : 0.0;
                        f = rw[NXA + 1];
                        cid = (int)rw[NXA + 2];
                    }
#pragma unroll
                    for (int i = 0; i < NXA; ++i) ax += e[i] * x[i];
                    if (CT) {
                        n2 = nrow[j];
                    } else {
#pragma unroll
                        for (int i = 0; i < NXA; ++i)
#pragma unroll
                            for (int i2 = 0; i2 < NXA; ++i2) n2 += (e[i] * e[i2]) * W[i][i2];
                        ax += gq * uk;
                        n2 += gq * gq;
                    }
                    const double s = f - ax;
                    const bool neg = (s < thr) & (cid >= 0);
                    if (FIRST && count_viol) nviol += neg ? 1 : 0;
                    const bool v = neg & (((mact[2 + j] >> k) & 1u) == 0u);
                    if (wave_any(v)) { // (a violated row is the exception: nothing below runs where no lane has one at this step)
                        const double l = s * s * nR, r = sR * sR * n2;
                        const bool better = v & ((cR < 0) | (l > r) | ((l == r) & (cid < cR)));
                        sR = better ? s : sR;
                        nR = better ? n2 : nR;
                        cR = better ? cid : cR;
                        lR = better ? ((NH + k * rpa + j) | (k << 8) | ((2 + j) << 13)) : lR;
                    }
                }
            }
        };
#pragma unroll
        for (int k = 0; k < NMAX; ++k) {
            if (EXACT || k < NH) {
                double u;
                if (FIRST) {
                    double acc = U[k];
#pragma unroll
                    for (int j = 0; j < NXA; ++j) acc += K[k][j] * x[j];
                    u = acc;
                } else {
                    double z = Tv[k];
#pragma unroll
                    for (int j = 0; j < NXA; ++j) z += K[k][j] * xi[j];
                    zz += z * z;
                    u = U[k] + tt * z;
                    double xn[NXA];
#pragma unroll
                    for (int i = 0; i < NXA; ++i) {
                        double acc = B[i] * z;
#pragma unroll
                        for (int j = 0; j < NXA; ++j) acc += A[i][j] * xi[j];
                        xn[i] = acc;
                    }
#pragma unroll
                    for (int i = 0; i < NXA; ++i) xi[i] = xn[i];
                }
                U[k] = u;
                rows_of(k, u, true);
                {
                    const double su = (CT ? ubc : UB[k]) - u, sl = u - (CT ? lbc : UB[NH + k]);
                    const bool free_k = ((mb >> k) & 1u) == 0u;
                    if (FIRST && count_viol) nviol += ((su < thr) ? 1 : 0) + ((sl < thr) ? 1 : 0);
                    const bool tu = free_k & (su < sU), tl = free_k & (sl < sL);
                    if (wave_any(tu | tl)) {
                        sU = tu ? su : sU;
                        kU = tu ? k : kU;
                        sL = tl ? sl : sL;
                        kL = tl ? k : kL;
                    }
                }
                double xn[NXA], gn[NXA];
#pragma unroll
                for (int i = 0; i < NXA; ++i) {
                    double acc = d[i] + B[i] * u, ag = 0.0;
#pragma unroll
                    for (int j = 0; j < NXA; ++j) {
                        acc += A[i][j] * x[j];
                        ag += A[i][j] * G[j];
                    }
                    xn[i] = acc;
                    gn[i] = ag;
                }
                if (CT) {
#pragma unroll
                    for (int j = 0; j < RPA; ++j) {
                        if (j < rpa) {
                            double pg = 0.0;
#pragma unroll
                            for (int i = 0; i < NXA; ++i) pg += re[j][i] * G[i];
                            nrow[j] += pg * pg;
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < NXA; ++i) {
#pragma unroll
                        for (int j = 0; j < NXA; ++j) W[i][j] += G[i] * G[j];
                    }
                }
#pragma unroll
                for (int i = 0; i < NXA; ++i) {
                    x[i] = xn[i];
                    G[i] = gn[i];
                }
                sched_fence();
            }
        }
        rows_of(NH, 0.0, false);
        // the worst bound (the upper one among equals), then against the worst row (rows come first in the stacked order: the row among equals)
        const bool lower = (kL >= 0) & ((kU < 0) | (sL < sU));
        const bool bnd = (kU >= 0) | (kL >= 0);
        const double sB = lower ? sL : sU;
        const int kB = lower ? kL : kU;
        const bool row = (cR >= 0) & (!bnd | (sR * sR >= sB * sB * nR));
        bfound = bnd | (cR >= 0);
        bs = row ? sR : sB;
        const int kBs = kB < 0 ? 0 : kB; // (no bound found: the value is not looked at -- and no shift of a negative number)
        bloc = row ? lR : (kBs | (kBs << 8) | (lower ? (1 << 13) : 0));
    };
    // backward recursion of y = Q^-1 n for the combined normal whose coefficients sit in RC, from stage ktop down: t_k = s_k / M_uu,k into Tv (zero
    // above ktop), returns n' Q^-1 n.  rows_live: some lane of the wave has a row in its combination (else the rows' coefficients are not even read)
    auto backward = [&](bool rows_live, int ktop, bool clear_above) __attribute__((always_inline)) -> double {
        const double* const T = lds + opaque(oT);
        double mu[NXA], nqn = 0.0;
#pragma unroll
        for (int i = 0; i < NXA; ++i) mu[i] = 0.0;
#pragma unroll
        for (int k = NMAX - 1; k >= 0; --k) {
            if (k <= ktop) {
                double nu = RC[k];
                if (rows_live) {
#pragma unroll
                    for (int j = 0; j < RPA; ++j) {
                        if (j < rpa) {
                            const double c1 = RC[NH + (k + 1) * rpa + j];
                            if (CT) {
#pragma unroll
                                for (int i = 0; i < NXA; ++i) mu[i] += c1 * re[j][i];
                            } else {
                                const double c0 = RC[NH + k * rpa + j];
                                const double* const r1 = T + oRows + ((k + 1) * rpa + j) * RW;
                                const double* const r0 = T + oRows + (k * rpa + j) * RW;
#pragma unroll
                                for (int i = 0; i < NXA; ++i) mu[i] += c1 * r1[i]; // the state part of a row at step k + 1 joins the adjoint here
                                nu += c0 * r0[NXA]; // the control part of a row at step k
                            }
                        }
                    }
                }
                double s = nu;
#pragma unroll
                for (int i = 0; i < NXA; ++i) s += B[i] * mu[i];
                const double t = s * MI[k];
                Tv[k] = t;
                nqn += s * t;
                double mn[NXA];
#pragma unroll
                for (int i = 0; i < NXA; ++i) { // mu <- Acl' mu + K' nu  =  A' mu + K' s
                    double acc = K[k][i] * s;
#pragma unroll
                    for (int j = 0; j < NXA; ++j) acc += A[j][i] * mu[j];
                    mn[i] = acc;
                }
#pragma unroll
                for (int i = 0; i < NXA; ++i) mu[i] = mn[i];
                sched_fence();
            } else if (clear_above) {
                Tv[k] = 0.0;
            }
        }
        return nqn;
    };
    // forward recursion of y = Q^-1 n+ up to stage ktop: the responses n_i' y of the ACTIVE constraints into RC (every other entry keeps its zero)
    auto forward_resp = [&](bool rows_live, int ktop) __attribute__((always_inline)) {
        const double* const T = lds + opaque(oT);
        double xi[NXA];
#pragma unroll
        for (int i = 0; i < NXA; ++i) xi[i] = 0.0;
        const unsigned mbd = mact[0] | mact[1];
#pragma unroll
        for (int k = 0; k < NMAX; ++k) {
            if (k <= ktop) {
                double y = Tv[k];
#pragma unroll
                for (int j = 0; j < NXA; ++j) y += K[k][j] * xi[j];
                if ((mbd >> k) & 1u) RC[k] = y;
                if (rows_live) {
#pragma unroll
                    for (int j = 0; j < RPA; ++j) {
                        if (j < rpa && ((mact[2 + j] >> k) & 1u)) {
                            double acc = 0.0;
                            if (CT) {
#pragma unroll
                                for (int i = 0; i < NXA; ++i) acc += re[j][i] * xi[i];
                            } else {
                                const double* const rw = T + oRows + (k * rpa + j) * RW;
                                acc = rw[NXA] * y;
#pragma unroll
                                for (int i = 0; i < NXA; ++i) acc += rw[i] * xi[i];
                            }
                            RC[NH + k * rpa + j] = acc;
                        }
                    }
                }
                double xn[NXA];
#pragma unroll
                for (int i = 0; i < NXA; ++i) {
                    double acc = B[i] * y;
#pragma unroll
                    for (int j = 0; j < NXA; ++j) acc += A[i][j] * xi[j];
                    xn[i] = acc;
                }
#pragma unroll
                for (int i = 0; i < NXA; ++i) xi[i] = xn[i];
                sched_fence();
            }
        }
        if (rows_live && ktop >= NH - 1) {
#pragma unroll
            for (int j = 0; j < RPA; ++j) {
                if (j < rpa && ((mact[2 + j] >> NH) & 1u)) {
                    double acc = 0.0;
                    if (CT) {
#pragma unroll
                        for (int i = 0; i < NXA; ++i) acc += re[j][i] * xi[i];
                    } else {
                        const double* const rw = T + oRows + (NH * rpa + j) * RW;
#pragma unroll
                        for (int i = 0; i < NXA; ++i) acc += rw[i] * xi[i];
                    }
                    RC[NH + NH * rpa + j] = acc;
                }
            }
        }
    };
    // the scan's result becomes this lane's pick (qpgen2: step 1)
    auto take_pick = [&]() __attribute__((always_inline)) {
        AX_WHY(1, it_main >= COPRA_AXIS_ITER_CAP);
        giveup = giveup | (it_main >= COPRA_AXIS_ITER_CAP);
        it_main += 1;
        have = bfound & !giveup;
        ploc = have ? bloc : (posSpare | kEmpty);
        psl = bs;
        plam = 0.0;
    };
    // An empty box (or a pinned control, ub = lb: gi_core.hpp, `pinned`) makes the twin of an active bound a candidate of qpgen2's scan; the scan
    // above never looks at twins: such a lane leaves its instance to the tier at once.
    // (an infinite bound: never empty -- inf > 1e-9 inf would say otherwise)
    if (CT) {
        giveup = giveup | !(ubc - lbc > 1e-9 * fmax(1.0, fmin(fabs(ubc), 1e300)));
    } else {
        const double* const UB = lds + oUB;
        for (int k = 0; k < NH; ++k) giveup = giveup | !(UB[k] - UB[NH + k] > 1e-9 * fmax(1.0, fmin(fabs(UB[k]), 1e300)));
    }

    // ---- 2. roll-out: the unconstrained minimiser and qpgen2's first scan ----
    forward_scan(std::true_type {}, 0.0);
    take_pick();
    const int nviol0 = nviol;
    stamp[3] = P.prof ? cycle_counter() : 0;

    // ---- 3. the active-set iteration: every lane that has a pick takes one step per trip ----
    while (wave_any(have)) {
        const unsigned mrows_act = mact[2] | mact[2 + RPA - 1];
        const bool rows_live = wave_any(have & ((loc_kind(ploc) >= 2) | (mrows_act != 0u)));
        // the stages the recursions have to visit: up to the highest one the active set reaches (responses), ... or the pick (its normal)
        const int pk_ = loc_kind(ploc);
        const unsigned pbit = have ? (1u << loc_step(ploc)) : 0u;
        const int ka = wave_top(have ? top_stage(mact[0] | mact[1], mrows_act) : -1);
        const int kp = wave_top(have ? top_stage((mact[0] | mact[1]) | (pk_ < 2 ? pbit : 0u), mrows_act | (pk_ >= 2 ? pbit : 0u)) : -1);
        // y = Q^-1 n+ ;  g = N' y ;  n+' Q^-1 n+
        const double psg = loc_sg(ploc);
        RC[loc_pos(ploc)] = have ? psg : 0.0; // (lanes without a pick: the spare entry)
        const double nqn = backward(rows_live, kp, true);
        if (ka >= 0) forward_resp(rows_live, ka);
        double g[BIG ? 1 : QMAX], r[BIG ? 1 : QMAX];
#pragma unroll
        for (int a = 0; a < (BIG ? 1 : QMAX); ++a) g[a] = r[a] = 0.0;
#define AX_G(a) (BIG ? gl[a] : g[BIG ? 0 : (a)])
#define AX_R(a) (BIG ? rl[a] : r[BIG ? 0 : (a)])
        double wbig[BIG ? QMAX : 1]; // (QMAX > 8) w = L^-1 g: the row the factor grows by when the pick joins
#pragma unroll
        for (int a = 0; a < (BIG ? QMAX : 1); ++a) wbig[a] = 0.0;
        if constexpr (BIG) {
#pragma unroll
            for (int a = 0; a < QMAX; ++a) gl[a] = rl[a] = 0.0;
        }
        if (ka >= 0) { // (no active constraint among the lanes that have a pick -- the first trip of most waves: r = 0)
            if constexpr (BIG) {
#pragma unroll
                for (int a = 0; a < QMAX; ++a) {
                    const double v = RC[loc_pos(aloc[a])];
                    gl[a] = (a < q) ? loc_sg(aloc[a]) * v : 0.0;
                }
                // r = S^-1 g from the factor S = L L' the lane KEEPS (L below the diagonal, 1 / L(i, i) on it, the identity beyond q): two
                // substitutions over the whole padded triangle -- every address a constant, the reads travel together.  (The factor grows by
                // a row when a constraint joins -- the row IS w = L^-1 g, see below -- and is made again from S when one leaves.)
                const int qw = wave_top(have ? q : 0); // (the largest active set among the wave's lanes: beyond it every row is the identity's)
#pragma unroll
                for (int i = 0; i < QMAX; ++i) {
                    if (i < qw) {
                        double sacc = gl[i];
#pragma unroll
                        for (int t = 0; t < i; ++t) sacc -= Ll[i * (i + 1) / 2 + t] * wbig[t];
                        wbig[i] = sacc * Ll[i * (i + 1) / 2 + i];
                    }
                }
                {
                    double rr[QMAX];
#pragma unroll
                    for (int i = QMAX - 1; i >= 0; --i) {
                        rr[i] = 0.0;
                        if (i < qw) {
                            double sacc = wbig[i];
#pragma unroll
                            for (int t = i + 1; t < QMAX; ++t) sacc -= Ll[t * (t + 1) / 2 + i] * rr[t];
                            rr[i] = sacc * Ll[i * (i + 1) / 2 + i];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < QMAX; ++i)
                        if (i < qw) rl[i] = rr[i];
                }
            } else {
#pragma unroll
                for (int a = 0; a < QMAX; ++a) {
                    const double v = RC[loc_pos(aloc[a])];
                    g[BIG ? 0 : a] = (a < q) ? loc_sg(aloc[a]) * v : 0.0;
                }
                // r = S^-1 g by Cholesky (S is the identity beyond q, g is zero there: so is r)
                constexpr int QS = BIG ? 1 : QMAX;
                double L[QS][QS], di[QS], w[QS];
#pragma unroll
                for (int i = 0; i < QS; ++i) {
#pragma unroll
                    for (int j = 0; j <= i; ++j) {
                        double sacc = Sl[i * (i + 1) / 2 + j];
#pragma unroll
                        for (int t = 0; t < j; ++t) sacc -= L[i][t] * L[j][t];
                        if (j < i) {
                            L[i][j] = sacc * di[j];
                        } else {
                            AX_WHY(2, have & !(sacc > 0.0));
                            giveup = giveup | (have & !(sacc > 0.0));
                            di[i] = fast_rsqrt(sacc > 0.0 ? sacc : 1.0);
                            L[i][i] = sacc * di[i];
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < QS; ++i) {
                    double sacc = g[i];
#pragma unroll
                    for (int t = 0; t < i; ++t) sacc -= L[i][t] * w[t];
                    w[i] = sacc * di[i];
                }
#pragma unroll
                for (int i = QS - 1; i >= 0; --i) {
                    double sacc = w[i];
#pragma unroll
                    for (int t = i + 1; t < QS; ++t) sacc -= L[t][i] * r[t];
                    r[i] = sacc * di[i];
                }
            }
        }
        double zn = nqn;
#pragma unroll
        for (int a = 0; a < QMAX; ++a) zn -= AX_G(a) * AX_R(a);
        // t1 = min lambda_i / r_i over r_i > 0, the lowest position among equals (cross-multiplied: one division)
        int l1 = -1;
        double lb_ = 0.0, rb_ = 1.0;
#pragma unroll
        for (int a = 0; a < QMAX; ++a) {
            const double ra = AX_R(a);
            const bool ok = (a < q) & (ra > 0.0);
            const double la = alam[a];
            const bool better = ok & ((l1 < 0) | (la * rb_ < lb_ * ra));
            lb_ = better ? la : lb_;
            rb_ = better ? ra : rb_;
            l1 = better ? a : l1;
        }
        const double t1 = lb_ / rb_;
        // t2 = -s / z'n; a direction that is zero (n+ in the span of the active normals: |z|^2 <= vsmall is checked behind the step) or not a
        // descent direction to rounding: the tier's business.  (z'n+ = n+' Q^-1 n+ - g'r loses its digits where a pick lies next to active rows on
        // a chain whose control barely moves it -- 1.4 % of a TIGHT workload's instances on the jerk-controlled model go to the tier for it.  The
        // sum-of-squares form (n+ - N r)' Q^-1 (n+ - N r), which the second backward recursion returns for free, was tried: nothing is listed
        // any more, but 3 % of those instances take one to three picks more than qpgen2 -- it leaves out r'(g - S r), which is not small there.)
        const bool zn_ok = (zn > 0.0) & (zn > COPRA_AXIS_ZN_REL * nqn);
        // (z'n+ zero to rounding: n+ lies in the span of the active normals -- a velocity row behind the bounds of every control in front of it.
        //  qpgen2 then takes no step in primal space but a DUAL one -- the multipliers move by t1, the blocking constraint leaves, the pick stays --
        //  if |z|^2 <= vsmall, which the recursion below tells)
        const bool nostep = have & !zn_ok;
        double tt = zn_ok ? -psl / zn : 0.0;
        const bool full = zn_ok & !((l1 >= 0) & (t1 < tt));
        tt = full ? tt : t1;
        const double tm = (have & !giveup) ? tt : 0.0; // what the multipliers move by
        tt = nostep ? 0.0 : tm; // ... and the iterate
        // z = Q^-1 (n+ - N r): the coefficients over the responses, the recursions again, U += tt z and the next scan
        if (ka >= 0) { // (else: the combined normal IS n+, its recursion has been done)
#pragma unroll
            for (int a = 0; a < QMAX; ++a) RC[loc_pos(aloc[a])] = -AX_R(a) * loc_sg(aloc[a]); // (empty slots: the spare entry)
            (void)backward(rows_live, kp, false);
        }
        forward_scan(std::false_type {}, tt);
        if (ka >= 0) {
#pragma unroll
            for (int a = 0; a < QMAX; ++a) RC[loc_pos(aloc[a])] = 0.0;
        }
        RC[loc_pos(ploc)] = 0.0;
        // qpgen2's test: |z|^2 <= vsmall -- no step in primal space.  Where z'n+ said so too and a multiplier blocks: the dual step.  Without one:
        // "no solution" -- the tier reports it.  Where the two tests disagree: the tier decides.
        AX_WHY(4, have & !nostep & !(zz > vsmall));
        AX_WHY(8, nostep & !(zz <= vsmall));
        AX_WHY(16, nostep & (zz <= vsmall) & (l1 < 0));
        giveup = giveup | (have & !nostep & !(zz > vsmall)) | (nostep & (!(zz <= vsmall) | (l1 < 0)));
        have = have & !giveup;
        if (have) {
#pragma unroll
            for (int a = 0; a < QMAX; ++a) alam[a] -= tm * AX_R(a);
            plam += tm;
            if (full) {
                // the pick joins the active set: slot q, S grows by the row [g' | n+' Q^-1 n+]
                if (q >= QMAX) {
                    AX_WHY(64, true);
                    giveup = true;
                    have = false;
                } else {
#pragma unroll
                    for (int a = 0; a < QMAX; ++a) aloc[a] = (a == q) ? ploc : aloc[a];
                    alam[q] = plam;
                    if constexpr (!BIG) {
                        double* const row = Sl + q * (q + 1) / 2;
#pragma unroll
                        for (int b = 0; b < QMAX; ++b)
                            if (b <= q) row[b] = (b == q) ? nqn : AX_G(b);
                    }
                    if constexpr (BIG) { // the factor's new row: [w' | sqrt(n+' Q^-1 n+ - w'w)], its diagonal kept as the reciprocal
                        double* const lrow = Ll + q * (q + 1) / 2;
                        double dd = nqn;
#pragma unroll
                        for (int b = 0; b < QMAX; ++b) dd -= (b < q) ? wbig[b] * wbig[b] : 0.0;
                        AX_WHY(32, !(dd > 0.0));
                        giveup = giveup | !(dd > 0.0);
#pragma unroll
                        for (int b = 0; b < QMAX; ++b)
                            if (b <= q) lrow[b] = (b == q) ? fast_rsqrt(dd > 0.0 ? dd : 1.0) : wbig[b];
                    }
                    const int pk = loc_kind(ploc);
                    const unsigned bit = 1u << loc_step(ploc);
#pragma unroll
                    for (int t = 0; t < 2 + RPA; ++t) mact[t] |= (t == pk) ? bit : 0u;
                    q += 1;
                    take_pick(); // (the scan that came with the step)
                }
            } else {
                // partial step: the blocking constraint l1 leaves the active set, the pick stays (its slack moved with the step)
                int dloc = 0;
#pragma unroll
                for (int a = 0; a < QMAX; ++a) dloc = (a == l1) ? aloc[a] : dloc;
                const int dk = loc_kind(dloc);
                const unsigned bit = 1u << loc_step(dloc);
#pragma unroll
                for (int t = 0; t < 2 + RPA; ++t) mact[t] &= (t == dk) ? ~bit : ~0u;
#pragma unroll
                for (int a = 0; a < QMAX; ++a) {
                    const bool sh = a >= l1; // slot a takes slot a + 1
                    const int nl = (a + 1 < QMAX) ? aloc[a + 1 < QMAX ? a + 1 : a] : (posSpare | kEmpty);
                    aloc[a] = sh ? nl : aloc[a];
                }
                if constexpr (BIG) {
                    // the factor without row l1 of L: the rows behind it move up, each with one entry beyond its diagonal -- plane rotations of
                    // the columns (j, j + 1), j = l1 .. q - 2, bring the triangle back (what qpgen2 does to its R when a constraint leaves);
                    // in place, in the lane's LDS
                    for (int j = l1; j < q - 1; ++j) {
                        alam[j] = alam[j + 1];
                        double* const src = Ll + (j + 1) * (j + 2) / 2; // the row that becomes row j: entries 0 .. j + 1
                        const double x = src[j], y = 1.0 / src[j + 1];
                        const double rinv = fast_rsqrt(x * x + y * y), cg = x * rinv, sg2 = y * rinv;
                        for (int i = j + 2; i < q; ++i) { // the rows behind: their entries in the two columns
                            double* const row = Ll + i * (i + 1) / 2;
                            const double xi = row[j], yi = row[j + 1];
                            row[j] = cg * xi + sg2 * yi;
                            row[j + 1] = cg * yi - sg2 * xi;
                        }
                        double* const dst = Ll + j * (j + 1) / 2;
                        for (int t = 0; t < j; ++t) dst[t] = src[t];
                        dst[j] = rinv;
                    }
                    alam[q - 1] = 0.0;
                    for (int b = 0; b < q; ++b) Ll[(q - 1) * q / 2 + b] = (b == q - 1) ? 1.0 : 0.0;
                } else {
                    constexpr int QS = BIG ? 1 : QMAX;
                    double lam_[QS], Sn[QS][QS];
#pragma unroll
                    for (int a = 0; a < QMAX; ++a) lam_[a] = alam[a];
                    // S without row and column l1 (rows and columns behind it move up; the last becomes the identity's)
#pragma unroll
                    for (int a = 0; a < QMAX; ++a)
#pragma unroll
                        for (int b = 0; b <= a; ++b) Sn[a][b] = Sl[a * (a + 1) / 2 + b];
#pragma unroll
                    for (int a = 0; a < QMAX; ++a) {
                        const double nm = (a + 1 < QMAX) ? lam_[a + 1 < QMAX ? a + 1 : a] : 0.0;
                        alam[a] = (a >= l1) ? nm : lam_[a];
#pragma unroll
                        for (int b = 0; b <= a; ++b) {
                            const double same = Sn[a][b];
                            const double down = (a + 1 < QMAX) ? Sn[a + 1 < QMAX ? a + 1 : a][b] : (a == b ? 1.0 : 0.0);
                            const double diag = (a + 1 < QMAX) ? Sn[a + 1 < QMAX ? a + 1 : a][b + 1 < QMAX ? b + 1 : b] : (a == b ? 1.0 : 0.0);
                            Sl[a * (a + 1) / 2 + b] = (b >= l1) ? diag : (a >= l1) ? down : same;
                        }
                    }
                }
                q -= 1;
                it_drop += 1;
                psl += tt * zn;
                if (it_drop + it_main > COPRA_AXIS_ITER_CAP) {
                    giveup = true;
                    have = false;
                }
            }
        }
    }
    stamp[4] = P.prof ? cycle_counter() : 0;

    // ---- 4. the instance: qpgen2's counters are the sums over its axes; one lane of it reports ----
#if !defined(__HIP_DEVICE_COMPILE__)
    if (giveup && valid && std::getenv("COPRA_EMU_AXIS_REPORT"))
        std::fprintf(stderr, "  emu axis solver: instance %d axis %d gives up: reasons %d (1 cap | 2 S not positive | 4 z'n+ > 0 but |z|^2 <= vsmall | 8 z'n+ ~ 0 but |z|^2 > vsmall | 16 dual step without a blocking multiplier | 32 factor row | 64 no room) after %d picks, %d drops, q %d\n",
            inst, c, why, it_main, it_drop, q);
#endif
    int fail_i = (giveup ? 1 : 0) | (bad ? 2 : 0), adds_i = it_main - 1, drops_i = it_drop, viol_i = nviol0;
#pragma unroll
    for (int a = 1; a < NU; ++a) {
        const int src = lane + a < kWave ? lane + a : lane;
        const int f2 = shfl_i32(fail_i, src), a2 = shfl_i32(adds_i, src), d2 = shfl_i32(drops_i, src), v2 = shfl_i32(viol_i, src);
        if (lane_on && c == 0) { // (lane of axis 0: its instance's other axes sit in the next lanes)
            fail_i |= f2;
            adds_i += a2;
            drops_i += d2;
            viol_i += v2;
        }
    }
    bool head = valid && lane_on && c == 0;
    if (SP > 0) {
        // an instance on spare lanes: its axes sit in NU different waves.  Their counters meet in a word of axis_acc -- adds | drops << 12 |
        // give-ups << 24 | failed factorisations << 26 | arrivals << 28 --; the lane that arrives last reports and leaves the word zero for the
        // next solve.
        bool last = false;
        if (orphan) {
            const int mine = adds_i | (drops_i << 12) | ((fail_i & 1) << 24) | (((fail_i >> 1) & 1) << 26) | (1 << 28);
            const int old = atomic_add_i32(P.axis_acc + orph_e, mine);
            last = ((old >> 28) & 3) == NU - 1;
            if (last) {
                const int tot = old + mine;
                P.axis_acc[orph_e] = 0;
                adds_i = tot & 0xfff;
                drops_i = (tot >> 12) & 0xfff;
                fail_i = (((tot >> 24) & 3) ? 1 : 0) | (((tot >> 26) & 3) ? 2 : 0);
            }
        }
        head = head || last;
    }
    const bool more = head && (fail_i & 1) != 0;
    {
        int total = 0;
        const int before = wave_prefix_count(more, total);
        if (total > 0) {
            int base = 0;
            if (lane == 0) base = atomic_add_i32(P.lane_count, total);
            base = bcast_i32(base, 0);
            if (more) P.lane_list[base + before] = (fail_i & 2) ? (inst | (int)0x80000000) : inst; // (top bit: a factorisation failed -- status 2)
        }
        int nsteps = 0;
        (void)wave_prefix_count(head && !more && adds_i > 0, nsteps); // instances the iteration finished here
        if (nsteps > 0 && lane == 0) (void)atomic_add_i32(P.lane_count + 2, nsteps);
    }
    if (head && !more) {
        P.status[inst] = 0;
        P.iter[2 * (size_t)inst] = 1 + adds_i; // (qpgen2's counters: the scan that found nothing counts)
        P.iter[2 * (size_t)inst + 1] = drops_i;
    }
    if (P.lane_hist) { // (first solve of a controller: the violated-row counts of the instances left over, lmpc_lane.hpp)
        const int bin = !more ? -1 : (viol_i < kLaneHistBins - 1 ? viol_i : kLaneHistBins - 1);
        for (int b = 0; b < kLaneHistBins; ++b) {
            int cnt = 0;
            (void)wave_prefix_count(bin == b, cnt);
            if (cnt > 0 && lane == 0) (void)atomic_add_i32(P.lane_hist + b, cnt);
        }
    }

    // ---- 5. results: U and the trajectory of the last iterate.  The lanes' arrays are dead: their place takes the results of the wave's
    //         instances exactly as they lie in memory -- [instance][X], then [instance][n] -- and leaves as one flat copy, 16 bytes per lane and
    //         store.  (Straight from the lanes -- 24 contiguous bytes per instance and store -- the stores were the phase that grew with the
    //         load on the chip: 6 k ticks for a wave alone on its CU, 14 - 18 k with every SIMD busy; profiles/r06/axis_load_sweep.txt.)  An
    //         instance on a spare lane has its other axes elsewhere: that lane stores for itself. ----
    if constexpr (LIST) { // (the wave's instances are scattered over the batch: every lane stores for itself)
        if (valid) {
            double* const xo = P.trajectory + (size_t)inst * P.X + z0;
            double* const uo = P.control + (size_t)inst * P.n + c;
            double x[NXA];
#pragma unroll
            for (int i = 0; i < NXA; ++i) x[i] = x0[i];
#pragma unroll
            for (int k = 0; k < NMAX; ++k) {
                if (EXACT || k < NH) {
                    const double u = U[k];
#pragma unroll
                    for (int i = 0; i < NXA; ++i) xo[k * NX + zs * i] = x[i];
                    uo[k * NU] = u;
                    double xn[NXA];
#pragma unroll
                    for (int i = 0; i < NXA; ++i) {
                        double acc = d[i] + B[i] * u;
#pragma unroll
                        for (int j = 0; j < NXA; ++j) acc += A[i][j] * x[j];
                        xn[i] = acc;
                    }
#pragma unroll
                    for (int i = 0; i < NXA; ++i) x[i] = xn[i];
                }
            }
#pragma unroll
            for (int i = 0; i < NXA; ++i) xo[NH * NX + zs * i] = x[i];
        }
    } else {
        const int nr = nreg - group * IPW < IPW ? (nreg - group * IPW > 0 ? nreg - group * IPW : 0) : IPW; // instances on this wave's regular lanes
        const int XI = P.X, UI = P.n;
        // (chains of three states: the controls take the place of the states once those have left -- axis_lds_doubles)
        constexpr bool TWO = NXA >= 3;
        double* const sx_ = lds + oRC_; // [il][X]
        double* const su_ = TWO ? sx_ : sx_ + IPW * XI; // [il][n]
        double* const so_ = TWO ? sx_ + IPW * XI : su_ + IPW * UI; // the axis of an instance on a spare lane, packed: x_k(i) at k NXA + i, then u_k
        wave_sync(); // (every lane is through with its arrays)
        {
            // (lanes without an instance write like a spare lane: nobody reads it)
            double* const xo = lane_on ? sx_ + il * XI + z0 : so_;
            double* const uo = lane_on ? su_ + il * UI + c : so_ + (NH + 1) * NXA;
            const int sxk = lane_on ? NX : NXA, sxi = lane_on ? zs : 1, suk = lane_on ? NU : 1; // strides: step | state of the axis; step
            double x[NXA];
#pragma unroll
            for (int i = 0; i < NXA; ++i) x[i] = x0[i];
#pragma unroll
            for (int k = 0; k < NMAX; ++k) {
                if (EXACT || k < NH) {
                    const double u = U[k];
#pragma unroll
                    for (int i = 0; i < NXA; ++i) xo[k * sxk + i * sxi] = x[i];
                    if (!TWO || !lane_on) uo[k * suk] = u;
                    double xn[NXA];
#pragma unroll
                    for (int i = 0; i < NXA; ++i) {
                        double acc = d[i] + B[i] * u;
#pragma unroll
                        for (int j = 0; j < NXA; ++j) acc += A[i][j] * x[j];
                        xn[i] = acc;
                    }
#pragma unroll
                    for (int i = 0; i < NXA; ++i) x[i] = xn[i];
                }
            }
#pragma unroll
            for (int i = 0; i < NXA; ++i) xo[NH * sxk + i * sxi] = x[i];
        }
        wave_sync();
        auto flat_out = [&](double* dst, const double* src, int count) __attribute__((always_inline)) {
            const int even = count & ~1;
            for (int e = 2 * lane; e < even; e += 2 * kWave) {
                const double v0 = src[e], v1 = src[e + 1];
#if defined(__HIP_DEVICE_COMPILE__)
                typedef double axis_pair __attribute__((ext_vector_type(2), aligned(8)));
                axis_pair pr;
                pr.x = v0;
                pr.y = v1;
                __builtin_nontemporal_store(pr, (axis_pair*)(dst + e)); // (written once, read by nobody on the chip: + 2 % on the headline)
#else
                dst[e] = v0;
                dst[e + 1] = v1;
#endif
            }
            if ((count & 1) && lane == 0) dst[count - 1] = src[count - 1];
        };
        flat_out(P.trajectory + (size_t)group * IPW * XI, sx_, nr * XI);
        if constexpr (TWO) {
            wave_sync(); // (the states have been read)
            if (lane_on) {
                double* const uo2 = su_ + il * UI + c;
#pragma unroll
                for (int k = 0; k < NMAX; ++k)
                    if (EXACT || k < NH) uo2[k * NU] = U[k];
            }
            wave_sync();
        }
        flat_out(P.control + (size_t)group * IPW * UI, su_, nr * UI);
        if (SP > 0 && wave_any(orphan)) { // the spare lane's axis: its (NH + 1) NXA states and NH controls, one lane each (spare lanes: NU = 3, NH <= 20)
            static_assert(SP == 0 || ((NMAX + 1) * NXA <= kWave && SP == 1), "one store for the states, one for the controls");
            const int oi = bcast_i32(inst, kWave - 1), oc = bcast_i32(c, kWave - 1);
            if (lane < (NH + 1) * NXA) {
                const int k = lane / NXA, i = lane - k * NXA;
                P.trajectory[(size_t)oi * XI + k * NX + (zord ? oc * NXA : oc) + zs * i] = so_[lane];
            }
            if (lane < NH) P.control[(size_t)oi * UI + lane * NU + oc] = so_[(NH + 1) * NXA + lane];
        }
    }
    {
        double pf_sum = 0.0;
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(pft[j])); // (not before: the sum must not start -- and wait -- where the touches were issued)
#endif
            pf_sum += pft[j];
        }
        if (pf_sum == -1.2345678901234567e300) P.status[0] = 5; // (never: the touches above must not be optimised away -- and are waited for HERE)
    }
    if (P.prof && lane == 0) {
        stamp[5] = cycle_counter();
        long long* pr = P.prof + 8 * (size_t)group;
        for (int t = 0; t < 5; ++t) pr[t] = stamp[t + 1] - stamp[t];
        pr[5] = stamp[0]; // (absolute: which waves ran side by side -- tools/exp/axis_phases.py)
#if defined(__HIP_DEVICE_COMPILE__)
        pr[6] = ((long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); // XCC_ID | HW_ID: where it ran
#else
        pr[6] = 0;
#endif
        pr[7] = stamp[5] - stamp[0];
    }
}

} // namespace copra_hip
