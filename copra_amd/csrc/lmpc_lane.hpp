// LMPC::solve() with ONE INSTANCE PER LANE: the pass in front of the Riccati-factor tier (lmpc_fused_ric.hpp).
//
// qpgen2 starts at the unconstrained minimiser -Q^-1 c and is finished at once when no constraint is violated there
// (QuadProgSolver.cpp:45-72 -> qpgen2: the first scan finds nothing).  For a controller whose costs are all per-step entries
// (costFunctions.cpp:63-215) that minimiser is the LQ roll-out u_k = K_k x_k + kv_k of ONE backward Riccati sweep -- N small dense
// steps without any search, the same work for every instance: data-parallel over the BATCH, not over the matrix.  So this pass
// gives every instance one lane: the lane keeps its own A, B, d, the cost-to-go P and the stage matrix M in registers, every
// multiply-add is a full-rate v_fma_f64 on 64 instances at once (no padding of 6 x 6 blocks to MFMA tiles, no cross-lane
// traffic, no LDS hand-overs), the stage records K_k | kv_k | Lam_k^-1 go to a lane-major HBM workspace (coalesced: 512 B per
// store), and the roll-out checks every constraint row and bound with qpgen2's own test (slack <= -vsmall).  An instance that
// violates nothing is DONE: U, X, status 0, iteration counts (1, 0) -- exactly what the first tier reports for it.  Every other
// instance is appended to a list, and the first tier (wave per instance) runs for those only -- and only its active-set iteration:
// it takes the stage records, the row-norm sums, U and X over from here instead of sweeping and rolling out again
// (lmpc_fused_ric.hpp, from_lane).  At the headline workload 47 % of the instances end here; DESIGN.md 3.12 has the measurements.
//     stage k < N:   l_k(x, u) = 1/2 [x; u]' H [x; u] + h' [x; u]          (H, h, HN, hN: plan_builder.hpp, build_lane_tables)
//     stage N:       l_N(x)    = 1/2 x' HN x + hN' x
//     M = H + [A B]' P+ [A B],  m = h + [A B]' (P+ d + p+),  K = -M_uu^-1 M_ux,  kv = -M_uu^-1 m_u,
//     P = M_xx + M_xu K,  p = m_x + M_xu kv                                  (the recursion of lmpc_fused_ric.hpp, dense per lane)
// Shapes: compile-time (NX, NU) with NU <= 3 (the Riccati-factor tier's), the horizon is a run-time value.
#pragma once
#include <type_traits>
#include "lmpc_fused.hpp"

// (experiments only, never defined in the product build: what a phase costs -- tools/exp/lane_variants.sh builds one library per value.
//  1: no result stores; 2: the gains are not read back; 4: no norm recursion; 8: no row checks; 16: no bound checks)
#ifndef COPRA_LANE_EXP
#define COPRA_LANE_EXP 0
#endif

namespace copra_hip {

// W doubles per instance, [batch][W] in HBM: the 64 instances of this wave are one contiguous block -- coalesced loads (ALL arrays
// are requested before the first one is used: one trip to memory), transposed through LDS (odd stride: no bank conflicts) so that
// every lane ends up with its own instance in registers
template <int W>
COPRA_DEV void lane_fetch(const double* src, int group, int batch, double (&raw)[W])
{
    const int lane = lane_id();
    const size_t base = (size_t)group * kWave * W;
    const int left = batch - group * kWave;
    const int count = (left < kWave ? left : kWave) * W;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int e = j * kWave + lane;
        raw[j] = src[base + (e < count ? e : 0)];
    }
}
template <int W>
COPRA_DEV void lane_transpose_in(const double (&raw)[W], int group, int batch, double* lds, double (&out)[W])
{
    constexpr int ST = W | 1;
    const int lane = lane_id();
    const int left = batch - group * kWave;
    const int count = (left < kWave ? left : kWave) * W;
    wave_sync(); // (the previous array has left the staging area)
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int e = j * kWave + lane;
        lds[e + (ST - W) * (e / W)] = raw[j];
    }
    wave_sync();
    const int ln = (lane * W < count) ? lane : 0; // (lanes past the end of the batch compute on a copy of the first instance)
#pragma unroll
    for (int c = 0; c < W; ++c) out[c] = lds[ln * ST + c];
}
// this lane's element of a wave-uniform workspace row, by a 32-bit byte offset from the row pointer.  (The row pointer is made opaque to
// the optimiser: left to itself it folds row and lane offsets into a fresh 64-bit address computation per access and keeps all of them
// live across the unrolled stage.)
// (The accesses go through the GLOBAL address space explicitly: behind the opaque pointer the compiler would fall back to flat loads and
//  stores, which probe the LDS aperture as well and count against both wait counters.)
#if defined(__HIP_DEVICE_COMPILE__)
typedef double __attribute__((address_space(1))) lane_gdouble;
#endif
COPRA_DEV void lane_put(double* row, unsigned byte_off, double v, bool streaming = false)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(row));
    lane_gdouble* const g = (lane_gdouble*)((unsigned long long)row + byte_off);
    if (streaming)
        __builtin_nontemporal_store(v, g);
    else
        *g = v;
#else
    (void)streaming;
    *(double*)((char*)row + byte_off) = v;
#endif
}
COPRA_DEV double lane_get(const double* row, unsigned byte_off)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(row));
    return *(const lane_gdouble*)((unsigned long long)row + byte_off);
#else
    return *(const double*)((const char*)row + byte_off);
#endif
}

// A group of the roll-out's stages leaves the LDS staging area: W doubles per instance (row `il` of the staging area, stride LS) to
// dst + il * stride, consecutive lanes writing consecutive PAIRS of an instance's segment -- one 16-byte store per lane, the (instance, column)
// of a lane's pair advanced by additions (round 5: the store phase was a quarter of the roll-out -- one 8-byte store per lane behind two integer
// divisions, a predicate and a wait for its own LDS read, 36 times per group).  Whole waves and whole groups only; the others go element by element.
template <int W>
COPRA_DEV void lane_group_out(double* dst, size_t stride, const double* stg, int LS, int ninst, int cnt, int lane)
{
    static_assert(W % 2 == 0, "pairs");
    constexpr int HW = W / 2; // pairs per instance
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(lane)); // (opaque per call: hoisted out of the roll-out's loop the addresses of every step would be live across it)
#endif
    if (ninst == kWave && cnt == W) {
        // (column, LDS index and offset in memory of this lane's pair, advanced together)
        const int il0 = lane / HW;
        int c = 2 * (lane - il0 * HW), lo = il0 * LS + c;
        unsigned go = (unsigned)il0 * (unsigned)stride + (unsigned)c; // (64 instances x one instance's results: far below 2^32 doubles)
        const int dl = (kWave / HW) * LS + 2 * (kWave % HW);
        const unsigned dg = (unsigned)(kWave / HW) * (unsigned)stride + 2u * (kWave % HW);
        // NB pairs at a time: their LDS reads first, then their stores (one at a time, every store waited for its own read: 18 LDS latencies per group)
        constexpr int NB = HW % 3 == 0 ? 3 : HW % 2 == 0 ? 2 : 1;
#pragma unroll 1
        for (int j = 0; j < HW; j += NB) { // (64 HW pairs, 64 per step)
            double v0[NB], v1[NB];
            unsigned gq[NB];
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                v0[t] = stg[lo];
                v1[t] = stg[lo + 1];
                gq[t] = go;
                c += 2 * (kWave % HW);
                const bool wrap = c >= W; // (into the next instance's segment)
                c -= wrap ? W : 0;
                lo += dl + (wrap ? LS - W : 0);
                go += dg + (wrap ? (unsigned)stride - (unsigned)W : 0u);
            }
            sched_fence();
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                double* const g = dst + gq[t];
#if defined(__HIP_DEVICE_COMPILE__)
                typedef double lane_pair __attribute__((ext_vector_type(2), aligned(8)));
                typedef lane_pair __attribute__((address_space(1))) lane_gpair;
                lane_pair pr;
                pr.x = v0[t];
                pr.y = v1[t];
                *(lane_gpair*)g = pr;
#else
                g[0] = v0[t];
                g[1] = v1[t];
#endif
            }
        }
        return;
    }
#pragma unroll 1
    for (int j = 0; j < W; ++j) {
        const int e = j * kWave + lane, il = e / W, c = e - il * W;
        if (il < ninst && c < cnt) dst[(size_t)il * stride + c] = stg[il * LS + c];
    }
}

// (streaming: a store that does not claim cache space -- what is written once and read, if at all, by a later kernel; measured: 226 -> 220 us
//  for the pass when Lam^-1 and the norm sums leave this way, no difference for U and X)

// The first steps of the active-set iteration where the picks are bounds on u_0 (see lmpc_lane_body): from the iterate us[0] of stage 0 the
// iterates us[1 .. KS] after one, two ... steps, with everything the verdict needs about each level.  W = M_uu,0^-1; a level without a step
// repeats the iterate before it.
template <int NU, int KS>
COPRA_DEV void lane_spec_steps(const double (&W)[NU][NU], const double (&ubk)[NU], const double (&lbk)[NU], double (&us)[KS + 1][NU], double vsmall,
    bool bad, bool (&specl)[KS + 1], int (&scl)[KS + 1], double (&sst)[KS + 1], double (&sst2)[KS + 1])
{
    constexpr int kSpec = KS;
    double lam1 = 0.0, sig1 = 0.0; // multiplier and orientation of the first active bound
#pragma unroll
    for (int l = 1; l <= kSpec; ++l) {
        // the most violated bound of u_0 at iterate l - 1 that is not active: qpgen2's order -- upper bounds (rows mgen + j) before
        // lower ones (mgen + n + j), the first of equals wins -- so a tie is not decided here: strictly worse than every other
        // one, or no speculation.  (The twin of an active bound cannot be violated: gi_core.hpp, `pinned`.)
        double best = 0.0, second = 0.0, sig = 0.0;
        int cb = -1;
#pragma unroll
        for (int i = 0; i < 2 * NU; ++i) {
            const int c = i < NU ? i : i - NU;
            const double sl = i < NU ? ubk[c] - us[l - 1][c] : us[l - 1][c] - lbk[c];
            const bool cand = sl <= -vsmall && c != scl[l - 1] && (l < 2 || c != scl[l - 2]);
            const bool better = cand && sl < best;
            second = better ? best : ((cand && sl < second) ? sl : second);
            best = better ? sl : best;
            cb = better ? c : cb;
            sig = better ? (i < NU ? -1.0 : 1.0) : sig;
        }
        double ubc = 0.0, lbc = 0.0, wcc = 0.0, w1c = 0.0, w11 = 0.0;
#pragma unroll
        for (int c = 0; c < NU; ++c) {
            ubc = (c == cb) ? ubk[c] : ubc;
            lbc = (c == cb) ? lbk[c] : lbc;
            wcc = (c == cb) ? W[c][c] : wcc;
#pragma unroll
            for (int c1 = 0; c1 < NU; ++c1) {
                w1c = (c == cb && c1 == scl[1]) ? W[c1][c] : w1c;
                w11 = (c1 == scl[1]) ? W[c1][c1] : w11;
            }
        }
        bool go = (l == 1 || specl[l - 1]) && cb >= 0 && best < second && !bad && !(ubc - lbc <= -vsmall); // (an empty box: the tier reports it)
        // the step: z = H n, n = sig e_cb; with one active bound (level 2) H = W - W n1 n1' W / (n1' W n1)
        double z[NU], zn, r1 = 0.0;
        if (l == 1) {
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                double wj = 0.0;
#pragma unroll
                for (int c = 0; c < NU; ++c) wj = (c == cb) ? W[j][c] : wj;
                z[j] = sig * wj;
            }
            zn = wcc; // z'n = |Lam^-1 e_c|^2
        } else {
            r1 = sig1 * sig * w1c / w11; // r = (n1' W n1)^-1 n1' W n
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                double wj = 0.0, w1j = 0.0;
#pragma unroll
                for (int c = 0; c < NU; ++c) {
                    wj = (c == cb) ? W[j][c] : wj;
                    w1j = (c == scl[1]) ? W[j][c] : w1j;
                }
                z[j] = sig * (wj - w1j * (w1c / w11));
            }
            zn = wcc - w1c * (w1c / w11);
        }
        double zz = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) zz += z[j] * z[j];
        go = go && zn > 0.0 && zz > vsmall; // (gi_core.hpp: no step in primal space -- the tier's business)
        const double t2 = go ? -best / zn : 0.0;
        // the dual step length: an active bound whose multiplier would reach zero first is DROPPED by the iteration -- the tier's business
        if (l == 2) go = go && !(r1 > 0.0 && lam1 / r1 <= t2 * (1.0 + 1e-9));
        const double tt = go ? t2 : 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) us[l][j] = go ? us[l - 1][j] + t2 * z[j] : us[l - 1][j]; // (no step: z may be 0 / 0)
        if (l == 1) {
            lam1 = tt;
            sig1 = sig;
        }
        specl[l] = go;
        scl[l] = go ? cb : -1;
        sst[l] = go ? best : 0.0;
        sst2[l] = sst[l] * sst[l];
        if (l < kSpec) { // (the next level starts from this iterate; without a step it repeats it)
#pragma unroll
            for (int j = 0; j < NU; ++j) us[l + 1][j] = us[l][j];
        }
    }
}

// The same on DECOUPLED AXES (lmpc_lane_body: `axes`).  The QP separates axis by axis there -- Q^-1 is block diagonal, a bound on a control of
// axis c touches axis c only -- so the active-set run of the whole problem is an interleaving of the axes' own runs: the picks of different axes
// do not see each other (no multiplier of one axis moves when another takes a step, nothing is dropped), the counters add up, and the order
// between axes does not matter.  So EVERY axis whose bound on u_0 is violated takes its step at once -- W is diagonal: the control lands on
// its bound -- and ONE more trajectory carries all of them: the iterate after as many steps as axes stepped.
template <int NU>
COPRA_DEV void lane_spec_axes(const double (&W)[NU][NU], const double (&ubk)[NU], const double (&lbk)[NU], const double (&u0)[NU], double (&u1)[NU],
    double vsmall, bool bad, bool (&stepped)[NU], double (&sst)[NU])
{
#pragma unroll
    for (int c = 0; c < NU; ++c) {
        const double slu = ubk[c] - u0[c], sll = u0[c] - lbk[c];
        const bool vu = slu <= -vsmall, vl = sll <= -vsmall; // (qpgen2's order: the upper bound before the lower one)
        const double best = vu ? slu : sll, sig = vu ? -1.0 : 1.0;
        const double z = sig * W[c][c], zn = W[c][c]; // z = H n, n = sig e_c; z'n = |Lam^-1 e_c|^2
        const bool go = (vu || vl) && !bad && !(ubk[c] - lbk[c] <= -vsmall) && zn > 0.0 && z * z > vsmall; // (as lane_spec_steps, level 1)
        const double t2 = go ? -best / zn : 0.0;
        u1[c] = go ? u0[c] + t2 * z : u0[c];
        stepped[c] = go;
        sst[c] = go ? best : 0.0;
    }
}

// SREFS: the build for controllers with reference trajectories (FusedPlan::stage_refs) -- its own instantiation, so that the registers of
// stage_h below are not the headline's (measured on the one build for both: 3 VGPRs of the sweep in scratch memory)
// SPEC: the build that takes the first steps of the active-set iteration itself (FusedPlan::lane_spec); without it no trajectory rides along
// (round 5 measured the hand-over form and tracking controllers whose instances end at their minimiser 45 us per 65 536 slower with them)
template <int NX, int NU, bool SREFS = false, bool SPEC = true>
COPRA_DEV void lmpc_lane_body(const FusedPlan& P, int group)
{
    constexpr int NZ = NX + NU, KW = NU * (NX + 1), RW = NZ + 2; // (a row of the table: E | G | f | its index)
    constexpr int WR = KW; // rows of the lane-major workspace per stage (plan.hpp: lane_ws_rows): K | kv
    // the instance-major hand-over block (plan.hpp: lane_ws2_doubles): Lam^-1 of every stage, then the norm sums -- staged in LDS per group
    // of kLaneGroup stages and written as contiguous segments per instance, like U and X
    constexpr int NL = NU * (NU + 1) / 2, NLU = NL + NU, GL = kLaneGroup; // (per stage of the block: Lam^-1 packed | kv)
    static_assert(NU >= 1 && NU <= 3, "the control block is eliminated in closed form");
    const int lane = lane_id();
    constexpr int GRP = kWave; // instances of this wave
    const int inst = group * GRP + lane;
    const bool valid = inst < P.batch;
    const int NH = P.N;
    const double* tab = P.params + P.lane_tab;
    int oh_, oHN_, ohN_, oRows_;
    lane_tab_offsets(NX, NU, oh_, oHN_, ohN_, oRows_);
    const int oh = oh_, oHN = oHN_, ohN = ohN_, oRows = oRows_;
    double* lds = lds_base();
    if (P.lane_zero && group == 0 && lane == 0) P.lane_zero[0] = P.lane_zero[2] = 0; // (the NEXT solve's counters: nobody reads them now)
    const int left = P.batch - group * GRP;
    const int ninst = left < GRP ? left : GRP; // instances of this wave
    const int T2 = NH * (NLU + NX); // doubles per instance of the hand-over block
    const bool handover = !SPEC && P.lane_handover && P.lane_ws2; // (the speculating build hands nothing over)
    double* const ws2 = handover ? P.lane_ws2 + (size_t)(group * GRP) * T2 : nullptr;
    // (phase stamps of the wave, with copra_batch_enable_phase_profile: staging | sweep | roll-out | verdict, in row `group` of the profile --
    //  the first tier's rows are per instance, tools/exp/lane_tier1_phases.py leaves the first 1024 out)
    long long stamp[5];
    stamp[0] = P.prof ? cycle_counter() : 0;

    // ---- 0. this lane's system ----
    double A[NX * NX], B[NX * NU], d[NX], x[NX];
    {
        double rA[NX * NX], rB[NX * NU], rd[NX], rx[NX];
        lane_fetch<NX * NX>(P.A, group, P.batch, rA);
        lane_fetch<NX * NU>(P.B, group, P.batch, rB);
        lane_fetch<NX>(P.d, group, P.batch, rd);
        lane_fetch<NX>(P.x0, group, P.batch, rx);
        lane_transpose_in<NX * NX>(rA, group, P.batch, lds, A);
        lane_transpose_in<NX * NU>(rB, group, P.batch, lds, B);
        lane_transpose_in<NX>(rd, group, P.batch, lds, d);
        lane_transpose_in<NX>(rx, group, P.batch, lds, x);
    }
    // the stage cost H | h into LDS, behind the staging area: read there by every stage of the sweep (a wave-uniform address: one
    // broadcast read per entry, in order -- scalar loads come back out of order and every use waited for all of them)
    int oHl_ = 0;
    (void)lane_lds_doubles(NX, NU, oHl_);
    double* Hl = lds + oHl_;
    for (int e = lane; e < NZ * NZ + NZ; e += kWave) Hl[e] = tab[e];
    // ... and behind it, when they fit (FusedPlan::lane_tlds), the rows of every step and the bounds: the roll-out reads them there
    double* const Tl = Hl + ((NZ * NZ + NZ + 1) & ~1);
    const int tl_rows = (NH + 1) * P.lane_rps * RW;
    const bool tlds = P.lane_tlds > 0;
    if (tlds) {
        for (int e = lane; e < tl_rows; e += kWave) Tl[e] = tab[oRows + e];
        for (int e = lane; e < P.n; e += kWave) {
            Tl[tl_rows + e] = P.ub[e];
            Tl[tl_rows + P.n + e] = P.lb[e];
        }
    }
    wave_sync();
    auto AB = [&](int l, int a) -> double { return a < NX ? A[l + NX * a] : B[l + NX * (a - NX)]; }; // [A B](l, a)

    stamp[1] = P.prof ? cycle_counter() : 0;
    // Per-instance cost references (copra_batch_set_cost_reference: every instance tracks its own goal): the affine terms h = -sum_t
    // [M N]_t' W_t p_t and hN differ per lane then.  They are rebuilt from the coefficient table of the plan builder (lane_cref), this
    // lane's references where a cost has them and the controller-wide ones elsewhere, and h waits in the (idle) staging area for the sweep.
    bool own_refs = false;
    for (int t = 0; t < P.ncost; ++t) own_refs = own_refs || P.cost_p[t] != nullptr;
    double hNl[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) hNl[i] = uniform_load(tab, ohN + i);
    constexpr int HS = (NZ + GL * NLU + 2 * NX + NL) | 1; // this lane's slot of the staging area during the sweep: h (NZ) | Lam^-1 of the group's stages |
                                                    // x0 and d, which wait here for the roll-out -- the sweep reads d from the slot (ONE address
                                                    // register for all; odd: no bank conflicts)
    double hl[NZ]; // (the sweep reads h from this lane's slot of the staging area either way: no branch per use in its loop)
#pragma unroll
    for (int a = 0; a < NZ; ++a) hl[a] = uniform_load(tab, oh + a);
    // Reference TRAJECTORIES (FusedPlan::stage_refs, CostTerm::pstride): the reference changes along the horizon, h is rebuilt the same way
    // before every stage of the sweep (stage_h below); the terminal term takes the reference of the last step.
    constexpr bool srefs = SREFS;
    const int li0 = valid ? inst : 0;
    if (own_refs || srefs) {
#pragma unroll
        for (int a = 0; a < NZ; ++a) hl[a] = 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) hNl[i] = 0.0;
        for (int t = 0; t < P.ncost; ++t) {
            const int rows_t = P.cost[t].rows;
            const double* const pr = (P.cost_p[t] ? P.cost_p[t] + (size_t)li0 * P.cost[t].prows : P.params + P.cost[t].offP)
                + (P.cost[t].pstride ? P.cost[t].prows - P.cost[t].pstride : 0);
            for (int r = 0; r < rows_t; ++r) {
                const double pv_r = pr[r];
                const int co = P.lane_cref + (t * 6 + r) * (NZ + NX);
#pragma unroll
                for (int a = 0; a < NZ; ++a) hl[a] += uniform_load(tab, co + a) * pv_r;
#pragma unroll
                for (int i = 0; i < NX; ++i) hNl[i] += uniform_load(tab, co + NZ + i) * pv_r;
            }
        }
    }
    wave_sync();
#pragma unroll
    for (int a = 0; a < NZ; ++a) lds[lane * HS + a] = hl[a];
    // (x0 is not needed before the roll-out: twelve registers the sweep -- at the full register file -- has better use for)
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        lds[lane * HS + NZ + GL * NLU + i] = x[i];
        lds[lane * HS + NZ + GL * NLU + NX + i] = d[i];
    }
    wave_sync();
    // ---- 1. backward Riccati sweep; K_k | kv_k to the workspace, lane-major: element e of stage k at ws[(k KW + e) bp + inst] ----
    double* const ws = P.lane_ws;
    const size_t bp = (size_t)P.lane_bp;
    // this lane's byte offset in a workspace row: its instance's column, or -- lanes without an instance -- one of the 64 spare columns
    // behind the batch (copra_batch_solve sizes the rows lane_bp = batch rounded up to 64, + 64)
    const unsigned ioff = (unsigned)(valid ? inst : P.lane_bp - kWave + lane) * 8u;
    // cost-to-go: the upper triangle only ((i, l), i <= l, at i + NX l) -- both halves would be 15 more doubles carried around the loop
    double Pm[NX * NX], pv[NX];
#pragma unroll
    for (int l = 0; l < NX; ++l)
#pragma unroll
        for (int i = 0; i <= l; ++i) Pm[i + NX * l] = uniform_load(tab, oHN + i + NX * l);
    auto Ps = [&](int i, int l) -> double { return i <= l ? Pm[i + NX * l] : Pm[l + NX * i]; };
#pragma unroll
    for (int i = 0; i < NX; ++i) pv[i] = hNl[i];
    // h of stage k into this lane's slot (reference trajectories): every load of the stage is issued before the first is used
    auto stage_h = [&](int k) {
        double pvk[kRicMaxCosts][6];
#pragma unroll
        for (int t = 0; t < kRicMaxCosts; ++t) {
            const int tt = t < P.ncost ? t : 0;
            const CostTerm& ct = P.cost[tt];
            const int ps = ct.pstride, rc = ct.rows;
            const int kk = (ps && (k + 1) * ps > ct.prows) ? ct.prows / ps - 1 : k; // (a cost without a step k: zero coefficients)
            const double* const pr = (P.cost_p[tt] ? P.cost_p[tt] + (size_t)li0 * ct.prows : P.params + ct.offP) + kk * ps;
#pragma unroll
            for (int r = 0; r < 6; ++r) pvk[t][r] = pr[r < rc ? r : (rc > 0 ? rc - 1 : 0)]; // (rows past the term's: zero coefficients)
            // (Measured and dropped: scalar loads for a controller-wide reference -- the same value for every lane --: the step 0.360 ->
            //  0.370 ms, they come back out of order and every wait for one of them is a wait for the LDS traffic of the sweep as well;
            //  the loads requested at the top of the stage and h added to the affine column behind the products of M: 0.360 -> 0.361 ms.)
        }
        double hk[NZ];
#pragma unroll
        for (int a = 0; a < NZ; ++a) hk[a] = 0.0;
#pragma unroll
        for (int t = 0; t < kRicMaxCosts; ++t) {
            if (t < P.ncost) {
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    const int co = P.lane_cref + (t * 6 + r) * (NZ + NX);
#pragma unroll
                    for (int a = 0; a < NZ; ++a) hk[a] += uniform_load(tab, co + a) * pvk[t][r];
                }
            }
        }
#pragma unroll
        for (int a = 0; a < NZ; ++a) lds[lane * HS + a] = hk[a];
    };
    // Decoupled axes (FusedPlan::lane_axes: the costs couple none; here: neither do the systems of this wave): K(c, j) is exactly zero for
    // j % NU != c at every stage -- the recursion stays axis by axis: sums of products with an exact zero factor -- so those entries are
    // neither formed nor written (stage_core below) nor read (roll_groups), and the products with the entries of A, B, P and K between axes
    // are left out of both phases: the same sums in the same order without the terms that are +-0 -- bit-identical results, 841 -> ~ 190
    // multiply-adds per stage of the sweep for the CoM model (three double integrators), 12 of its 18 gains never in memory.  ONE vote per
    // wave decides (round 5 had tried masks tested per stage: they cost the sweep what they saved the roll-out).  The speculating build
    // only: the hand-over form's tier gathers K from the workspace.
    bool axes = false;
    if constexpr (SPEC && NU > 1 && NX % NU == 0) {
        if (P.lane_axes) {
            double stray = 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j)
#pragma unroll
                for (int i = 0; i < NX; ++i)
                    if (i % NU != j % NU) stray += fabs(A[i + NX * j]);
#pragma unroll
            for (int c = 0; c < NU; ++c)
#pragma unroll
                for (int i = 0; i < NX; ++i)
                    if (i % NU != c) stray += fabs(B[i + NX * c]);
            axes = !wave_any(!(stray == 0.0)); // (a NaN couples)
        }
    }
    bool bad = false;
    // (the sweep, the hand-over between the phases and the roll-out are DEFINED in this order and run further down, one build of each per kind of
    //  wave -- decoupled axes or not: ONE branch around all three; with one branch per phase the register allocator left two values of the
    //  dense sweep in scratch memory)
    auto sweep_all = [&](auto axes_tag) {
    for (int k = NH - 1; k >= 0; --k) {
        if constexpr (SREFS) stage_h(k);
        int hoff = oHl_;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(hoff)); // (opaque per stage: the reads stay inside the loop -- hoisted, they would cost 180 registers)
#endif
        const double* const Hk = lds + hoff;
        double M[NZ][NZ], mz[NZ]; // upper triangle (a <= b)
        double Ni[NU][NU], K[NU][NX], kv[NU];
        double* const wk = ws + ((size_t)k * WR) * bp; // (a wave-uniform row pointer + the lane's 32-bit index: one address register per lane, not one pair per store)
        // One stage of the sweep.  AXT: the wave's systems and the controller's costs couple no two axes (`axes` above) -- every matrix of the
        // recursion is then zero between entries of different axes, exactly, and the products with those entries are left out: the SAME sums in
        // the same order without the terms that are +-0 (841 -> ~ 190 multiply-adds per stage for three axes of a double integrator), the gains
        // between axes neither formed nor stored.  `sm(a, b)`: z-indices a and b (states 0 .. NX-1, controls NX ..) belong to one axis.
        auto stage_core = [&](auto axes_tag) {
            constexpr bool AXT = decltype(axes_tag)::value;
            auto sm = [](int a2, int b2) -> bool { return !AXT || (a2 < NX ? a2 % NU : a2 - NX) == (b2 < NX ? b2 % NU : b2 - NX); };
            {
                double tq[NX], dl[NX]; // P+ d + p+  (d from this lane's slot: twelve registers less across the stage)
#pragma unroll
                for (int i = 0; i < NX; ++i) dl[i] = lds[lane * HS + NZ + GL * NLU + NX + i];
#pragma unroll
                for (int l = 0; l < NX; ++l) {
                    double s = pv[l];
#pragma unroll
                    for (int i = 0; i < NX; ++i)
                        if (sm(l, i)) s += Ps(l, i) * dl[i];
                    tq[l] = s;
                }
#pragma unroll
                for (int a2 = 0; a2 < NZ; ++a2) {
                    double s = lds[lane * HS + a2];
#pragma unroll
                    for (int l = 0; l < NX; ++l)
                        if (sm(l, a2)) s += AB(l, a2) * tq[l];
                    mz[a2] = s;
                }
            }
#pragma unroll
            for (int b2 = 0; b2 < NZ; ++b2) {
                double Tb[NX]; // column b of P+ [A B]
#pragma unroll
                for (int l = 0; l < NX; ++l) {
                    double s = 0.0;
#pragma unroll
                    for (int i = 0; i < NX; ++i)
                        if (sm(l, i) && sm(i, b2)) s += Ps(l, i) * AB(i, b2);
                    Tb[l] = s;
                }
#pragma unroll
                for (int a2 = 0; a2 <= b2; ++a2) {
                    if (sm(a2, b2)) {
                        double s = Hk[a2 + NZ * b2];
#pragma unroll
                        for (int l = 0; l < NX; ++l)
                            if (sm(l, a2)) s += AB(l, a2) * Tb[l];
                        M[a2][b2] = s;
                    } else {
                        M[a2][b2] = 0.0;
                    }
                }
            }
            // -M_uu^-1 (symmetric; positive definite <=> the trailing minors are positive)
            if constexpr (NU == 1) {
                const double m00 = M[NX][NX];
                bad = bad || !(m00 > 0.0);
                Ni[0][0] = -1.0 / m00;
            } else if constexpr (NU == 2) {
                const double m00 = M[NX][NX], m01 = M[NX][NX + 1], m11 = M[NX + 1][NX + 1];
                const double det = m00 * m11 - m01 * m01;
                bad = bad || !(m11 > 0.0) || !(det > 0.0);
                const double nr = -1.0 / det;
                Ni[0][0] = m11 * nr;
                Ni[0][1] = Ni[1][0] = -m01 * nr;
                Ni[1][1] = m00 * nr;
            } else {
                const double m00 = M[NX][NX], m01 = M[NX][NX + 1], m02 = M[NX][NX + 2], m11 = M[NX + 1][NX + 1], m12 = M[NX + 1][NX + 2],
                             m22 = M[NX + 2][NX + 2];
                const double c00 = m11 * m22 - m12 * m12, c01 = m02 * m12 - m01 * m22, c02 = m01 * m12 - m02 * m11;
                const double det = m00 * c00 + (m01 * c01 + m02 * c02);
                bad = bad || !(m22 > 0.0) || !(c00 > 0.0) || !(det > 0.0);
                const double nr = -1.0 / det;
                Ni[0][0] = c00 * nr;
                Ni[0][1] = Ni[1][0] = c01 * nr;
                Ni[0][2] = Ni[2][0] = c02 * nr;
                Ni[1][1] = (m00 * m22 - m02 * m02) * nr;
                Ni[1][2] = Ni[2][1] = (m01 * m02 - m00 * m12) * nr;
                Ni[2][2] = (m00 * m11 - m01 * m01) * nr;
            }
#pragma unroll
            for (int c = 0; c < NU; ++c) {
#pragma unroll
                for (int j = 0; j < NX; ++j) {
                    double s = 0.0;
#pragma unroll
                    for (int e = 0; e < NU; ++e)
                        if (sm(j, NX + e) && sm(NX + c, NX + e)) s += Ni[c][e] * M[j][NX + e];
                    K[c][j] = s;
                }
                double s = 0.0;
#pragma unroll
                for (int e = 0; e < NU; ++e)
                    if (sm(NX + c, NX + e)) s += Ni[c][e] * mz[NX + e];
                kv[c] = s;
            }
#pragma unroll
            for (int j = 0; j < NX; ++j)
#pragma unroll
                for (int i = 0; i <= j; ++i) {
                    if (!sm(i, j)) continue; // (stays the zero it has been since the terminal cost)
                    double s = M[i][j];
#pragma unroll
                    for (int c = 0; c < NU; ++c)
                        if (sm(i, NX + c)) s += M[i][NX + c] * K[c][j];
                    Pm[i + NX * j] = s; // (i <= j)
                }
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double s = mz[i];
#pragma unroll
                for (int c = 0; c < NU; ++c)
                    if (sm(i, NX + c)) s += M[i][NX + c] * kv[c];
                pv[i] = s;
            }
            if (!(COPRA_LANE_EXP & 128)) {
#pragma unroll
                for (int c = 0; c < NU; ++c) {
#pragma unroll
                    for (int j = 0; j < NX; ++j)
                        if (sm(j, NX + c)) lane_put(wk + (size_t)(c + NU * j) * bp, ioff, K[c][j]); // (between axes: nothing -- the roll-out does not read them)
                    lane_put(wk + (size_t)(NU * NX + c) * bp, ioff, kv[c]);
                }
            } else if (k == 0) { // (experiment: the sweep leaves nothing -- something of it has to be alive)
                lane_put(wk, ioff, K[0][0] + kv[0] + K[NU - 1][NX - 1]);
            }
        };
        stage_core(axes_tag);
        if (k == 0) { // W = M_uu,0^-1 = -Ni, packed by rows: what the speculative steps of the roll-out compute with (parked in this lane's slot)
#pragma unroll
            for (int i = 0; i < NU; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) lds[lane * HS + NZ + GL * NLU + 2 * NX + i * (i + 1) / 2 + j] = -Ni[i][j];
        }
        // Lam^-1 (Lam Lam' = M_uu), packed by rows: what the two products of the factor use (ric_factor.hpp) -- for the first tier,
        // and only when it takes the factor over (FusedPlan::lane_handover; round-3 advisor finding: in front of the other tiers the
        // pass only filters, and these values and the norm sums below were dead traffic).  Into this lane's slot of the LDS staging area
        // (behind h), the group's segment of every instance leaves when its last stage is done.
        if (handover) {
            double lm[NU][NU], rd[NU], lid[NU][NU];
#pragma unroll
            for (int cb = 0; cb < NU; ++cb)
#pragma unroll
                for (int ca = 0; ca <= cb; ++ca) lm[cb][ca] = M[NX + ca][NX + cb];
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
                for (int r2 = c; r2 < NU; ++r2) {
                    double li;
                    if (r2 == c) {
                        li = rd[c];
                    } else {
                        double v = 0.0;
#pragma unroll
                        for (int q = c; q < r2; ++q) v -= lm[r2][q] * lid[q][c];
                        li = v * rd[r2];
                    }
                    lid[r2][c] = li;
                    lds[lane * HS + NZ + (k % GL) * NLU + r2 * (r2 + 1) / 2 + c] = li;
                }
#pragma unroll
            for (int c = 0; c < NU; ++c) lds[lane * HS + NZ + (k % GL) * NLU + NL + c] = kv[c]; // (the tier rolls out itself: lmpc_fused_ric.hpp)
            if (k % GL == 0) { // stages k .. k + GL - 1 (fewer in the group that holds the last stage)
                const int cnt = NH - k < GL ? NH - k : GL;
                wave_sync();
                double* const dst = ws2 + (size_t)k * NLU;
                int lq = lane;
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" : "+v"(lq)); // (opaque per group: the index arithmetic below stays here -- hoisted out of the sweep's loop its
                                             //  48 lane-dependent values pushed six registers of the sweep into scratch memory)
#endif
#pragma unroll 4
                for (int j = 0; j < GL * NLU; ++j) {
                    const int e = j * kWave + lq, il = e / (GL * NLU), c = e - il * (GL * NLU);
                    if (il < ninst && c < cnt * NLU) dst[(size_t)il * T2 + c] = lds[il * HS + NZ + c];
                }
                wave_sync(); // (the staging area is written again by the next group)
            }
        }
    }
    };
    auto mid = [&]() {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        double xi = lds[lane * HS + NZ + GL * NLU + i], di = lds[lane * HS + NZ + GL * NLU + NX + i];
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(xi), "+v"(di)); // (really read back: the values do not stay in registers across the sweep)
#endif
        x[i] = xi;
        d[i] = di;
    }
    stamp[2] = P.prof ? cycle_counter() : 0;
    // W = M_uu,0^-1 from the slot (the sweep's last stage left it there) moves to where the roll-out's staging does not reach -- the place of the
    // norm sums' staging, which only the hand-over form of the pass uses, and that form does not speculate: the speculative steps below read it
    // there (in registers it would be live across the whole roll-out loop: that was what pushed the sweep into scratch memory)
    {
        double w0[NL];
#pragma unroll
        for (int e = 0; e < NL; ++e) w0[e] = lds[lane * HS + NZ + GL * NLU + 2 * NX + e];
        wave_sync(); // (every lane has read x0, d and W from its slot)
#pragma unroll
        for (int e = 0; e < NL; ++e) lds[kWave * (((kLaneGroup * NX) | 1) + ((kLaneGroup * NU) | 1)) + lane * ((kLaneGroup * NX) | 1) + e] = w0[e];
    }
    };
    // ---- 2. roll-out from x0 with qpgen2's first scan inside: rows of step k on (x_k, u_k), the bounds of u_k ----
    const double vsmall = P.vsmall;
    const int rps = P.lane_rps;
    int nviol = 0; // violated rows and bounds at the unconstrained minimiser: what the size of the final active set goes with (lane_hist below)
    const int li = valid ? inst : 0;
    const double* const lbp = P.lb_inst ? P.lb_inst + (size_t)li * P.n : P.lb;
    const double* const ubp = P.ub_inst ? P.ub_inst + (size_t)li * P.n : P.ub;
    // Stages in groups of GS.  One wave per SIMD: nothing else hides a trip to memory, so the gains of a stage are requested kLaneAhead stages
    // ahead -- rotating buffers, each refilled as soon as its stage has used it.  The group's states and controls are collected in
    // LDS and leave as contiguous segments per instance.  Everything is written for every instance -- the verdict comes last --, the
    // first tier overwrites what it solves again.
    constexpr int GS = kLaneGroup, SX = (GS * NX) | 1, SU = (GS * NU) | 1;
    double* const ldx = lds; // [lane][SX]
    double* const ldu = lds + kWave * SX; // [lane][SU]
    double* const ldn = lds + kWave * (SX + SU); // [lane][SX]: the norm sums of the group (hand-over block)
    // What a stage reads from memory -- its gains, the bounds of its controls, the right-hand sides of its rows -- is requested TOGETHER, a
    // stage or two ahead, and from one kind of address each: memory operations retire in order, so anything requested behind the gains waits for
    // them, and a load on one side of a branch makes the compiler wait for EVERYTHING in flight where the paths join.  (Round 5, measured with
    // stamps inside the loop: an instance's own right-hand side -- a load behind `if (rhs_mine)` in the rows' loop -- cost every stage of every
    // controller a wait for the gains requested a moment before: 114 k of the roll-out's 197 k cycles, the prefetch bought nothing.)
    //   bounds: this instance's own (copra_batch_set_control_bounds) or the controller's, a pointer chosen once
    //   right-hand sides: this instance's own (copra_batch_set_constraint_rhs: [batch][mgen] in the stacked order; a row of the table that
    //     is not there keeps +inf) or the table's -- the first RQ rows of a step; a step with more rows reads the others where it needs them
    const double* const rhs_mine = P.row_f_inst ? P.row_f_inst + (size_t)li * P.mgen : nullptr;
    constexpr int RQ = 4, KF = KW + 2 * NU + RQ; // a stage's buffer: K | kv | ub | lb | f of its first rows
    auto row_index = [&](int ro) -> int { return tlds ? (int)Tl[ro + NZ + 1] : (int)uniform_load(tab, oRows + ro + NZ + 1); }; // (ro: the row's offset in the table)
    auto fetch_stage = [&](auto axes_tag, double (&buf)[KF], int k) {
        constexpr bool AXT = decltype(axes_tag)::value; // (decoupled axes: the gains between them are not there -- and not used, roll_groups below)
        const int kk = k < NH ? k : NH - 1; // (past the end: the last stage once more, unused)
        const double* const wk = ws + ((size_t)kk * WR) * bp;
#pragma unroll
        for (int e = 0; e < KW; ++e) { // (entry e < NU NX of the gains: K(e % NU, e / NU))
            if (AXT && e < NU * NX && (e / NU) % NU != e % NU)
                buf[e] = 0.0;
            else
                buf[e] = (COPRA_LANE_EXP & 2) ? 1e-3 * (e + kk) : lane_get(wk + (size_t)e * bp, ioff);
        }
#pragma unroll
        for (int c = 0; c < NU; ++c) {
            buf[KW + c] = ubp[kk * NU + c];
            buf[KW + NU + c] = lbp[kk * NU + c];
        }
#pragma unroll
        for (int r = 0; r < RQ; ++r) {
            const int kr = k < NH ? k : NH; // (the rows of the LAST STATE ride with the request past the end: read behind the loop, roll_groups)
            const int ro = (kr * rps + (r < rps ? r : 0)) * RW;
            const int idx = r < rps ? row_index(ro) : -1;
            const double* const fp = r >= rps ? tab : (rhs_mine && idx >= 0) ? rhs_mine + idx : tab + oRows + ro + NZ;
            buf[KW + 2 * NU + r] = *fp;
        }
    };
    // ---- the FIRST STEPS of the active-set iteration, speculatively (round 5) ----
    // Where the unconstrained minimiser saturates an actuator NOW -- a bound on u_0 is its most violated constraint, qpgen2's first pick: on
    // BASELINE configs[2] that is every one of the 53 % of the instances this pass used to leave to the first tier, 71 % of them are finished
    // by that one constraint and most of the rest by a second bound on u_0 -- the steps the iteration takes have a closed form in the
    // quantities of THIS roll-out.  The normal of a bound on component c of u_0 is n = -+e_c in stage 0 and zero elsewhere, so w = R^-T n
    // stops at stage 0 (ric_factor.hpp: the backward recursion never starts), everything the method computes with such normals lives in the
    // NU x NU block W = M_uu,0^-1 = Lam_0^-T Lam_0^-1 (the stage-0 block of Q^-1: M_uu,0 is the Schur complement of the later stages), and
    // from stage 1 on the step direction z_k = K_k xi_k IS the closed loop: the new iterate is the roll-out from x0 with the new u_0 in front
    // and the same gains behind.  So kSpec more trajectories ride along (one more K x + kv and A x + B u + d per stage each); trajectory l
    // is the iterate after l steps, its rows and bounds are scanned like the first one's, and at the end, with l the first level that
    // violates nothing:
    //     every pick up to l was a bound on u_0 AND strictly the worst of ALL rows at its iterate, normalised as qpgen2 does
    //                                              -> done, iterations (l + 1, 0): U_l, X_l are the results
    //     anything else (another pick, a tie, a constraint to drop, more than kSpec steps)  -> the first tier, from scratch
    // Same decisions as the tier: its pick, its step lengths t2 = -s / z'n and t1 = lambda / r (a partial step -- the tier's business), its test
    // slack <= -vsmall with the active rows left out; near-ties (1e-9 relative) go to the tier.  A lane writes its DEEPEST trajectory to
    // the results -- levels it does not speculate on repeat the one before, the verdict comes last --, which is why the tier behind this
    // pass rolls out for itself (lmpc_fused_ric.hpp, from_lane).
#ifndef COPRA_LANE_KSPEC
#define COPRA_LANE_KSPEC 2
#endif
    constexpr int kSpec = !SPEC ? 0 : NU >= 2 ? COPRA_LANE_KSPEC : 1; // levels: steps taken here at most (a third one on u_0 would fix all of it: 1 % of the headline's batch)
    const bool spec_on = SPEC && P.lane_spec != 0; // (the controller's rows are the compact variant's: one component of a state, or controls only)
    bool specl[kSpec + 1], uniql[kSpec + 1], violl[kSpec + 1]; // [l]: level l was speculated on | its pick was the pick | trajectory l violates something
    int scl[kSpec + 1]; // component of u_0 whose bound level l >= 1 adds (-1: none)
    double sst[kSpec + 1], sst2[kSpec + 1]; // the slack of level l's pick at trajectory l - 1 (negative), and its square
#pragma unroll
    for (int l = 0; l <= kSpec; ++l) {
        specl[l] = false;
        uniql[l] = true;
        violl[l] = false;
        scl[l] = -1;
        sst[l] = sst2[l] = 0.0;
    }
    double bmin[kSpec + 1], bmin_other[kSpec + 1]; // worst bound slack per level: of all bounds | of those that are not the next level's pick
#pragma unroll
    for (int l = 0; l <= kSpec; ++l) bmin[l] = bmin_other[l] = 0.0;
    double xs[kSpec + 1][NX]; // xs[0]: the unconstrained minimiser's trajectory (x above)
    auto mid2 = [&]() { // (what the roll-out starts from, once the sweep is behind it)
        violl[0] = bad;
#pragma unroll
        for (int l = 0; l <= kSpec; ++l)
#pragma unroll
            for (int c = 0; c < NX; ++c) xs[l][c] = x[c];
    };
    // one row  e' x + g' u <= f  at every iterate
    auto row_eval = [&](auto levels_tag, const double (&e)[NX], const double (&g)[NU], double f, const double (&xk)[kSpec + 1][NX], const double (&uk)[kSpec + 1][NU]) {
        constexpr int KS = decltype(levels_tag)::value; // (levels carried: kSpec, or 1 on decoupled axes)
#pragma unroll
        for (int l = 0; l <= KS; ++l) {
            double ax = 0.0;
#pragma unroll
            for (int c = 0; c < NX; ++c) ax += e[c] * xk[l][c];
#pragma unroll
            for (int c = 0; c < NU; ++c) ax += g[c] * uk[l][c];
            const double sl = f - ax;
            const bool v = sl <= -vsmall; // (gi_core.hpp: |s| < vsmall counts as zero, a negative slack is a violation)
            violl[l] = violl[l] || v;
            if (l == 0) nviol += v ? 1 : 0;
            // A violated ROW below the last level: the tier's business.  (Whether the bound was still qpgen2's pick takes the row's norm --
            // |row of Psi|^2, a recursion G <- A G of its own through the roll-out: 126 multiply-adds per stage and 48 registers, the ones that sent
            // the roll-out to scratch memory.  Measured on the headline's batch: 186 of 65 536 instances ended here by that comparison, the pass
            // 278 -> 208 k cycles per wave without it.)
            if (l < KS) uniql[l + 1] = uniql[l + 1] && !v;
        }
    };
    // E x_k + G u_k <= f: the rows of step k; fq: the right-hand sides of its first RQ rows as fetch_stage left them (nullptr: none were fetched)
    auto check_rows = [&](auto levels_tag, int k, const double (&xk)[kSpec + 1][NX], const double (&uk)[kSpec + 1][NU], const double* fq) {
        for (int r = 0; r < rps; ++r) {
            double e[NX], g[NU];
            const int ro = (k * rps + r) * RW;
            if (tlds) {
#pragma unroll
                for (int c = 0; c < NX; ++c) e[c] = Tl[ro + c];
#pragma unroll
                for (int c = 0; c < NU; ++c) g[c] = Tl[ro + NX + c];
            } else {
#pragma unroll
                for (int c = 0; c < NX; ++c) e[c] = uniform_load(tab, oRows + ro + c);
#pragma unroll
                for (int c = 0; c < NU; ++c) g[c] = uniform_load(tab, oRows + ro + NX + c);
            }
            double f = 0.0;
            if (fq && r < RQ) {
#pragma unroll
                for (int t = 0; t < RQ; ++t) f = (t == r) ? fq[t] : f; // (r is wave-uniform: one select per entry, no indexed register)
            } else { // (the last state's rows, and rows past the buffer: read here, waiting for whatever is in flight)
                const int idx = row_index(ro);
                const double* const fp = (rhs_mine && idx >= 0) ? rhs_mine + idx : tab + oRows + ro + NZ;
                f = *fp;
            }
            row_eval(levels_tag, e, g, f, xk, uk);
        }
    };
#ifndef COPRA_LANE_KB
#define COPRA_LANE_KB 1
#endif
    constexpr int KB = SPEC ? COPRA_LANE_KB : kLaneAhead; // gain buffers: stages requested ahead
    static_assert(GS % KB == 0, "the buffers rotate inside a group");
    // block-row norms of the preview blocks G_s = A^s B as running sums over s (one stage of the roll-out = one block): what the row norms
    // of the first tier's compact variant are read from (lmpc_fused_ric.hpp: NB2) -- and what the rows of step k are normalised by above
    // (a row on x_k has the squared norm sum_{t < k} |row of G_t|^2: the sums BEFORE block k joins them)
    double Gp[NX * NU], ncum[NX];
#pragma unroll
    for (int e = 0; e < NX * NU; ++e) Gp[e] = B[e];
#pragma unroll
    for (int i = 0; i < NX; ++i) ncum[i] = 0.0;
    constexpr bool fine = (COPRA_LANE_EXP & 64) != 0; // (experiment: where the roll-out's cycles go -- gains and controls | rows and bounds | dynamics; the rest is the store phase)
    long long fineA = 0, fineB = 0, fineC = 0, fineT = 0;
    // The roll-out's loop, in two builds: AXT -- decoupled axes (`axes`, see the sweep): the products with the entries of A, B and K between
    // axes are left out of u = K x + kv and x+ = A x + B u + d (the same sums without the terms that are exactly zero) and those gains
    // are not read.
    int axes_iters = -1; // (decoupled axes: qpgen2's first counter of an instance that ends here, 0: it does not -- set by roll_groups)
    auto roll_groups = [&](auto axes_tag) {
    constexpr bool AXT = decltype(axes_tag)::value;
    auto sm = [](int a2, int b2) -> bool { return !AXT || (a2 < NX ? a2 % NU : a2 - NX) == (b2 < NX ? b2 % NU : b2 - NX); };
    // decoupled axes: ONE more trajectory -- every axis whose bound on u_0 is violated takes its step in it (lane_spec_axes) -- and the
    // bookkeeping per AXIS: did it step | the slack of its pick | its worst bound slack at the minimiser, the same without the pick, and after the step
    constexpr int KS = AXT ? (kSpec > 0 ? 1 : 0) : kSpec;
    using levels_t = std::integral_constant<int, KS>;
    bool ax_step[NU];
    double ax_sst[NU], ax_b0[NU], ax_b0o[NU], ax_b1[NU];
#pragma unroll
    for (int c = 0; c < NU; ++c) {
        ax_step[c] = false;
        ax_sst[c] = ax_b0[c] = ax_b0o[c] = ax_b1[c] = 0.0;
    }
    double Kq[KB][KF];
#pragma unroll
    for (int q = 0; q < KB; ++q) fetch_stage(axes_tag, Kq[q], q);
    for (int k0 = 0; k0 < NH; k0 += GS) {
        wave_sync(); // (the previous group has left the staging area)
#pragma unroll
        for (int q = 0; q < GS; ++q) {
            const int k = k0 + q;
            const bool on = k < NH;
            if (fine) fineT = cycle_counter();
            double ubk[NU], lbk[NU], fk[RQ]; // (out of the buffer before it is refilled)
#pragma unroll
            for (int c = 0; c < NU; ++c) {
                ubk[c] = Kq[q % KB][KW + c];
                lbk[c] = Kq[q % KB][KW + NU + c];
            }
#pragma unroll
            for (int r = 0; r < RQ; ++r) fk[r] = Kq[q % KB][KW + 2 * NU + r];
            double us[kSpec + 1][NU];
#pragma unroll
            for (int l = 0; l <= KS; ++l)
#pragma unroll
                for (int c = 0; c < NU; ++c) {
                    double acc = Kq[q % KB][NU * NX + c];
#pragma unroll
                    for (int j = 0; j < NX; ++j)
                        if (sm(j, NX + c)) acc += Kq[q % KB][c + NU * j] * xs[l][j];
                    us[l][c] = acc;
                }
            fetch_stage(axes_tag, Kq[q % KB], k + KB);
            if (fine) {
                double a0 = us[KS][0];
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" : "+v"(a0)); // (the controls are there)
#endif
                us[KS][0] = a0;
                const long long t = cycle_counter();
                fineA += t - fineT;
                fineT = t;
            }
            if (q == 0 && k0 == 0 && spec_on) {
                // W = M_uu,0^-1 (symmetric, packed by rows)
                double W[NU][NU];
#pragma unroll
                for (int i = 0; i < NU; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) W[i][j] = W[j][i] = ldn[lane * SX + i * (i + 1) / 2 + j];
                if constexpr (AXT && kSpec > 0)
                    lane_spec_axes<NU>(W, ubk, lbk, us[0], us[1], vsmall, bad, ax_step, ax_sst);
                else if constexpr (kSpec > 0)
                    lane_spec_steps<NU, kSpec>(W, ubk, lbk, us, vsmall, bad, specl, scl, sst, sst2);
            }
            if (on) {
                if (!(COPRA_LANE_EXP & 8)) check_rows(levels_t {}, k, xs, us, fk);
                if constexpr (AXT) { // the bounds of u_k, axis by axis (the pick and, after the step, the active bound -- stage 0 only -- left out)
                    const bool stage0 = q == 0 && k0 == 0;
#pragma unroll
                    for (int c = 0; c < NU; ++c) {
                        const double m0 = fmin(ubk[c] - us[0][c], us[0][c] - lbk[c]);
                        nviol += (m0 <= -vsmall) ? 1 : 0;
                        const bool mine = stage0 && ax_step[c];
                        ax_b0[c] = fmin(ax_b0[c], m0);
                        ax_b0o[c] = fmin(ax_b0o[c], mine ? 0.0 : m0);
                        if constexpr (KS > 0) {
                            const double m1 = fmin(ubk[c] - us[1][c], us[1][c] - lbk[c]);
                            ax_b1[c] = fmin(ax_b1[c], mine ? 0.0 : m1);
                        }
                    }
                } else if (!(COPRA_LANE_EXP & 16)) {
                    // the bounds of u_k at every iterate: the worst slack per level is all that is kept -- a level violates a bound iff its
                    // minimum is <= -vsmall, and the next level's pick was not the pick iff the minimum (the active bounds and the pick itself
                    // left out: they only exist at stage 0) comes within 1e-9 of its slack.  Two subtractions and two minima per bound and level.
                    const bool stage0 = q == 0 && k0 == 0;
#pragma unroll
                    for (int c = 0; c < NU; ++c) {
#pragma unroll
                        for (int l = 0; l <= kSpec; ++l) {
                            const double m = fmin(ubk[c] - us[l][c], us[l][c] - lbk[c]);
                            if (l == 0) nviol += (m <= -vsmall) ? 1 : 0;
                            // (the bounds level <= l holds active -- their twins cannot be violated -- count for nothing; the pick of level l + 1
                            //  counts as a violation of level l, not against itself)
                            const bool active = stage0 && ((l >= 1 && c == scl[kSpec >= 1 ? 1 : 0]) || (l >= 2 && c == scl[kSpec]));
                            const bool pick = stage0 && l < kSpec && c == scl[l < kSpec ? l + 1 : kSpec];
                            bmin[l] = fmin(bmin[l], active ? 0.0 : m);
                            if (l < kSpec) bmin_other[l] = fmin(bmin_other[l], (active || pick) ? 0.0 : m);
                        }
                    }
                }
            }
            if (on && handover && !(COPRA_LANE_EXP & 4)) { // |row i of G_k|^2 added (staged for the hand-over block), and the next block
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    double sq = 0.0;
#pragma unroll
                    for (int c = 0; c < NU; ++c) sq += Gp[i + NX * c] * Gp[i + NX * c];
                    ncum[i] += sq;
                    ldn[lane * SX + q * NX + i] = ncum[i];
                }
#pragma unroll
                for (int c = 0; c < NU; ++c) { // (column by column, in place: NX temporaries instead of a second block)
                    double gn[NX];
#pragma unroll
                    for (int i = 0; i < NX; ++i) {
                        double acc = 0.0;
#pragma unroll
                        for (int l = 0; l < NX; ++l) acc += A[i + NX * l] * Gp[l + NX * c];
                        gn[i] = acc;
                    }
#pragma unroll
                    for (int i = 0; i < NX; ++i) Gp[i + NX * c] = gn[i];
                }
            }
            if (fine) {
                double a0 = bmin[0];
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" : "+v"(a0));
#endif
                bmin[0] = a0;
                const long long t = cycle_counter();
                fineB += t - fineT;
                fineT = t;
            }
            // (the deepest trajectory: levels without a step repeat the one before)
#pragma unroll
            for (int c = 0; c < NX; ++c) ldx[lane * SX + q * NX + c] = xs[KS][c];
#pragma unroll
            for (int c = 0; c < NU; ++c) ldu[lane * SU + q * NU + c] = us[KS][c];
#pragma unroll
            for (int l = 0; l <= KS; ++l) {
                double xn[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    double acc = d[i];
#pragma unroll
                    for (int j = 0; j < NX; ++j)
                        if (sm(i, j)) acc += A[i + NX * j] * xs[l][j];
#pragma unroll
                    for (int c = 0; c < NU; ++c)
                        if (sm(i, NX + c)) acc += B[i + NX * c] * us[l][c];
                    xn[i] = acc;
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) xs[l][i] = on ? xn[i] : xs[l][i];
            }
            if (fine) {
                double a0 = xs[KS][0];
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" : "+v"(a0));
#endif
                xs[KS][0] = a0;
                fineC += cycle_counter() - fineT;
            }
            sched_fence(); // (nothing of the next stage moves up here: its operands would be live twice)
        }
        wave_sync();
        if (!(COPRA_LANE_EXP & 1)) { // the group's states and controls: one contiguous segment per instance each
            const int nst = NH - k0 < GS ? NH - k0 : GS; // stages of this group
            lane_group_out<GS * NX>(P.trajectory + (size_t)(group * GRP) * P.X + (size_t)k0 * NX, (size_t)P.X, ldx, SX, ninst, nst * NX, lane);
            lane_group_out<GS * NU>(P.control + (size_t)(group * GRP) * P.n + (size_t)k0 * NU, (size_t)P.n, ldu, SU, ninst, nst * NU, lane);
        }
        if (handover) // the group's norm sums of every instance: one contiguous segment of its hand-over block
            lane_group_out<GS * NX>(ws2 + (size_t)NH * NLU + (size_t)k0 * NX, (size_t)T2, ldn, SX, ninst, (NH - k0 < GS ? NH - k0 : GS) * NX, lane);
    }
    { // the last state's rows: their right-hand sides came with the last request (stage NH - 1 + KB >= NH: buffer (NH - 1) % KB ... the one
      // refilled last) -- read in place they waited for the results' stores in flight, three times
        double u0[kSpec + 1][NU], fl[RQ];
#pragma unroll
        for (int l = 0; l <= kSpec; ++l)
#pragma unroll
            for (int c = 0; c < NU; ++c) u0[l][c] = 0.0;
        const int qb = (NH - 1) % KB; // (stage NH - 1 refilled its own buffer last, with the request for stage NH - 1 + KB)
#pragma unroll
        for (int r = 0; r < RQ; ++r) {
            double v = Kq[0][KW + 2 * NU + r];
#pragma unroll
            for (int t = 1; t < KB; ++t) v = (t == qb) ? Kq[t][KW + 2 * NU + r] : v;
            fl[r] = v;
        }
        check_rows(levels_t {}, NH, xs, u0, fl);
        if (valid) { // the last state: out
            double* const xo = P.trajectory + (size_t)inst * P.X + (size_t)NH * NX;
#pragma unroll
            for (int c = 0; c < NX; ++c) xo[c] = xs[KS][c];
        }
    }
    if constexpr (AXT) {
        // the verdict on decoupled axes: every axis either violates no bound, or its bound on u_0 was its pick -- strictly its worst, and no ROW is
        // violated at the minimiser (whose normalised slack would have to be compared) -- and after the step it violates none; the rows hold at the
        // last iterate.  qpgen2's counters: one iteration per axis that stepped, and the scan that finds nothing; nothing dropped.
        bool okv = valid && !bad && !violl[KS] && (KS == 0 || uniql[KS]);
        int steps = 0;
        bool any0 = violl[0];
#pragma unroll
        for (int c = 0; c < NU; ++c) {
            any0 = any0 || (ax_b0[c] <= -vsmall);
            const bool st = ax_step[c];
            steps += st ? 1 : 0;
            okv = okv && (st ? (!(ax_b0o[c] <= -vsmall && ax_b0o[c] <= ax_sst[c] * (1.0 - 1e-9)) && !(ax_b1[c] <= -vsmall)) : !(ax_b0[c] <= -vsmall));
        }
        violl[0] = any0;
        axes_iters = okv ? 1 + steps : 0;
    }
    };
    auto sweep_and_roll_out = [&](auto axes_tag) {
        sweep_all(axes_tag);
        mid();
        mid2();
        roll_groups(axes_tag);
    };
    if constexpr (SPEC && NU > 1 && NX % NU == 0) {
        if (axes)
            sweep_and_roll_out(std::true_type {});
        else
            sweep_and_roll_out(std::false_type {});
    } else {
        sweep_and_roll_out(std::false_type {});
    }
    // the verdict: the first level that violates nothing, if every pick on the way to it was the pick
#pragma unroll
    for (int l = 0; l <= kSpec; ++l) {
        violl[l] = violl[l] || (bmin[l] <= -vsmall);
        if (l < kSpec) uniql[l + 1] = uniql[l + 1] && !(bmin_other[l] <= -vsmall && bmin_other[l] <= sst[l + 1] * (1.0 - 1e-9));
    }
    const bool viol = violl[0];
    int done_iters = 0; // qpgen2's first counter of an instance that ends here (0: it does not)
    {
        bool chain = valid && !bad;
#pragma unroll
        for (int l = 0; l <= kSpec; ++l) {
            if (l >= 1) chain = chain && specl[l] && uniql[l];
            if (chain && !violl[l] && done_iters == 0) done_iters = l + 1;
            chain = chain && violl[l]; // (a deeper level only exists behind a violated one)
        }
    }
    if (axes_iters >= 0) done_iters = axes_iters; // (decoupled axes: the verdict was taken axis by axis, roll_groups)
    const bool done1 = done_iters >= 2; // finished by the bounds on u_0 it speculated on

    stamp[3] = P.prof ? cycle_counter() : 0;
    // ---- 3. verdict: done, or one more entry of the first tier's list (one atomic per wave) ----
    const bool more = valid && viol && !done1;
    int total = 0;
    const int before = wave_prefix_count(more, total);
    if (total > 0) {
        int base = 0;
        if (lane == 0) base = atomic_add_i32(P.lane_count, total);
        base = bcast_i32(base, 0);
        if (more) P.lane_list[base + before] = bad ? (inst | (int)0x80000000) : inst; // (top bit: the factorisation failed -- status 2)
    }
    { // instances that ended by the steps taken here: what copra_batch_solve weighs the speculation's cost against (adapt_lane_pass)
        int nspec = 0;
        (void)wave_prefix_count(done1, nspec);
        if (nspec > 0 && lane == 0) (void)atomic_add_i32(P.lane_count + 2, nspec);
    }
    if ((valid && !viol) || done1) {
        P.status[inst] = 0;
        P.iter[2 * (size_t)inst] = done1 ? done_iters : 1; // (qpgen2's counters: the scan that found nothing counts)
        P.iter[2 * (size_t)inst + 1] = 0;
    }
    // Histogram of the violated-row counts over the batch (first solve of a controller: copra_batch_solve reads it BEFORE it launches the
    // first tier and starts on the layout of the tier's ladder that has room for the active sets to expect -- the final active set of an
    // instance is ~ 1.1 x its violated rows, correlation 0.9 over five constraint levels, DESIGN.md 3.12 -- instead of learning the layout
    // from the overflow counts of its first solves).  kLaneHistBins bins, the last one open; one ballot per bin, one atomic per non-empty bin.
    if (P.lane_hist) {
        const int bin = !more ? -1 : (nviol < kLaneHistBins - 1 ? nviol : kLaneHistBins - 1);
        for (int b = 0; b < kLaneHistBins; ++b) {
            int cnt = 0;
            (void)wave_prefix_count(bin == b, cnt);
            if (cnt > 0 && lane == 0) (void)atomic_add_i32(P.lane_hist + b, cnt);
        }
    }
    if (P.prof && lane == 0) {
        stamp[4] = cycle_counter();
        long long* pr = P.prof + 8 * (size_t)group;
        for (int q = 0; q < 4; ++q) pr[q] = stamp[q + 1] - stamp[q];
        if (fine) {
            pr[4] = fineA;
            pr[5] = fineB;
            pr[6] = fineC;
        }
        pr[7] = stamp[4] - stamp[0];
    }
}

// The same pass for the SHARED-MODEL path (copra_batch_set_shared_system; receding-horizon ticks): the batch shares (A, B, d), so the
// sweep was done once by the prepare launch (FusedPlan::ric_model: the stage records of lmpc_fused_ric.hpp) and every instance only
// rolls out from its own x0 -- with one instance per lane the records are WAVE-UNIFORM: every matrix entry is a scalar operand of the
// multiply-add (s_load, no vector registers, no LDS), a stage is 80 v_fma_f64 for 64 instances.  Rows, bounds, verdict and the
// hand-over of U and X to the first tier exactly as above.
template <int NX, int NU, bool SPEC = true>
COPRA_DEV void lmpc_lane_shared_body(const FusedPlan& P, int group)
{
    constexpr int NZ = NX + NU, RW = NZ + 2;
    using RR = RicRec<NX, NU>;
    const int lane = lane_id();
    const int inst = group * kWave + lane;
    const bool valid = inst < P.batch;
    const int NH = P.N;
    const double* tab = P.params + P.lane_tab;
    int oh_, oHN_, ohN_, oRows_;
    lane_tab_offsets(NX, NU, oh_, oHN_, ohN_, oRows_);
    const int oRows = oRows_;
    double* lds = lds_base();
    if (P.lane_zero && group == 0 && lane == 0) P.lane_zero[0] = P.lane_zero[2] = 0;
    const double* const F = P.ric_model; // N stage records + the constant block (B | d | ...)
    int oKvB_ = 0, oG_ = 0, oNb_ = 0; // the feed-forward terms kv: a block of their own behind the constant block (RicRec)
    (void)ric_model_offsets(NX, NU, NH, P.mgen, oKvB_, oG_, oNb_);
    const int oKvB = oKvB_;
    const int li = valid ? inst : 0;
    double x[NX];
#pragma unroll
    for (int c = 0; c < NX; ++c) x[c] = P.x0[(size_t)li * NX + c];
    const double vsmall = P.vsmall;
    const int rps = P.lane_rps;
    const double* const lbp = P.lb_inst ? P.lb_inst + (size_t)li * P.n : P.lb;
    const double* const ubp = P.ub_inst ? P.ub_inst + (size_t)li * P.n : P.ub;
    const bool own_bounds = P.lb_inst != nullptr;
    constexpr int GS = kLaneGroup, SX = (GS * NX) | 1, SU = (GS * NU) | 1;
    double* const ldx = lds;
    double* const ldu = lds + kWave * SX;
    const int left = P.batch - group * kWave;
    const int ninst = left < kWave ? left : kWave;
    int oHl_ = 0;
    (void)lane_lds_doubles(NX, NU, oHl_);
    double* const Tl = lds + oHl_ + ((NZ * NZ + NZ + 1) & ~1); // (the same place as in lmpc_lane_body: the rows of every step and the bounds)
    const int tl_rows = (NH + 1) * P.lane_rps * RW;
    const bool tlds = P.lane_tlds > 0;
    if (tlds) {
        for (int e = lane; e < tl_rows; e += kWave) Tl[e] = tab[oRows + e];
        for (int e = lane; e < P.n; e += kWave) {
            Tl[tl_rows + e] = P.ub[e];
            Tl[tl_rows + P.n + e] = P.lb[e];
        }
    }
    wave_sync();
    // The first steps of the active-set iteration, speculatively -- as in lmpc_lane_body (see there): bounds on u_0 as the first picks have
    // closed-form steps in W = M_uu,0^-1 = Lam_0^-T Lam_0^-1 (here a wave-uniform block of the batch-wide records), the later stages follow in
    // closed loop, kSpec more trajectories ride along.  The row norms qpgen2 normalises by are the model's (FusedPlan::ric_model, per row).
    constexpr int kSpec = !SPEC ? 0 : NU >= 2 ? 2 : 1;
    const bool spec_on = SPEC && P.lane_spec != 0;
    bool specl[kSpec + 1], uniql[kSpec + 1], violl[kSpec + 1];
    int scl[kSpec + 1];
    double sst[kSpec + 1], sst2[kSpec + 1], bmin[kSpec + 1], bmin_other[kSpec + 1];
#pragma unroll
    for (int l = 0; l <= kSpec; ++l) {
        specl[l] = false;
        uniql[l] = true;
        violl[l] = false;
        scl[l] = -1;
        sst[l] = sst2[l] = bmin[l] = bmin_other[l] = 0.0;
    }
    double xs[kSpec + 1][NX];
#pragma unroll
    for (int l = 0; l <= kSpec; ++l)
#pragma unroll
        for (int c = 0; c < NX; ++c) xs[l][c] = x[c];
    auto row_eval = [&](const double (&e)[NX], const double (&g)[NU], double f, int idx, const double (&xk)[kSpec + 1][NX],
                        const double (&uk)[kSpec + 1][NU]) {
        const double nrm = uniform_load(F, oNb_ + (idx >= 0 ? idx : 0)); // (a row that is not there: f = +inf, never violated)
        const double n2 = nrm * nrm;
#pragma unroll
        for (int l = 0; l <= kSpec; ++l) {
            double ax = 0.0;
#pragma unroll
            for (int c = 0; c < NX; ++c) ax += e[c] * xk[l][c];
#pragma unroll
            for (int c = 0; c < NU; ++c) ax += g[c] * uk[l][c];
            const double sl = f - ax;
            const bool v = sl <= -vsmall;
            violl[l] = violl[l] || v;
            if (l < kSpec) uniql[l + 1] = uniql[l + 1] && !(v && sl * sl >= (sst2[l + 1] * n2) * (1.0 - 1e-9));
        }
    };
    auto check_rows = [&](int k, const double (&xk)[kSpec + 1][NX], const double (&uk)[kSpec + 1][NU]) {
        if (tlds) {
            for (int r = 0; r < rps; ++r) {
                const double* const rt = Tl + (k * rps + r) * RW;
                double e[NX], g[NU];
#pragma unroll
                for (int c = 0; c < NX; ++c) e[c] = rt[c];
#pragma unroll
                for (int c = 0; c < NU; ++c) g[c] = rt[NX + c];
                row_eval(e, g, rt[NZ], uniform_i32((int)rt[NZ + 1]), xk, uk);
            }
            return;
        }
        for (int r = 0; r < rps; ++r) {
            const int ro = oRows + (k * rps + r) * RW;
            double e[NX], g[NU];
#pragma unroll
            for (int c = 0; c < NX; ++c) e[c] = uniform_load(tab, ro + c);
#pragma unroll
            for (int c = 0; c < NU; ++c) g[c] = uniform_load(tab, ro + NX + c);
            row_eval(e, g, uniform_load(tab, ro + NZ), (int)uniform_load(tab, ro + NZ + 1), xk, uk);
        }
    };
    // Per-instance cost references (copra_batch_set_cost_reference: one model, every instance its own goal or reference trajectory).  The
    // batch-wide records were swept with the controller-wide references p0; the recursion is AFFINE in the references, so this instance's
    // feed-forward terms are  kv_k + dkv_k  with the DELTA sweep over the same records, no cost-to-go matrix needed:
    //     q = dh_u + B' dpv+,   dkv_k = -Lam_k^-1 q  (Lam^-1 = Li' Li, RicRec::oLi),   dpv_k = dh_x + K_k' dh_u + Acl_k' dpv+,   dpv_N = dhN,
    // dh = sum_t sum_r c_t(r) (p_t[r] - p0_t[r]) from the plan builder's coefficient table (lane_cref, as lmpc_lane_body).  About 80
    // multiply-adds per stage with scalar operands -- the roll-out's cost once more.  dkv waits in the pass's workspace, lane-major.
    bool own_refs = false, own_traj = false;
    for (int t = 0; t < P.ncost; ++t) {
        own_refs = own_refs || P.cost_p[t] != nullptr;
        own_traj = own_traj || (P.cost_p[t] != nullptr && P.cost[t].pstride != 0);
    }
    own_refs = own_refs && P.lane_ws != nullptr && P.lane_cref >= 0;
    double* const dkw = P.lane_ws;
    const size_t dbp = (size_t)P.lane_bp;
    if (own_refs) {
        auto delta_h = [&](int k, bool terminal, double (&dh)[NZ], double (&dhN)[NX]) {
#pragma unroll
            for (int a = 0; a < NZ; ++a) dh[a] = 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) dhN[i] = 0.0;
            for (int t = 0; t < P.ncost; ++t) {
                if (!P.cost_p[t]) continue;
                const CostTerm& ct = P.cost[t];
                const int ps = ct.pstride;
                // (the terminal term takes the reference of the last step; a cost without a step k has zero coefficients there)
                const int kk = terminal ? (ps ? ct.prows / ps - 1 : 0) : ((ps && (k + 1) * ps > ct.prows) ? ct.prows / ps - 1 : k);
                const double* const pi = P.cost_p[t] + (size_t)li * ct.prows + kk * ps;
                const double* const p0 = P.params + ct.offP + kk * ps;
                for (int r = 0; r < ct.rows; ++r) {
                    const double dp = pi[r] - uniform_load(p0, r);
                    const int co = P.lane_cref + (t * 6 + r) * (NZ + NX);
#pragma unroll
                    for (int a = 0; a < NZ; ++a) dh[a] += uniform_load(tab, co + a) * dp;
#pragma unroll
                    for (int i = 0; i < NX; ++i) dhN[i] += uniform_load(tab, co + NZ + i) * dp;
                }
            }
        };
        double dh[NZ], dpv[NX], unused[NX];
        delta_h(0, true, dh, dpv); // dpv_N = dhN (dh: the stage term with the last reference -- replaced per stage for trajectories)
        if (!own_traj) delta_h(0, false, dh, unused);
        for (int k = NH - 1; k >= 0; --k) {
            if (own_traj) delta_h(k, false, dh, unused);
            const int rb = k * RR::SZ;
            double qu[NU], tt[NU];
#pragma unroll
            for (int c = 0; c < NU; ++c) {
                double s2 = dh[NX + c];
#pragma unroll
                for (int i = 0; i < NX; ++i) s2 += uniform_load(F, NH * RR::SZ + RR::cB + i + NX * c) * dpv[i];
                qu[c] = s2;
            }
#pragma unroll
            for (int r = 0; r < NU; ++r) {
                double s2 = 0.0;
#pragma unroll
                for (int c = 0; c <= r; ++c) s2 += uniform_load(F, rb + RR::oLi + r * (r + 1) / 2 + c) * qu[c];
                tt[r] = s2;
            }
#pragma unroll
            for (int c = 0; c < NU; ++c) {
                double s2 = 0.0;
#pragma unroll
                for (int r = c; r < NU; ++r) s2 += uniform_load(F, rb + RR::oLi + r * (r + 1) / 2 + c) * tt[r];
                dkw[((size_t)k * NU + c) * dbp + inst] = -s2;
            }
            double dn[NX];
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                double s2 = dh[j];
#pragma unroll
                for (int c = 0; c < NU; ++c) s2 += uniform_load(F, rb + RR::oK + c + NU * j) * dh[NX + c];
#pragma unroll
                for (int i = 0; i < NX; ++i) s2 += uniform_load(F, rb + RR::oAcl + i + NX * j) * dpv[i];
                dn[j] = s2;
            }
#pragma unroll
            for (int j = 0; j < NX; ++j) dpv[j] = dn[j];
        }
    }
    for (int k0 = 0; k0 < NH; k0 += GS) {
        wave_sync();
#pragma unroll
        for (int q = 0; q < GS; ++q) {
            const int k = k0 + q;
            const bool on = k < NH;
            const int kk = on ? k : NH - 1;
            const int rb = kk * RR::SZ;
            double us[kSpec + 1][NU], kvk[NU]; // kvk: this instance's feed-forward term (the records' + its own delta)
#pragma unroll
            for (int c = 0; c < NU; ++c) {
                kvk[c] = uniform_load(F, oKvB + kk * NU + c);
                if (own_refs) kvk[c] += dkw[((size_t)kk * NU + c) * dbp + inst];
            }
#pragma unroll
            for (int l = 0; l <= kSpec; ++l)
#pragma unroll
                for (int c = 0; c < NU; ++c) {
                    double acc = kvk[c];
#pragma unroll
                    for (int j = 0; j < NX; ++j) acc += uniform_load(F, rb + RR::oK + c + NU * j) * xs[l][j];
                    us[l][c] = acc;
                }
            double ubk[NU], lbk[NU];
#pragma unroll
            for (int c = 0; c < NU; ++c) { // (three separate paths: a pointer chosen between LDS and memory would make these flat loads)
                const int kc = kk * NU + c;
                if (own_bounds) {
                    ubk[c] = ubp[kc];
                    lbk[c] = lbp[kc];
                } else if (tlds) {
                    ubk[c] = Tl[tl_rows + kc];
                    lbk[c] = Tl[tl_rows + P.n + kc];
                } else {
                    ubk[c] = uniform_load(ubp, kc);
                    lbk[c] = uniform_load(lbp, kc);
                }
            }
            double du[kSpec + 1][NU]; // what the speculative steps add to u_0 (zero at every other stage): x+ = Acl x + B (kv + du) + d
#pragma unroll
            for (int l = 0; l <= kSpec; ++l)
#pragma unroll
                for (int c = 0; c < NU; ++c) du[l][c] = 0.0;
            if (q == 0 && k0 == 0 && spec_on) {
                double W[NU][NU]; // M_uu,0^-1 = Lam^-T Lam^-1 of the batch-wide records (Lam^-1 lower triangular, packed by rows)
#pragma unroll
                for (int i = 0; i < NU; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) {
                        double acc = 0.0;
#pragma unroll
                        for (int r = i; r < NU; ++r) acc += uniform_load(F, RR::oLi + r * (r + 1) / 2 + i) * uniform_load(F, RR::oLi + r * (r + 1) / 2 + j);
                        W[i][j] = W[j][i] = acc;
                    }
                double u00[NU];
#pragma unroll
                for (int c = 0; c < NU; ++c) u00[c] = us[0][c];
                if constexpr (kSpec > 0) lane_spec_steps<NU, kSpec>(W, ubk, lbk, us, vsmall, false, specl, scl, sst, sst2);
#pragma unroll
                for (int l = 1; l <= kSpec; ++l)
#pragma unroll
                    for (int c = 0; c < NU; ++c) du[l][c] = us[l][c] - u00[c];
            }
            if (on) {
                check_rows(k, xs, us);
                const bool stage0 = q == 0 && k0 == 0;
#pragma unroll
                for (int c = 0; c < NU; ++c) {
#pragma unroll
                    for (int l = 0; l <= kSpec; ++l) { // (the worst bound slack per level: see lmpc_lane_body)
                        const double m = fmin(ubk[c] - us[l][c], us[l][c] - lbk[c]);
                        const bool active = stage0 && ((l >= 1 && c == scl[kSpec >= 1 ? 1 : 0]) || (l >= 2 && c == scl[kSpec]));
                        const bool pick = stage0 && l < kSpec && c == scl[l < kSpec ? l + 1 : kSpec];
                        bmin[l] = fmin(bmin[l], active ? 0.0 : m);
                        if (l < kSpec) bmin_other[l] = fmin(bmin_other[l], (active || pick) ? 0.0 : m);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NX; ++c) ldx[lane * SX + q * NX + c] = xs[kSpec][c]; // (the deepest trajectory: levels without a step repeat the one before)
#pragma unroll
            for (int c = 0; c < NU; ++c) ldu[lane * SU + q * NU + c] = us[kSpec][c];
            // x+ = Acl x + B (kv + du) + d  (= A x + B u + d: the tier's own roll-out, ric_factor.hpp)
#pragma unroll
            for (int l = 0; l <= kSpec; ++l) {
                double xn[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    double acc = uniform_load(F, NH * RR::SZ + RR::cD + i);
#pragma unroll
                    for (int j = 0; j < NX; ++j) acc += uniform_load(F, rb + RR::oAcl + i + NX * j) * xs[l][j];
#pragma unroll
                    for (int c = 0; c < NU; ++c) acc += uniform_load(F, NH * RR::SZ + RR::cB + i + NX * c) * (kvk[c] + du[l][c]);
                    xn[i] = acc;
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) xs[l][i] = on ? xn[i] : xs[l][i];
            }
        }
        wave_sync();
        const int nst = NH - k0 < GS ? NH - k0 : GS;
        lane_group_out<GS * NX>(P.trajectory + (size_t)(group * kWave) * P.X + (size_t)k0 * NX, (size_t)P.X, ldx, SX, ninst, nst * NX, lane);
        lane_group_out<GS * NU>(P.control + (size_t)(group * kWave) * P.n + (size_t)k0 * NU, (size_t)P.n, ldu, SU, ninst, nst * NU, lane);
    }
    {
        double u0[kSpec + 1][NU];
#pragma unroll
        for (int l = 0; l <= kSpec; ++l)
#pragma unroll
            for (int c = 0; c < NU; ++c) u0[l][c] = 0.0;
        check_rows(NH, xs, u0);
        if (valid) {
            double* const xo = P.trajectory + (size_t)inst * P.X + (size_t)NH * NX;
#pragma unroll
            for (int c = 0; c < NX; ++c) xo[c] = xs[kSpec][c];
        }
    }
    // the verdict: the first level that violates nothing, if every pick on the way to it was the pick (lmpc_lane_body)
#pragma unroll
    for (int l = 0; l <= kSpec; ++l) {
        violl[l] = violl[l] || (bmin[l] <= -vsmall);
        if (l < kSpec) uniql[l + 1] = uniql[l + 1] && !(bmin_other[l] <= -vsmall && bmin_other[l] <= sst[l + 1] * (1.0 - 1e-9));
    }
    int done_iters = 0;
    {
        bool chain = valid;
#pragma unroll
        for (int l = 0; l <= kSpec; ++l) {
            if (l >= 1) chain = chain && specl[l] && uniql[l];
            if (chain && !violl[l] && done_iters == 0) done_iters = l + 1;
            chain = chain && violl[l];
        }
    }
    const bool more = valid && done_iters == 0;
    int total = 0;
    const int before = wave_prefix_count(more, total);
    if (total > 0) {
        int base = 0;
        if (lane == 0) base = atomic_add_i32(P.lane_count, total);
        base = bcast_i32(base, 0);
        if (more) P.lane_list[base + before] = inst;
    }
    {
        int nspec = 0;
        (void)wave_prefix_count(done_iters >= 2, nspec);
        if (nspec > 0 && lane == 0) (void)atomic_add_i32(P.lane_count + 2, nspec);
    }
    if (done_iters > 0) {
        P.status[inst] = 0;
        P.iter[2 * (size_t)inst] = done_iters; // (qpgen2's counters: the scan that found nothing counts)
        P.iter[2 * (size_t)inst + 1] = 0;
    }
}

} // namespace copra_hip
