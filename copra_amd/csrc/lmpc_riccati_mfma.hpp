// lmpc_riccati_mfma.hpp -- the stage-wise Riccati interior-point method of lmpc_riccati.hpp with the iterate RESIDENT on the
// CU and the stage algebra on the matrix cores (BASELINE config 5: InitialStateLMPC, nx = 12, nu = 6, N = 50).
//
// What it replaces: the same call stack as lmpc_riccati.hpp (LMPC::solve, src/LMPC.cpp:79-101, for controllers whose pieces are
// stage-wise -- stage_plan.hpp -- incl. InitialStateLMPC::makeQPForm, src/InitialStateLMPC.cpp:77-122).  Same algorithm (Mehrotra
// predictor-corrector, Riccati recursion per Newton system), same optimum; what changes is where the data lives and how
// a stage is computed:
//   * lmpc_riccati.hpp streams a 126 KB per-wave workspace (iterate, multipliers, stage gains) through HBM on every one of its
//     four sweeps per Newton step: 6 MB of traffic per solve for 9 KB of algorithmic I/O, and every stage waits on it.  Here
//     NOTHING of an instance leaves the CU between its first load and its last store:
//       registers (static indices, 64 rows per register):  right-hand sides, slacks, multipliers, residuals, the predictor's
//                 ds * dl -- everything that is only touched row by row ("bulk" phases, all 64 lanes busy, instead of stage by
//                 stage with 18 lanes);
//       LDS (<= 80 KB, two instances per CU):  the iterate z, the stage gains of the current factorisation (99 doubles per
//                 stage), one buffer for the row weights D (backward sweep) and the step dz (forward sweeps), one for the
//                 gradient coefficients of the rows, the blocks in flight.
//   * the 18 x 18 stage (T = P [A B], M = H + [A B]' T, elimination of the controls, P = Schur complement) runs on
//     v_mfma_f64_4x4x4 in the lane layout lmpc_fused_ric.hpp found for the headline kernel (lane = 16 q + 4 b + r holds
//     A_b[i = r][k = q], B_b[k = q][j = r], D_b[i = q][j = r]: a result is laid out like the B operand of the next product):
//     hardware block b = column block of the result (x_0..3 | x_4..7 | x_8..11 | the gradient column; second pass: the two
//     control blocks), one accumulator per row block; the six controls are eliminated as two blocks of three by adjugate /
//     determinant (one reciprocal each, every lane its own cofactor -- the headline's 3 x 3 idiom twice), which keeps the
//     6 x 6 inverse off the chain.  50 MFMAs per stage; three LDS hand-overs (P as the left factor of T, the rows u_b and u_a
//     of M as left factors of the two Schur updates).
//   * the three vector sweeps (two forward, one backward) keep their state in the accumulator layout as well: 8 - 10 MFMAs
//     and three or four DPP row broadcasts per stage, operands from the stage records in LDS.
//
// Shapes: compiled once for three blocks of four states and two blocks of three controls; smaller systems are padded by the
// stage plan (zero rows / columns of A, B; unit diagonal for controls that do not exist), larger ones and controllers whose
// rows do not fit the fixed-width tables (stage_plan.hpp: fast_ok) stay on lmpc_riccati.hpp.
#pragma once

#include "plan.hpp"
#include "stage_plan.hpp"
#include "wave_prims.hpp"
#include "lmpc_riccati.hpp" // ric_converged

namespace copra_hip {

struct RfLds {
    double *X, *Y, *Cb, *KF, *Hb, *Pb, *Rb, *Ra, *H0, *G0, *AB, *dv, *GJ, *pv0, *dx0, *Zs, *gk, *TT;
};

constexpr int kRfHStride = kRfNZ * kRfNZ + 2; // one copy of the stage Hessian + a zero pad (what lanes without an entry read)

COPRA_DEV RfLds carve_rf(double* lds, int N, int ntmpl)
{
    RfLds L;
    double* p = lds;
    L.X = p, p += 64 * kRfZR; // the iterate z = (x_k, u_k)_k, padded
    L.Y = p, p += 64 * kRfMR; // row weights D (backward factorisation) | the step dz (forward sweeps)
    L.Cb = p, p += 64 * kRfMR; // gradient coefficients c of the rows (g_k += A_k' c)
    L.KF = p, p += kRfRing * kRfRingStride; // a ring of kRfRing stage records (record k in slot k % kRfRing while a sweep passes it)
    L.Hb = p, p += 2 * kRfHStride; // stage Hessian H = W + A' D A of the stage in flight | of the next one (being assembled)
    L.Pb = p, p += kRfNX * kRfNX;
    L.Rb = p, p += 64;
    L.Ra = p, p += 64;
    L.H0 = p, p += kRfNX * kRfNX;
    L.G0 = p, p += 16;
    L.AB = p, p += kRfNX * kRfNZ;
    L.dv = p, p += 16;
    L.GJ = p, p += kRfNX * (kRfNX + 1);
    L.pv0 = p, p += 12;
    L.dx0 = p, p += 12;
    L.Zs = p, p += 32; // 16 zeros (absent operands) | 16 doubles nobody reads (stores of lanes with nothing to store)
    L.gk = p, p += 2 * 20; // gradient of the stage in flight | of the next one (18 entries + a zero pad each)
    L.TT = p, p += 2 * ntmpl; // the (at most two) coefficients of every row template
    return L; // (stage_plan.hpp: fast_lds_doubles)
}

COPRA_DEV double rf_rcp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    y = fma(fma(-x, y, 1.0), y, y);
    return y;
#else
    return 1.0 / x;
#endif
}

enum { kRfIneq = 0, kRfEq = 1, kRfOff = 2 };

// the four values are needed at this point TOGETHER: the compiler then issues their loads back to back and waits once (left to
// itself it interleaved every pair of LDS reads of the stage gradient with a full wait: four round trips instead of one)
#if defined(__HIP_DEVICE_COMPILE__)
#define COPRA_RF_TOGETHER4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#else
#define COPRA_RF_TOGETHER4(a, b, c, d) ((void)0)
#endif

COPRA_DEV void lmpc_riccati_mfma_body(const FusedPlan& P, const StagePlan& S)
{
    constexpr int NX = kRfNX, NZ = kRfNZ, MR = kRfMR, KS = kRfKStride;
    // stage record: K_a 3 x 12 | K_b 3 x 12 | K_ba 3 x 3 | -M_bb^-1 | -M'_aa^-1 | kv_a | kv_b | spare (dummy stores) | zero (absent operands)
    constexpr int oKa = 0, oKb = 36, oKba = 72, oNb = 81, oNa = 90, oKva = 99, oKvb = 102, oSpare = 105, oZero = 106;
    const int lane = lane_id();
    const int nx = S.nx, nu = S.nu, N = S.N, m = S.m;
    const int NE = (N + 1) * NZ; // entries of the padded stage vectors
    const RfLds L = carve_rf(lds_base(), N, S.fast_ntmpl);
    const double delta = S.delta;
    const double BIGF = 1e299;
    // class and first row of stage `lane` (N + 1 <= 53 stages): one v_readlane per stage instead of two scalar loads from memory
    const int stinfo = lane <= N ? (S.cls_of_stage[lane] | (S.stage_row0[lane] << 8)) : 0;

    for (int witem = instance_id();; witem += instance_stride()) {
        int inst = witem;
        if (S.next_instance) {
            int v = 0;
            if (lane == 0) v = atomic_append(S.next_instance);
            inst = bcast_i32(v, 0);
        }
        if (inst >= P.batch) break;
        long long prof[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        long long tprev = P.prof ? cycle_counter() : 0;
        const long long tstart = tprev;
        auto stamp = [&](int slot) {
            if (P.prof) {
                const long long t = cycle_counter();
                prof[slot] += t - tprev;
                tprev = t;
            }
        };
        const bool x0_free = S.x0_free && P.x0lb && P.x0ub;
#ifdef COPRA_RF_FINE // (development: the stamps go INSIDE the stage of sweep 1 -- slot s = time up to stamp s of a stage, summed)
#if COPRA_RF_FINE == 2 // ... or inside prepare_stage: slot 0 = the stage up to it, 1 = class check / refill, 2 = touched entries, 3 = gradient, 4 = rest
#define COPRA_RF_STAMP(s) ((s) == 2 ? stamp(0) : (s) == 6 ? stamp(4) : (void)0)
#define COPRA_RF_PSTAMP(s) stamp(s)
#else
#define COPRA_RF_STAMP(s) stamp(s)
#define COPRA_RF_PSTAMP(s) ((void)0)
#endif
#define stamp_outer(s) ((void)0)
#else
#define COPRA_RF_STAMP(s) ((void)0)
#define COPRA_RF_PSTAMP(s) ((void)0)
#define stamp_outer(s) stamp(s)
#endif
        // ------------------------------------------------------------------ 0. this instance's system, padded
        for (int e = lane; e < NX * NZ; e += kWave) {
            const int j = e / NX, i = e - j * NX;
            double v = 0.0;
            if (j < NX) {
                if (i < nx && j < nx) v = P.A[(size_t)inst * nx * nx + i + nx * j];
            } else {
                if (i < nx && j - NX < nu) v = P.B[(size_t)inst * nx * nu + i + nx * (j - NX)];
            }
            L.AB[e] = v;
        }
        if (lane < 16) L.dv[lane] = lane < nx ? P.d[(size_t)inst * nx + lane] : 0.0;
        if (lane < 32) L.Zs[lane] = 0.0;
        if (lane < 4) L.Hb[NZ * NZ + (lane & 1) + kRfHStride * (lane >> 1)] = 0.0, L.gk[18 + (lane & 1) + 20 * (lane >> 1)] = 0.0;
        if (lane < kRfRing) L.KF[lane * kRfRingStride + oSpare] = 0.0, L.KF[lane * kRfRingStride + oZero] = 0.0; // (... of every slot: they travel with the record)
        for (int e = lane; e < 2 * S.fast_ntmpl; e += kWave) L.TT[e] = S.f_rval[e];
        wave_sync();
        const double* const AB = L.AB;
        // ---- the stage records (round 5).  N x 107 doubles = 42.8 KB of the 77.6 KB of LDS an instance held -- two instances per CU, two of the
        //      four SIMDs idle.  Every sweep passes the records strictly in order, so only a ring of kRfRing of them stays in LDS: the
        //      factorisation writes record k into slot k % 4 and copies it out to this wave's workspace one stage later; the vector sweeps
        //      request record k -+ 3 (two 8-byte loads per lane) while they work on stage k and drop it into its slot one stage later.
        //      38.7 KB per instance: FOUR per CU, one per SIMD.
        double* const rec = S.rec_ws + (size_t)instance_id() * (size_t)N * KS;
        constexpr int RS = kRfRingStride;
        auto rec_slot = [&](int k) -> double* { return L.KF + (k & (kRfRing - 1)) * RS; };
        auto rec_copy_out = [&](int k) { // (after a wave_sync that follows the last write to the slot)
            const double* const sl = rec_slot(k);
            rec[(size_t)k * KS + lane] = sl[lane];
            if (lane < KS - kWave) rec[(size_t)k * KS + kWave + lane] = sl[kWave + lane];
        };
        auto rec_request = [&](int k, double& g0, double& g1) { // (no wait: the values are used one stage later)
            const bool in = k >= 0 && k < N;
            g0 = in ? rec[(size_t)k * KS + lane] : 0.0;
            g1 = (in && lane < KS - kWave) ? rec[(size_t)k * KS + kWave + lane] : 0.0;
        };
        auto rec_put = [&](int k, double g0, double g1) {
            if (k < 0 || k >= N) return;
            double* const sl = rec_slot(k);
            sl[lane] = g0;
            if (lane < KS - kWave) sl[kWave + lane] = g1;
        };
        // (the operands of the matrix products that are constants of the instance -- the element of [A B] each lane hands in -- are
        //  read from this copy at the head of every sweep: registers are what limits this kernel)

        // ------------------------------------------------------------------ per-row state (registers: row 64 j + lane in element j)
        int rinf[MR]; // stage | template << 6 | components of its two coefficients << 15, << 20 | flag << 28
        double Fr[MR], Sv[MR], Lam[MR], Dd[MR];
        // right-hand sides (Cb serves as the exchange buffer for the flags below)
#pragma unroll
        for (int j = 0; j < MR; ++j) {
            const int gi = 64 * j + lane;
            int info = 0;
            double f = BIGF;
            if (gi < m) {
                info = S.f_rinfo[gi];
                const int k = info & 255, t = info >> 8, idx = S.r_sidx[t] + S.r_sstride[t] * k;
                switch (S.r_src[t]) {
                case kSrcRowF: f = P.row_f_inst ? P.row_f_inst[(size_t)inst * P.mgen + idx] : P.row_f[idx]; break;
                case kSrcUb: f = P.ub_inst ? P.ub_inst[(size_t)inst * P.n + idx] : P.ub[idx]; break;
                case kSrcNegLb: f = -(P.lb_inst ? P.lb_inst[(size_t)inst * P.n + idx] : P.lb[idx]); break;
                case kSrcX0Ub: f = x0_free ? P.x0ub[(size_t)inst * nx + idx] : BIGF; break;
                default: f = x0_free ? -P.x0lb[(size_t)inst * nx + idx] : BIGF; break;
                }
                L.Cb[gi] = f;
                const int c0 = S.f_rcomp[2 * t], c1 = S.f_rcomp[2 * t + 1];
                info = k | (t << 6) | ((c0 < 0 ? 0 : c0) << 15) | ((c1 < 0 ? 0 : c1) << 20);
            }
            rinf[j] = info;
            Fr[j] = f, Sv[j] = 1.0, Lam[j] = 0.0, Dd[j] = 0.0;
            if (j % 5 == 4) sched_fence();
        }
        wave_sync();
        // flags; a bound pair  lb == ub  (up to rounding) is one equality row: the upper row becomes it, the lower row is off
        auto base_flag = [&](int gi) -> int {
            const int t = S.f_rinfo[gi] >> 8;
            return (L.Cb[gi] >= BIGF) ? kRfOff : (S.r_eq[t] ? kRfEq : kRfIneq);
        };
        auto pinned = [&](int gu) -> bool { // rows (gu, gu + 1): upper and lower bound of one component, equal
            if (gu < 0 || gu + 1 >= m) return false;
            const int iu = S.f_rinfo[gu], il = S.f_rinfo[gu + 1];
            if ((iu & 255) != (il & 255)) return false;
            const int tu = iu >> 8, tl = il >> 8;
            if (S.r_kind[tu] != 1 || S.r_kind[tl] != 1 || S.r_aoff[tu] != S.r_aoff[tl]) return false;
            if (!(S.r_sign[tu] == 1.0 && S.r_sign[tl] == -1.0)) return false;
            if (base_flag(gu) != kRfIneq || base_flag(gu + 1) != kRfIneq) return false;
            const double up = L.Cb[gu], lo = -L.Cb[gu + 1];
            return up - lo <= 1e-12 * fmax(1.0, fabs(up));
        };
        int n_ineq = 0;
#pragma unroll
        for (int j = 0; j < MR; ++j) {
            const int gi = 64 * j + lane;
            int fl = kRfOff;
            if (gi < m) {
                fl = base_flag(gi);
                if (pinned(gi))
                    fl = kRfEq;
                else if (pinned(gi - 1))
                    fl = kRfOff;
            }
            rinf[j] |= fl << 28;
            n_ineq += fl == kRfIneq ? 1 : 0;
            if (j % 3 == 2) sched_fence();
        }
        n_ineq = (int)(wave_sum((double)n_ineq) + 0.5);
        const double inv_mi = n_ineq > 0 ? 1.0 / (double)n_ineq : 0.0;

        // ---- helpers ------------------------------------------------------------------------------------------------
        // a_gi' v for this lane's row (v: a padded stage-vector buffer in LDS)
        auto row_dot = [&](int info, const double* v) -> double {
            const int t = (info >> 6) & 511;
            const double* vk = v + (info & 63) * NZ;
            return L.TT[2 * t] * vk[(info >> 15) & 31] + L.TT[2 * t + 1] * vk[(info >> 20) & 31]; // (absent coefficients carry value 0)
        };
        // x_{k+1} = A x_k + B u_k + d along X (the controls as stored in X), from the x_0 stored in X[0 .. 12)
        auto rollout = [&]() {
            for (int k = 0; k < N; ++k) {
                wave_sync();
                double acc = 0.0;
                if (lane < NX) {
                    acc = L.dv[lane];
#pragma unroll
                    for (int j = 0; j < NZ; ++j) acc += AB[lane + NX * j] * L.X[k * NZ + j];
                }
                if (lane < NX) L.X[(k + 1) * NZ + lane] = acc;
            }
            wave_sync();
        };
        // ---- the sweeps.  Every address a lane uses in their stage loops is an OFFSET from the LDS base, worked out at the head of the
        //      sweep from the lane's position (q, b, r) in the v_mfma_f64_4x4x4 layout -- no selects, no predicates in the loops: a
        //      lane with nothing to read reads the zero pad, a lane with nothing to write writes the spare slot.
        //      (The first version of this kernel selected pointers inside the loops: ~ 50 lane predicates alive at once = 100 scalar
        //       registers, which the compiler spilled and reloaded, 51 v_readlane per stage.)
        double* const lds = lds_base();
        const int oY = (int)(L.Y - lds), oHb = (int)(L.Hb - lds);
        const int oPb = (int)(L.Pb - lds), oRb = (int)(L.Rb - lds), oRa = (int)(L.Ra - lds), oGk = (int)(L.gk - lds);
        const int oZ0 = (int)(L.Zs - lds), oDU = oZ0 + 16, oPv0 = (int)(L.pv0 - lds);
        auto lane_qbr = [&](int& q, int& b, int& r) { // (opaque: what is derived from it is worked out HERE, not ahead of the Newton loop)
            int ln = lane;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(ln));
#endif
            q = ln >> 4, b = (ln >> 2) & 3, r = ln & 3;
        };
        // this lane's element (r, q) of -inverse of the symmetric 3 x 3 block m(a, b) = base[20 a + b]: adjugate over determinant,
        // ONE reciprocal (every lane computes the determinant from six wave-uniform reads and ITS cofactor from four reads at its
        // own offsets a1 .. a4 -- all zero outside the block, which makes the result zero there).  Positive definite <=> m22, C00,
        // det > 0 (Sylvester).
        auto neg_inv3 = [&](const double* base, int a1, int a2, int a3, int a4, bool& bad) -> double {
            const double m00 = base[0], m01 = base[1], m02 = base[2], m11 = base[21], m12 = base[22], m22 = base[42];
            const double x1 = base[a1], x2 = base[a2], x3 = base[a3], x4 = base[a4];
            const double c00 = m11 * m22 - m12 * m12, c01 = m12 * m02 - m01 * m22, c02 = m01 * m12 - m11 * m02;
            const double det = m00 * c00 + (m01 * c01 + m02 * c02);
            bad = bad || !(m22 > 0.0) || !(c00 > 0.0) || !(det > 0.0);
            return (x1 * x2 - x3 * x4) * (-rf_rcp(det));
        };

        // ---- tables of the stage class being PREPARED (registers): the entries of H = W + A' D A the rows touch (one per lane), and
        //      for the lanes 0 .. 17 the non-zeros of row `lane` of W and the rows that touch component `lane`.  The stage loops run
        //      class by class (inner loops over the stages of one class), so that these are loop constants there.
        int t_ent = 0, t_r0 = 0, t_r1 = 0, t_r2 = 0, t_r3 = 0, t_cnt = 0;
        double t_w = 0.0, t_v0 = 0.0, t_v1 = 0.0, t_v2 = 0.0, t_v3 = 0.0;
        int w_c0 = 0, w_c1 = 0, w_c2 = 0, w_c3 = 0, g_r0 = 0, g_r1 = 0, g_r2 = 0, g_r3 = 0;
        double w_v0 = 0.0, w_v1 = 0.0, w_v2 = 0.0, w_v3 = 0.0, g_v0 = 0.0, g_v1 = 0.0, g_v2 = 0.0, g_v3 = 0.0;
        int hb_cls0 = -1, hb_cls1 = -1; // class whose padded W the two copies of the stage Hessian start from
        double q_cls = 0.0;
        auto load_class = [&](int c, int kfirst) { // kfirst: any stage of the class (where its q is read from)
            const double* Wp = S.blob + S.f_Wp[c];
            const int t0 = S.f_tptr[c];
            t_cnt = S.f_tptr[c + 1] - t0;
            t_ent = 0, t_w = 0.0;
            if (lane < t_cnt) {
                const int t = t0 + lane;
                t_ent = S.f_tent[t];
                t_w = Wp[t_ent];
                t_r0 = S.f_trow[4 * t], t_r1 = S.f_trow[4 * t + 1], t_r2 = S.f_trow[4 * t + 2], t_r3 = S.f_trow[4 * t + 3];
                t_v0 = S.f_tval[4 * t], t_v1 = S.f_tval[4 * t + 1], t_v2 = S.f_tval[4 * t + 2], t_v3 = S.f_tval[4 * t + 3];
                t_r0 = t_r0 < 0 ? 0 : t_r0, t_r1 = t_r1 < 0 ? 0 : t_r1, t_r2 = t_r2 < 0 ? 0 : t_r2, t_r3 = t_r3 < 0 ? 0 : t_r3;
            }
            if (lane < NZ) {
                q_cls = S.f_q[kfirst * NZ + lane];
                const size_t at = ((size_t)c * NZ + lane) * 4;
                w_c0 = S.f_wcol[at], w_c1 = S.f_wcol[at + 1], w_c2 = S.f_wcol[at + 2], w_c3 = S.f_wcol[at + 3];
                w_v0 = S.f_wval[at], w_v1 = S.f_wval[at + 1], w_v2 = S.f_wval[at + 2], w_v3 = S.f_wval[at + 3];
                w_c0 = w_c0 < 0 ? 0 : w_c0, w_c1 = w_c1 < 0 ? 0 : w_c1, w_c2 = w_c2 < 0 ? 0 : w_c2, w_c3 = w_c3 < 0 ? 0 : w_c3;
                g_r0 = S.f_grow[at], g_r1 = S.f_grow[at + 1], g_r2 = S.f_grow[at + 2], g_r3 = S.f_grow[at + 3];
                g_v0 = S.f_gval[at], g_v1 = S.f_gval[at + 1], g_v2 = S.f_gval[at + 2], g_v3 = S.f_gval[at + 3];
            }
        };
        // everything of stage k that does not depend on the cost-to-go: its Hessian H = W + A' D A (the entries the rows touch; the
        // copy of the stage's parity starts from the padded W of the class) and its gradient
        //     g = W z + q (+ the InitialStateLMPC terms at stage 0) (+ A' c, c in Cb)  ->  gk[k & 1],
        // q_k = - sum_rows w p a (costFunctions.cpp: the p-dependent part of the gradient; references shared by the batch) from the
        // plan's table, fetched one stage ahead.  Runs one stage AHEAD of the sweep, in the shadow of its matrix products; the
        // class tables above must be those of stage k's class `c`.
        double qnext = 0.0;
        const bool q_uniform = S.f_q_uniform != 0; // (per-step references: q is a constant of the class, held with its tables)
        auto prepare_stage = [&](int k, int c, int gi0, bool hessian, bool with_rows, bool with_x0_terms) {
            const double qk = q_uniform ? q_cls : qnext;
            if (!q_uniform) qnext = (lane < NZ && k > 0) ? S.f_q[(k - 1) * NZ + lane] : 0.0;
            if (hessian) {
                double* Hk = L.Hb + kRfHStride * (k & 1);
                int& have = (k & 1) ? hb_cls1 : hb_cls0;
                if (have != c) {
                    have = c;
                    const double* Wp = S.blob + S.f_Wp[c];
                    for (int e = lane; e < NZ * NZ; e += kWave) Hk[e] = Wp[e];
                    wave_sync();
                }
                COPRA_RF_PSTAMP(1);
                if (with_rows && lane < t_cnt) {
                    const double* Yk = L.Y + gi0;
                    double y0 = Yk[t_r0], y1 = Yk[t_r1], y2 = Yk[t_r2], y3 = Yk[t_r3];
                    COPRA_RF_TOGETHER4(y0, y1, y2, y3); // (ONE round trip: all four reads in flight before the first is used)
                    Hk[t_ent] = t_w + ((t_v0 * y0 + t_v1 * y1) + (t_v2 * y2 + t_v3 * y3));
                }
                COPRA_RF_PSTAMP(2);
            }
            if (lane < NZ) {
                const double* Xk = L.X + k * NZ;
                const double* Ck = L.Cb + (with_rows ? gi0 : 0);
                double x0v = Xk[w_c0], x1v = Xk[w_c1], x2v = Xk[w_c2], x3v = Xk[w_c3];
                double c0v = Ck[g_r0], c1v = Ck[g_r1], c2v = Ck[g_r2], c3v = Ck[g_r3];
                COPRA_RF_TOGETHER4(x0v, x1v, x2v, x3v);
                COPRA_RF_TOGETHER4(c0v, c1v, c2v, c3v);
                double g = qk + ((w_v0 * x0v + w_v1 * x1v) + (w_v2 * x2v + w_v3 * x3v));
                if (with_rows) g += (g_v0 * c0v + g_v1 * c1v) + (g_v2 * c2v + g_v3 * c3v);
                if (with_x0_terms && k == 0 && lane < NX) {
                    g += L.G0[lane];
#pragma unroll
                    for (int l = 0; l < NX; ++l) g += L.H0[lane + NX * l] * L.X[l];
                }
                L.gk[20 * (k & 1) + lane] = g;
            }
            COPRA_RF_PSTAMP(3);
        };
        auto stage_cls = [&](int k) -> int { return bcast_i32(stinfo, k) & 255; };
        auto stage_row0 = [&](int k) -> int { return bcast_i32(stinfo, k) >> 8; };

        // ---- sweep 1: backward factorisation.  Before: X holds z, Cb the rows' gradient coefficients, Y their weights D (with_rows).
        //      After: stage records in KF, P_0 in Pb, p_0 in pv0.  Returns false when a control block is not positive definite.
        auto sweep1 = [&](bool with_rows, bool with_x0_terms) -> bool {
            int q, b, r;
            lane_qbr(q, b, r);
            const bool v3 = q < 3 && r < 3;
            bool bad = false;
            double PX[3] = { 0.0, 0.0, 0.0 };
            double bA[3], bBu[3], aMx[3][3], aMu[2][3]; // B operands of T = P [A | B], A operands of M = H + [A B]' T
#pragma unroll
            for (int K = 0; K < 3; ++K) {
                const int row = 4 * K + q;
                bA[K] = *(b < 3 ? AB + row + NX * (4 * b + r) : L.Zs);
                bBu[K] = *((b < 2 && r < 3) ? AB + row + NX * (NX + 3 * b + r) : L.Zs);
#pragma unroll
                for (int I = 0; I < 3; ++I) aMx[I][K] = AB[row + NX * (4 * I + r)];
#pragma unroll
                for (int J = 0; J < 2; ++J) aMu[J][K] = *(r < 3 ? AB + row + NX * (NX + 3 * J + r) : L.Zs);
            }
            // offsets (doubles from the LDS base) of what this lane reads and writes
            int hx[5], hu[2];
            const int hstr = b < 3 ? kRfHStride : 20; // the accumulators start from an entry of H (copy of the stage's parity) | the gradient
#pragma unroll
            for (int I = 0; I < 5; ++I) {
                const int row = I < 3 ? 4 * I + q : (q < 3 ? NX + 3 * (I - 3) + q : -1);
                if (b < 3)
                    hx[I] = row >= 0 ? oHb + row + NZ * (4 * b + r) : oHb + NZ * NZ;
                else
                    hx[I] = (row >= 0 && r == 0) ? oGk + row : oGk + 18;
            }
#pragma unroll
            for (int J = 0; J < 2; ++J) hu[J] = (v3 && b < 2) ? oHb + (NX + 3 * J + q) + NZ * (NX + 3 * b + r) : oHb + NZ * NZ;
            const double m3 = b == 3 ? 1.0 : 0.0; // (the gradient column of T starts from p)
            const int wP = b < 3 ? oPb + q + NX * (4 * b + r) : oDU, rP = oPb + r + NX * q;
            const int wPend = b < 3 ? wP : (r == 0 ? oPv0 + q : oDU);
            const int wRb1 = q < 3 ? (b < 3 ? oRb + 20 * q + 4 * b + r : (r == 0 ? oRb + 20 * q + 18 : oDU)) : oDU;
            const int wRb2 = (v3 && b < 2) ? oRb + 20 * q + NX + 3 * b + r : oDU;
            const int wRa1 = q < 3 ? (b < 3 ? oRa + 20 * q + 4 * b + r : (r == 0 ? oRa + 20 * q + 18 : oDU)) : oDU;
            const int wRa2 = (v3 && b == 0) ? oRa + 20 * q + NX + r : oDU;
            const int rRb = q < 3 ? oRb + 20 * q + r : oZ0, rRb3 = v3 ? oRb + 20 * q + NX + r : oZ0;
            const int rRa = q < 3 ? oRa + 20 * q + r : oZ0;
            const int i1 = (r + 1) % 3, i2 = (r + 2) % 3, j1 = (q + 1) % 3, j2 = (q + 2) % 3;
            const int a1 = v3 ? 20 * i1 + j1 : 0, a2 = v3 ? 20 * i2 + j2 : 0, a3 = v3 ? 20 * i1 + j2 : 0, a4 = v3 ? 20 * i2 + j1 : 0;
            const int kKb = q < 3 ? (b < 3 ? oKb + 12 * q + 4 * b + r : (r == 0 ? oKvb + q : oSpare)) : oSpare;
            const int kKa = q < 3 ? (b < 3 ? oKa + 12 * q + 4 * b + r : (r == 0 ? oKva + q : oSpare)) : oSpare;
            const int kKba = (v3 && b == 0) ? oKba + 3 * q + r : oSpare;
            const int kNb = (v3 && b == 0) ? oNb + 3 * r + q : oSpare, kNa = (v3 && b == 0) ? oNa + 3 * r + q : oSpare;
            // one stage; `prep`: prepare stage k - 1 (class cn, rows from gn) in the shadow of the products
            auto stage = [&](int k, bool prep, int cn, int gn) {
                const int par = k & 1;
                double HX[5], HU[2];
                if (k == N) { // P_N = H_xx, p_N = g_x
                    wave_sync(); // (what prepare_stage(N) wrote is read below)
#pragma unroll
                    for (int I = 0; I < 3; ++I) PX[I] = lds[hx[I] + hstr * par];
                    if (prep) prepare_stage(k - 1, cn, gn, true, with_rows, with_x0_terms);
                    return;
                }
                // P as the left factor: through LDS, which replicates it over the hardware blocks; the stage's own H and gradient
                // (written one stage ago by prepare_stage) come back in the same round trip
#pragma unroll
                for (int I = 0; I < 3; ++I) lds[wP + 4 * I] = PX[I];
                wave_sync();
                if (k + 1 < N) rec_copy_out(k + 1); // (the record the stage before this one wrote: complete behind the sync)
                COPRA_RF_STAMP(0);
                double aP[3][3];
#pragma unroll
                for (int I = 0; I < 3; ++I)
#pragma unroll
                    for (int K = 0; K < 3; ++K) aP[I][K] = lds[rP + 4 * I + 4 * NX * K];
#pragma unroll
                for (int I = 0; I < 5; ++I) HX[I] = lds[hx[I] + hstr * par];
#pragma unroll
                for (int J = 0; J < 2; ++J) HU[J] = lds[hu[J] + kRfHStride * par];
                COPRA_RF_STAMP(1);
                // T = P [A B]  (+ p in the gradient column)
                double TX[3], TU[3];
#pragma unroll
                for (int I = 0; I < 3; ++I) TX[I] = PX[I] * m3, TU[I] = 0.0;
#pragma unroll
                for (int K = 0; K < 3; ++K)
#pragma unroll
                    for (int I = 0; I < 3; ++I) {
                        TX[I] = mfma_f64_4x4x4(aP[I][K], bA[K], TX[I]);
                        TU[I] = mfma_f64_4x4x4(aP[I][K], bBu[K], TU[I]);
                    }
                COPRA_RF_STAMP(2);
                // (the next stage's Hessian and gradient, in the shadow of these products)
                if (prep) prepare_stage(k - 1, cn, gn, true, with_rows, with_x0_terms);
                COPRA_RF_STAMP(3);
                // M = H + [A B]' T : the rows u_b FIRST -- they go to LDS (left factor of the first Schur update, the 3 x 3 block to
                // invert), and that round trip runs in the shadow of the other fifteen products
                double MX[5], MU[2];
#pragma unroll
                for (int I = 0; I < 5; ++I) MX[I] = HX[I];
                MU[0] = HU[0], MU[1] = HU[1];
#pragma unroll
                for (int K = 0; K < 3; ++K) {
                    MX[4] = mfma_f64_4x4x4(aMu[1][K], TX[K], MX[4]);
                    MU[1] = mfma_f64_4x4x4(aMu[1][K], TU[K], MU[1]);
                }
                lds[wRb1] = MX[4];
                lds[wRb2] = MU[1];
#pragma unroll
                for (int K = 0; K < 3; ++K) {
                    MX[3] = mfma_f64_4x4x4(aMu[0][K], TX[K], MX[3]);
                    MU[0] = mfma_f64_4x4x4(aMu[0][K], TU[K], MU[0]);
#pragma unroll
                    for (int I = 0; I < 3; ++I) MX[I] = mfma_f64_4x4x4(aMx[I][K], TX[K], MX[I]);
                }
                COPRA_RF_STAMP(4);
                // ---- eliminate u_b (controls 3 .. 5)
                wave_sync();
                double aRb[4];
#pragma unroll
                for (int I = 0; I < 3; ++I) aRb[I] = lds[rRb + 4 * I];
                aRb[3] = lds[rRb3];
                const double nB = neg_inv3(L.Rb + NX + 3, a1, a2, a3, a4, bad);
                const double KbX = mfma_f64_4x4x4(nB, MX[4], 0.0);
                const double KbU = mfma_f64_4x4x4(nB, MU[1], 0.0);
                // (the rows u_a first: they are the next ones to go through LDS)
                MX[3] = mfma_f64_4x4x4(aRb[3], KbX, MX[3]);
                MU[0] = mfma_f64_4x4x4(aRb[3], KbU, MU[0]);
                lds[wRa1] = MX[3];
                lds[wRa2] = MU[0];
#pragma unroll
                for (int I = 0; I < 3; ++I) MX[I] = mfma_f64_4x4x4(aRb[I], KbX, MX[I]);
                double* const Fk = rec_slot(k);
                Fk[kKb] = KbX;
                Fk[kKba] = KbU;
                Fk[kNb] = nB;
                COPRA_RF_STAMP(5);
                // ---- eliminate u_a (controls 0 .. 2)
                wave_sync();
                double aRa[3];
#pragma unroll
                for (int I = 0; I < 3; ++I) aRa[I] = lds[rRa + 4 * I];
                const double nA = neg_inv3(L.Ra + NX, a1, a2, a3, a4, bad);
                const double KaX = mfma_f64_4x4x4(nA, MX[3], 0.0);
#pragma unroll
                for (int I = 0; I < 3; ++I) PX[I] = mfma_f64_4x4x4(aRa[I], KaX, MX[I]);
                Fk[kKa] = KaX;
                Fk[kNa] = nA;
                COPRA_RF_STAMP(6);
            };
            qnext = lane < NZ ? S.f_q[N * NZ + lane] : 0.0;
            load_class(stage_cls(N), N);
            prepare_stage(N, stage_cls(N), stage_row0(N), true, with_rows, with_x0_terms);
            int k = N;
            while (k >= 0) { // runs of stages whose NEXT stage (the one being prepared) is of one class: its tables are loop constants
                const int cn = stage_cls(k > 0 ? k - 1 : 0);
                load_class(cn, k > 0 ? k - 1 : 0);
                do {
                    stage(k, k > 0, cn, k > 0 ? stage_row0(k - 1) : 0);
                    --k;
                } while (k >= 0 && (k == 0 || stage_cls(k - 1) == cn));
            }
            // P_0, p_0 for the step in x_0
            wave_sync();
            if (N > 0) rec_copy_out(0);
#pragma unroll
            for (int I = 0; I < 3; ++I) lds[wPend + 4 * I] = PX[I];
            wave_sync();
            return !bad;
        };
        // ---- sweep 3: backward VECTOR sweep through the stored factors.  Before: X holds z, Cb the rows' gradient coefficients.
        //      After: kv in the stage records, p_0 in pv0.
        auto sweep3 = [&]() {
            int q, b, r;
            lane_qbr(q, b, r);
            const bool v3 = q < 3 && r < 3;
            double pB[3] = { 0.0, 0.0, 0.0 };
            double hM[3], hMb[3]; // A operands of h = g + [A B]' p: row block = hardware block (x | x | x | u_a), and u_b
#pragma unroll
            for (int K = 0; K < 3; ++K) {
                const int row = 4 * K + q;
                hM[K] = *(b < 3 ? AB + row + NX * (4 * b + r) : (r < 3 ? AB + row + NX * (NX + r) : L.Zs));
                hMb[K] = *(r < 3 ? AB + row + NX * (NX + 3 + r) : L.Zs);
            }
            // operands of a stage from its record: -M_bb^-1, -M'_aa^-1 (element (r, q)), K_b' | K_ba' and K_a' (row block = hardware
            // block); absent ones read the record's zero slot
            const int oN1 = v3 ? oNb + 3 * r + q : oZero, oN2 = v3 ? oNa + 3 * r + q : oZero;
            const int oB = q < 3 ? (b < 3 ? oKb + 12 * q + 4 * b + r : (r < 3 ? oKba + 3 * q + r : oZero)) : oZero;
            const int oA = (q < 3 && b < 3) ? oKa + 12 * q + 4 * b + r : oZero;
            const int gh = r == 0 ? (b < 3 ? 4 * b + q : (q < 3 ? NX + q : 18)) : 18, ghb = (r == 0 && q < 3) ? NX + 3 + q : 18;
            const int wkv = (r == 0 && q < 3 && b < 2) ? (b == 0 ? oKva + q : oKvb + q) : oSpare;
            const double sel = b == 0 ? 1.0 : 0.0;
            double nBop = 0.0, nAop = 0.0, aKbT = 0.0, aKaT = 0.0;
            auto fetch = [&](int k) {
                const double* Fk = rec_slot(k);
                nBop = Fk[oN1], nAop = Fk[oN2], aKbT = Fk[oB], aKaT = Fk[oA];
            };
            // the records come back from the workspace: N - 1 and N - 2 now, k - 3 while stage k runs (dropped into its slot at stage k - 1,
            // fetched from there at stage k - 2)
            double rq0, rq1;
            {
                wave_sync_full(); // (this wave's copies of the records -- the factorisation's, the last vector sweep's kv -- are behind it)
                double a0, a1, b0, b1;
                rec_request(N - 1, a0, a1);
                rec_request(N - 2, b0, b1);
                rec_request(N - 3, rq0, rq1);
                rec_put(N - 1, a0, a1);
                rec_put(N - 2, b0, b1);
            }
            auto stage = [&](int k, bool prep, int cn, int gn) {
                wave_sync(); // (the gradient prepare_stage(k) wrote)
                if (k < N) { // record k - 2 into its slot (requested one stage ago), k - 3 on its way
                    rec_put(k - 2, rq0, rq1);
                    rec_request(k - 3, rq0, rq1);
                }
                const double* gk = L.gk + 20 * (k & 1);
                if (k == N) {
#pragma unroll
                    for (int K = 0; K < 3; ++K) pB[K] = gk[4 * K + q];
                    if (prep) fetch(k - 1), prepare_stage(k - 1, cn, gn, false, true, x0_free);
                    return;
                }
                // h = g + [A B]' p : hardware block b = row block (x_0..3 | x_4..7 | x_8..11 | u_a), u_b on its own (replicated).  The
                // products start from zero and g is added afterwards: its way back from LDS runs under them
                const double gv = gk[gh], gbv = gk[ghb];
                const double c_nB = nBop, c_nA = nAop, c_KbT = aKbT, c_KaT = aKaT;
                double hv = 0.0, hbv = 0.0;
#pragma unroll
                for (int K = 0; K < 3; ++K) {
                    hv = mfma_f64_4x4x4(hM[K], pB[K], hv);
                    hbv = mfma_f64_4x4x4(hMb[K], pB[K], hbv);
                }
                if (prep) fetch(k - 1), prepare_stage(k - 1, cn, gn, false, true, x0_free); // (in the shadow of the products)
                hv += gv, hbv += gbv;
                const double kvb = mfma_f64_4x4x4(c_nB, hbv, 0.0); // kv_b = -M_bb^-1 h_b
                const double hp = mfma_f64_4x4x4(c_KbT, hbv, hv); // h' = h + K_b' h_b  (x and u_a)
                const double ha = row_bcast_f64<12>(hp); // h'_a to every hardware block
                const double kva = mfma_f64_4x4x4(c_nA, ha, 0.0); // kv_a = -M'_aa^-1 h'_a
                const double pn = mfma_f64_4x4x4(c_KaT, ha, hp); // p = h'_x + K_a' h'_a
                rec[(size_t)k * KS + wkv] = sel * kva + (1.0 - sel) * kvb; // (straight to the workspace: the forward sweep reads it from there)
                pB[0] = row_bcast_f64<0>(pn), pB[1] = row_bcast_f64<4>(pn), pB[2] = row_bcast_f64<8>(pn);
            };
            qnext = lane < NZ ? S.f_q[N * NZ + lane] : 0.0;
            load_class(stage_cls(N), N);
            prepare_stage(N, stage_cls(N), stage_row0(N), false, true, x0_free);
            int k = N;
            while (k >= 0) {
                const int cn = stage_cls(k > 0 ? k - 1 : 0);
                load_class(cn, k > 0 ? k - 1 : 0);
                do {
                    stage(k, k > 0, cn, k > 0 ? stage_row0(k - 1) : 0);
                    --k;
                } while (k >= 0 && (k == 0 || stage_cls(k - 1) == cn));
            }
            wave_sync();
            if (r == 0 && b < 3) L.pv0[4 * b + q] = b == 0 ? pB[0] : b == 1 ? pB[1] : pB[2];
            wave_sync();
        };
        // ---- forward sweep: dz_k into Y, from dx_0 in dx0[]; the controls through the stored gains, the states through [A B]
        auto forward = [&]() {
            int q, b, r;
            lane_qbr(q, b, r);
            const bool v3 = q < 3 && r < 3;
            double xB[3], fA[5]; // A operands of x+ = [A B] (x, u_a, u_b): row block = hardware block
#pragma unroll
            for (int K = 0; K < 3; ++K) fA[K] = *(b < 3 ? AB + (4 * b + r) + NX * (4 * K + q) : L.Zs);
            fA[3] = *((b < 3 && q < 3) ? AB + (4 * b + r) + NX * (NX + q) : L.Zs);
            fA[4] = *((b < 3 && q < 3) ? AB + (4 * b + r) + NX * (NX + 3 + q) : L.Zs);
#pragma unroll
            for (int K = 0; K < 3; ++K) xB[K] = L.dx0[4 * K + q];
            if (r == 0 && b < 3) L.Y[4 * b + q] = b == 0 ? xB[0] : b == 1 ? xB[1] : xB[2];
            // operands of a stage from its record (fetched one stage ahead): rows of K_a, K_b (A operand: row r, column 4 K + q), K_ba, kv
            int oRa_[3], oRb_[3];
#pragma unroll
            for (int K = 0; K < 3; ++K) oRa_[K] = r < 3 ? oKa + 12 * r + 4 * K + q : oZero, oRb_[K] = r < 3 ? oKb + 12 * r + 4 * K + q : oZero;
            const int oBa = v3 ? oKba + 3 * r + q : oZero;
            const int oVa = (r == 0 && q < 3) ? oKva + q : oZero, oVb = (r == 0 && q < 3) ? oKvb + q : oZero;
            const int wu = (r == 0 && q < 3 && b < 2) ? oY + NX + 3 * b + q : oDU, wus = (r == 0 && q < 3 && b < 2) ? NZ : 0;
            const int wx = (r == 0 && b < 3) ? oY + NZ + 4 * b + q : oDU, wxs = (r == 0 && b < 3) ? NZ : 0;
            const double sel = b == 0 ? 1.0 : 0.0;
            double nKa[3], nKb[3], nKba, nva, nvb;
            auto fetch = [&](int k) {
                const double* Fk = rec_slot(k);
#pragma unroll
                for (int K = 0; K < 3; ++K) nKa[K] = Fk[oRa_[K]], nKb[K] = Fk[oRb_[K]];
                nKba = Fk[oBa], nva = Fk[oVa], nvb = Fk[oVb];
            };
            // the records come back from the workspace: 0 and 1 now, k + 3 while stage k runs (dropped into its slot at stage k + 1)
            double rq0, rq1;
            {
                wave_sync_full(); // (the vector sweep's kv)
                double a0, a1, b0, b1;
                rec_request(0, a0, a1);
                rec_request(1, b0, b1);
                rec_request(2, rq0, rq1);
                rec_put(0, a0, a1);
                rec_put(1, b0, b1);
                wave_sync();
            }
            if (N > 0) fetch(0);
            for (int k = 0; k < N; ++k) {
                rec_put(k + 2, rq0, rq1);
                rec_request(k + 3, rq0, rq1);
                wave_sync(); // (record k + 1, put one stage ago, is fetched below: LDS runs in issue order, the compiler must not reorder)
                double fKa[3], fKb[3];
#pragma unroll
                for (int K = 0; K < 3; ++K) fKa[K] = nKa[K], fKb[K] = nKb[K];
                const double fKba = nKba;
                double ua = nva, ub = nvb;
                double xn = 0.0;
#pragma unroll
                for (int K = 0; K < 3; ++K) {
                    ua = mfma_f64_4x4x4(fKa[K], xB[K], ua);
                    ub = mfma_f64_4x4x4(fKb[K], xB[K], ub);
                    xn = mfma_f64_4x4x4(fA[K], xB[K], xn);
                }
                if (k + 1 < N) fetch(k + 1); // (the next stage's operands: requested in the shadow of the products)
                ub = mfma_f64_4x4x4(fKba, ua, ub);
                xn = mfma_f64_4x4x4(fA[3], ua, xn);
                xn = mfma_f64_4x4x4(fA[4], ub, xn);
                lds[wu + wus * k] = sel * ua + (1.0 - sel) * ub;
                lds[wx + wxs * k] = xn;
                xB[0] = row_bcast_f64<0>(xn), xB[1] = row_bcast_f64<4>(xn), xB[2] = row_bcast_f64<8>(xn);
            }
            if (lane < kRfNU) L.Y[N * NZ + NX + lane] = 0.0;
            wave_sync();
        };
        // dx_0 = -(P_0 + R - P0)^-1 p_0  (InitialStateLMPC) into dx0[]; Gauss-Jordan on [P | -p] (nx x (nx+1)); fixed x_0: zero
        auto solve_x0 = [&]() -> bool {
            if (!x0_free) {
                if (lane < NX) L.dx0[lane] = 0.0;
                wave_sync();
                return true;
            }
            // Gaussian elimination without pivoting (the matrix is symmetric positive definite) on the padded 12 x 13 system [P | -p]:
            // entry (r, cc) = e / 13, e % 13 of lane + 64 i; ONE pass per pivot -- it reads column p and row p, and writes the columns
            // beyond p of the other rows --, the solution is rhs / diagonal at the end
            constexpr int W1 = NX + 1;
            double* GJ = L.GJ;
            int er[3], ec[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int e = lane + 64 * i;
                er[i] = e / W1, ec[i] = e - er[i] * W1;
                if (e < NX * W1) {
                    const int r = er[i], cc = ec[i];
                    double v;
                    if (cc < NX)
                        v = (r < nx && cc < nx) ? L.Pb[r + NX * cc] + L.H0[r + NX * cc] : (r == cc ? 1.0 : 0.0);
                    else
                        v = r < nx ? -L.pv0[r] : 0.0;
                    GJ[e] = v;
                }
            }
            wave_sync();
            bool ok = true;
#pragma unroll
            for (int p = 0; p < NX; ++p) {
                const double piv = GJ[p * W1 + p];
                ok = ok && (piv > 0.0);
                const double ip = rf_rcp(piv);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int e = lane + 64 * i;
                    if (e < NX * W1 && er[i] != p && ec[i] > p) GJ[e] -= (GJ[er[i] * W1 + p] * ip) * GJ[p * W1 + ec[i]];
                }
                wave_sync();
            }
            if (lane < NX) L.dx0[lane] = GJ[lane * W1 + NX] * rf_rcp(GJ[lane * W1 + lane]);
            wave_sync();
            return ok;
        };

        // ------------------------------------------------------------------ 1. starting point
        bool good = true;
        for (int e = lane; e < 64 * kRfZR; e += kWave) L.X[e] = 0.0;
        for (int e = lane; e < 64 * MR; e += kWave) L.Y[e] = 0.0, L.Cb[e] = 0.0;
        for (int e = lane; e < NX * NX; e += kWave) L.H0[e] = 0.0;
        if (lane < 16) L.G0[lane] = 0.0;
        wave_sync();
        if (x0_free) {
            // P0 (unconstrained cost-to-go Hessian) and g0 = dJ/dx0 at (x0, U) = 0
            rollout(); // x0 = 0, U = 0
            good = sweep1(false, false) && good;
            // adjoint sweep for g0: lam_N = g_N,x ; lam_k = g_k,x + A' lam_{k+1}   (g = W z + q)
            qnext = lane < NZ ? S.f_q[N * NZ + lane] : 0.0;
            for (int k = N; k >= 0; --k) {
                load_class(stage_cls(k), k); // (one-off sweep: the class tables are simply reloaded per stage)
                prepare_stage(k, stage_cls(k), stage_row0(k), false, false, false);
                wave_sync();
                double acc = 0.0;
                if (lane < NX) {
                    acc = L.gk[20 * (k & 1) + lane];
                    if (k < N) {
#pragma unroll
                        for (int l = 0; l < NX; ++l) acc += AB[l + NX * lane] * L.dx0[l];
                    }
                }
                wave_sync();
                if (lane < NX) L.dx0[lane] = acc;
                wave_sync();
            }
            for (int e = lane; e < NX * NX; e += kWave) {
                const int j = e / NX, i = e - j * NX;
                L.H0[e] = (i < nx && j < nx) ? P.is_R[i + nx * j] - L.Pb[e] : 0.0;
            }
            if (lane < NX) L.G0[lane] = lane < nx ? P.is_r[lane] - L.dx0[lane] : 0.0;
            wave_sync();
            for (int e = lane; e < 64 * kRfZR; e += kWave) L.X[e] = 0.0;
            wave_sync();
        }
        if (lane < NX) {
            double v = 0.0;
            if (lane < nx) {
                v = P.x0[(size_t)inst * nx + lane];
                if (x0_free) v = fmin(fmax(v, P.x0lb[(size_t)inst * nx + lane]), P.x0ub[(size_t)inst * nx + lane]);
            }
            L.X[lane] = v;
        }
        rollout();
#pragma unroll
        for (int j = 0; j < MR; ++j) {
            const int gi = 64 * j + lane;
            if (gi < m && (rinf[j] >> 28) == kRfIneq) Sv[j] = fmax(Fr[j] - row_dot(rinf[j], L.X), S.s_floor);
        }
        { // multipliers: lam0 > 0: that value; lam0 < 0 (the default): every complementarity product s lam starts at |lam0| x the MEAN
          // slack -- a centred start at the problem's own scale (bounds of 200 and bounds of 0.5 both start with lam ~ 1 on a typical row)
            double ssum = 0.0;
#pragma unroll
            for (int j = 0; j < MR; ++j)
                if (64 * j + lane < m && (rinf[j] >> 28) == kRfIneq) ssum += Sv[j];
            const double mu0 = S.lam0 > 0.0 ? 0.0 : -S.lam0 * wave_sum(ssum) * inv_mi;
#pragma unroll
            for (int j = 0; j < MR; ++j)
                if (64 * j + lane < m && (rinf[j] >> 28) == kRfIneq) Lam[j] = S.lam0 > 0.0 ? S.lam0 : mu0 / Sv[j];
        }
        stamp_outer(0);
        // ------------------------------------------------------------------ 2. Newton iterations
        // Per row only the slack, the multiplier, the right-hand side and the predictor's ds * dl live in registers; the residual
        // rp = a' z + s - f  is formed again from the iterate (which never leaves LDS) wherever it is needed: three LDS reads
        // instead of two registers per row.
        auto residual = [&](int j, int fl) -> double { // of an inequality row (with its slack) or an equality row
            const double az = row_dot(rinf[j], L.X);
            return fl == kRfEq ? az - Fr[j] : az + Sv[j] - Fr[j];
        };
        // crossover (lmpc_riccati.hpp): rows with lam > s become regularised equality rows, the others are switched off
        auto cross_over = [&]() {
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                const int gi = 64 * j + lane;
                if (gi < m && (rinf[j] >> 28) == kRfIneq) {
                    const bool active = Lam[j] > Sv[j];
                    rinf[j] = (rinf[j] & 0x0FFFFFFF) | ((active ? kRfEq : kRfOff) << 28);
                    Sv[j] = active ? kRicWasActive : kRicWasIdle;
                    Lam[j] = active ? Lam[j] : 0.0;
                }
            }
        };
        int it = 0;
        double prev_step = 1.0e300; // the step before
        bool tail_ok = false; // ric_tail_ok of the iterate the loop stands on (lmpc_riccati.hpp)
        bool polishing = false;
        int polish_it = 0, refinements = 0;
        bool converged = false;
        for (it = 1; it <= S.max_iter && good; ++it) {
#if defined(__HIP_DEVICE_COMPILE__)
            // (the row descriptors never change, so the compiler would decode all fifteen of them -- stage offset, template, two
            //  components, flag -- ONCE, ahead of this loop, and keep ~ 90 registers live across it: make them opaque per iteration)
#pragma unroll
            for (int j = 0; j < MR; ++j) asm volatile("" : "+v"(rinf[j]));
#endif
            // ---- bulk phase: residuals, barrier weights D (-> Y) and gradient coefficients c (-> Cb) of every row
            double musum = 0.0, maxr = 0.0;
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                if (j % 5 == 0) sched_fence(); // (bounds how many rows' operands the scheduler keeps in flight: registers)
                const int gi = 64 * j + lane, fl = rinf[j] >> 28;
                double Dv = 0.0, Cv = 0.0;
                if (gi < m && fl != kRfOff) {
                    const double rp = residual(j, fl);
                    if (fl == kRfEq) {
                        Dv = 1.0 / delta;
                        Cv = Lam[j] + rp / delta;
                    } else {
                        Dv = Lam[j] * rf_rcp(Sv[j]);
                        Cv = Dv * rp;
                        maxr = fmax(maxr, fabs(rp));
                        musum += Sv[j] * Lam[j];
                    }
                }
                if (gi < m) L.Y[gi] = Dv, L.Cb[gi] = Cv;
            }
            wave_sync();
            const double mu = wave_sum(musum) * inv_mi;
            const double maxres = wave_max(maxr);
            stamp_outer(1);
            // ---- sweep 1 (backward): factorisation and the predictor's right-hand side
            good = sweep1(true, x0_free) && good;
            stamp_outer(2);
            if (!good) { // (the factorisation broke down; the iterate itself is untouched: ric_tail_ok)
                if (!polishing && n_ineq > 0 && maxres <= 1e-9 && mu <= kRicEarlySwitchMu && it > 1) { // ... under large barrier weights: cross over here
                    cross_over();
                    polishing = true;
                    good = true;
                    prev_step = 1.0e300;
                    tail_ok = false;
                    continue;
                }
                converged = tail_ok;
                break;
            }
            // ---- predictor: forward sweep, then the rows in bulk
            good = solve_x0() && good;
            stamp_outer(3);
            forward();
            stamp_outer(4);
            double amin = 1.0e300;
            // mu_aff = sum (s + a ds)(lam + a dl) / n  needs the step length a of the whole wave first: its three coefficients in a
            // are summed in the same pass (no second pass over the directions, no registers to keep them in)
            double q0 = 0.0, q1 = 0.0, q2 = 0.0;
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                if (j % 5 == 0) sched_fence();
                const int gi = 64 * j + lane, fl = rinf[j] >> 28;
                double dsdl = 0.0;
                if (gi < m && fl == kRfIneq) {
                    const double ds = -residual(j, fl) - row_dot(rinf[j], L.Y);
                    const double dl = (-Lam[j] * Sv[j] - Lam[j] * ds) * rf_rcp(Sv[j]);
                    if (ds < 0.0) amin = fmin(amin, -Sv[j] * rf_rcp(ds));
                    if (dl < 0.0) amin = fmin(amin, -Lam[j] * rf_rcp(dl));
                    dsdl = ds * dl;
                    q0 += Sv[j] * Lam[j], q1 += Sv[j] * dl + Lam[j] * ds, q2 += dsdl;
                }
                Dd[j] = dsdl; // (kept for the final direction)
            }
            amin = -wave_max(-amin);
            double sigma_mu = 0.0;
            {
                const double aaff = fmin(1.0, amin);
                const double mu_aff = wave_sum(q0 + aaff * (q1 + aaff * q2)) * inv_mi;
                const double ratio = mu > 0.0 ? mu_aff / mu : 0.0;
                sigma_mu = ratio * ratio * ratio * mu;
            }
            // ---- corrector right-hand side: gradient coefficients of the rows
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                if (j % 5 == 0) sched_fence();
                const int gi = 64 * j + lane, fl = rinf[j] >> 28;
                double Cv = 0.0;
                if (gi < m && fl != kRfOff) {
                    const double rp = residual(j, fl);
                    if (fl == kRfEq)
                        Cv = Lam[j] + rp / delta; // as in the predictor
                    else
                        Cv = ((sigma_mu - Dd[j]) + Lam[j] * rp) * rf_rcp(Sv[j]);
                }
                if (gi < m) L.Cb[gi] = Cv;
            }
            wave_sync();
            stamp_outer(6);
            sweep3();
            stamp_outer(5);
            good = solve_x0() && good;
            stamp_outer(3);
            forward();
            stamp_outer(4);
            // ---- final direction of the rows: the step length first, then (the direction formed once more) the update
            amin = 1.0e300;
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                if (j % 5 == 0) sched_fence();
                const int gi = 64 * j + lane, fl = rinf[j] >> 28;
                if (gi < m && fl == kRfIneq) {
                    const double ds = -residual(j, fl) - row_dot(rinf[j], L.Y);
                    const double dl = ((sigma_mu - Dd[j]) - Lam[j] * Sv[j] - Lam[j] * ds) * rf_rcp(Sv[j]);
                    if (ds < 0.0) amin = fmin(amin, -Sv[j] * rf_rcp(ds));
                    if (dl < 0.0) amin = fmin(amin, -Lam[j] * rf_rcp(dl));
                }
            }
            amin = -wave_max(-amin);
            const double tau = mu > 1e-10 ? 0.995 : 0.9999;
            const double alpha = amin < 1.0 ? fmin(1.0, tau * amin) : 1.0;
            double musum2 = 0.0, maxe = 0.0;
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                if (j % 5 == 0) sched_fence();
                const int gi = 64 * j + lane, fl = rinf[j] >> 28;
                if (gi < m && fl != kRfOff) {
                    const double rp = residual(j, fl), adz = row_dot(rinf[j], L.Y);
                    if (fl == kRfIneq) {
                        const double ds = -rp - adz;
                        const double dl = ((sigma_mu - Dd[j]) - Lam[j] * Sv[j] - Lam[j] * ds) * rf_rcp(Sv[j]);
                        Sv[j] += alpha * ds;
                        Lam[j] += alpha * dl;
                        musum2 += Sv[j] * Lam[j];
                    } else {
                        const double re = rp + alpha * adz; // residual of the row at the new point
                        // Newton direction of the multiplier of the regularised row (a'dz - delta dnu = -(a'z - f)): dnu = (rp + a'dz) / delta,
                        // damped like every other unknown.  (Rounds 2-3 added re / delta -- the same after a full step; after a damped one it
                        // differs by (1 - alpha) rp / delta, 1e9 x a residual that is not small yet: the multiplier was thrown to +-1e7 and took
                        // the rest of the iterations to come back, lmpc_riccati.hpp.)
                        Lam[j] += alpha * (rp + adz) / delta;
                        maxe = fmax(maxe, fabs(re));
                    }
                }
            }
            wave_sync(); // (every row has read the old iterate)
            double z_inf = 0.0, step_inf = 0.0;
            for (int e = lane; e < NE; e += kWave) {
                const double dz = L.Y[e];
                const double zn = L.X[e] + alpha * dz;
                L.X[e] = zn;
                step_inf = fmax(step_inf, fabs(dz));
                z_inf = fmax(z_inf, fabs(zn));
            }
            step_inf = wave_max(step_inf); // (the FULL Newton step: lmpc_riccati.hpp)
            z_inf = wave_max(z_inf);
            wave_sync();
            const double mu_new = wave_sum(musum2) * inv_mi;
            const double res_new = fmax((1.0 - alpha) * maxres, wave_max(maxe));
            if (!(mu_new == mu_new) || !(step_inf == step_inf)) {
                good = false;
                break;
            }
            stamp_outer(6);
#if !defined(__HIP_DEVICE_COMPILE__) && defined(COPRA_EMU_TRACE)
            if (lane == 0) fprintf(stderr, "it %2d alpha %.4f mu %.3e -> %.3e res %.3e (maxres %.3e eq %.3e) step %.3e z %.3e\n", it, alpha, mu, mu_new, res_new, maxres, maxe, step_inf, z_inf);
#endif
            if (!polishing && n_ineq > 0 && res_new <= 1e-9 && mu_new <= kRicSwitchMu) { // crossover: lmpc_riccati.hpp
                cross_over();
                polishing = true;
                prev_step = 1.0e300;
                tail_ok = false;
                continue;
            }
            polish_it += polishing ? 1 : 0;
            bool conv = ric_converged(S, res_new, mu_new, step_inf, prev_step, z_inf) && (!polishing || polish_it >= 2);
            tail_ok = !polishing && ric_tail_ok(res_new, mu_new, step_inf, prev_step, z_inf);
            if (conv && polishing) { // was the active set the right one?  A held row that pulls is released, a row left out that is violated is taken
                double flips = 0.0;
#pragma unroll
                for (int j = 0; j < MR; ++j) {
                    const int gi = 64 * j + lane, fl = rinf[j] >> 28;
                    if (gi < m && fl == kRfEq && Sv[j] == kRicWasActive) {
                        const double lv = Lam[j] + (row_dot(rinf[j], L.X) - Fr[j]) / delta;
                        if (!(lv >= -1e-9 * (1.0 + fabs(Lam[j])))) {
                            rinf[j] = (rinf[j] & 0x0FFFFFFF) | (kRfOff << 28);
                            Sv[j] = kRicWasIdle;
                            Lam[j] = 0.0;
                            flips += 1.0;
                        }
                    } else if (gi < m && fl == kRfOff && Sv[j] == kRicWasIdle && !(row_dot(rinf[j], L.X) - Fr[j] <= 1e-9 * (1.0 + fabs(Fr[j])))) {
                        rinf[j] = (rinf[j] & 0x0FFFFFFF) | (kRfEq << 28);
                        Sv[j] = kRicWasActive;
                        Lam[j] = 0.0;
                        flips += 1.0;
                    }
                }
                if (wave_sum(flips) > 0.0) { // another round on the corrected set (at most kRicRefinements of them)
                    refinements += 1;
                    if (refinements > kRicRefinements) {
                        good = false;
                        break;
                    }
                    polish_it = 0;
                    prev_step = 1.0e300;
                    conv = false;
                }
            }
            prev_step = step_inf;
            if (conv) {
                converged = true;
                break;
            }
        }

        // ------------------------------------------------------------------ 3. results (LMPC.cpp:282-286)
        if (converged) {
            rollout(); // trajectory = Phi x0 + Psi U + xi, recomputed from the final x0 and U
            for (int e = lane; e < N * nu; e += kWave) {
                const int k = e / nu, i = e - k * nu;
                P.control[(size_t)inst * P.n + e] = L.X[k * NZ + NX + i];
            }
            for (int e = lane; e < P.X; e += kWave) {
                const int k = e / nx, i = e - k * nx;
                P.trajectory[(size_t)inst * P.X + e] = L.X[k * NZ + i];
            }
            if (P.initial_state && P.x0_opt)
                for (int e = lane; e < nx; e += kWave) P.x0_opt[(size_t)inst * nx + e] = L.X[e];
            if (lane == 0) {
                P.status[inst] = 0;
                P.iter[2 * (size_t)inst] = it;
                P.iter[2 * (size_t)inst + 1] = 0;
            }
        } else {
            // not converged (infeasible / degenerate): the condensed Goldfarb-Idnani kernel decides its status
            if (lane == 0) {
                P.status[inst] = 3;
                P.iter[2 * (size_t)inst] = it;
                P.iter[2 * (size_t)inst + 1] = 0;
                if (P.ovf_count) P.ovf_list[atomic_append(P.ovf_count)] = inst;
            }
        }
        if (P.prof && lane == 0) {
            stamp_outer(6);
            prof[7] = cycle_counter() - tstart;
            for (int q = 0; q < 8; ++q) P.prof[8 * (size_t)inst + q] = prof[q];
        }
        wave_sync();
    }
#undef COPRA_RF_STAMP
#undef COPRA_RF_PSTAMP
#undef stamp_outer
}

} // namespace copra_hip
