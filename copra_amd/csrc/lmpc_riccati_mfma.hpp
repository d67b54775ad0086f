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

namespace copra_hip {

struct RfLds {
    double *X, *Y, *Cb, *KF, *KV, *Hb, *Pb, *Rb, *Ra, *H0, *G0, *AB, *dv, *GJ, *pv0, *dx0, *Zs, *gk, *TT;
};

COPRA_DEV RfLds carve_rf(double* lds, int N, int ntmpl)
{
    RfLds L;
    double* p = lds;
    L.X = p, p += 64 * kRfZR; // the iterate z = (x_k, u_k)_k, padded
    L.Y = p, p += 64 * kRfMR; // row weights D (backward factorisation) | the step dz (forward sweeps)
    L.Cb = p, p += 64 * kRfMR; // gradient coefficients c of the rows (g_k += A_k' c)
    L.KF = p, p += N * kRfKStride;
    L.KV = p, p += N * 8;
    L.Hb = p, p += kRfNZ * kRfNZ;
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
    L.Zs = p, p += 4;
    L.gk = p, p += 20; // gradient of the stage in flight
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

COPRA_DEV void lmpc_riccati_mfma_body(const FusedPlan& P, const StagePlan& S)
{
    constexpr int NX = kRfNX, NZ = kRfNZ, MR = kRfMR, KS = kRfKStride;
    constexpr int oKa = 0, oKb = 36, oKba = 72, oNb = 81, oNa = 90;
    const int lane = lane_id();
    const int nx = S.nx, nu = S.nu, N = S.N, m = S.m;
    const int NE = (N + 1) * NZ; // entries of the padded stage vectors
    const RfLds L = carve_rf(lds_base(), N, S.fast_ntmpl);
    const int q4 = lane >> 4, hb = (lane >> 2) & 3, r4 = lane & 3; // lane = 16 q + 4 b + r
    const double delta = S.delta;
    const double BIGF = 1e299;
    const bool qr3 = q4 < 3 && r4 < 3;

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
        if (lane < 4) L.Zs[lane] = 0.0;
        for (int e = lane; e < 2 * S.fast_ntmpl; e += kWave) L.TT[e] = S.f_rval[e];
        wave_sync();
        const double* const AB = L.AB;
        // operands of the matrix products that are constants of the instance (element of [A B] each lane hands in)
        double bA[3], bBu[3], aMx[3][3], aMu[2][3], fA[5], hM[3], hMb[3];
#pragma unroll
        for (int K = 0; K < 3; ++K) {
            const int row = 4 * K + q4;
            bA[K] = hb < 3 ? AB[row + NX * (4 * hb + r4)] : 0.0;
            bBu[K] = (hb < 2 && r4 < 3) ? AB[row + NX * (NX + 3 * hb + r4)] : 0.0;
#pragma unroll
            for (int I = 0; I < 3; ++I) aMx[I][K] = AB[row + NX * (4 * I + r4)];
#pragma unroll
            for (int J = 0; J < 2; ++J) aMu[J][K] = r4 < 3 ? AB[row + NX * (NX + 3 * J + r4)] : 0.0;
            fA[K] = hb < 3 ? AB[(4 * hb + r4) + NX * row] : 0.0;
            hM[K] = hb < 3 ? AB[row + NX * (4 * hb + r4)] : (r4 < 3 ? AB[row + NX * (NX + r4)] : 0.0);
            hMb[K] = r4 < 3 ? AB[row + NX * (NX + 3 + r4)] : 0.0;
        }
        fA[3] = (hb < 3 && q4 < 3) ? AB[(4 * hb + r4) + NX * (NX + q4)] : 0.0;
        fA[4] = (hb < 3 && q4 < 3) ? AB[(4 * hb + r4) + NX * (NX + 3 + q4)] : 0.0;

        // ------------------------------------------------------------------ per-row state (registers: row 64 j + lane in element j)
        int rinf[MR]; // stage | template << 8 | flag << 28
        int roff[MR]; // entries of the padded stage vectors the row's two coefficients multiply: o0 | o1 << 16
        double Fr[MR], Sv[MR], Lam[MR], Rp[MR], Dd[MR];
        // right-hand sides (Cb serves as the exchange buffer for the flags below)
#pragma unroll
        for (int j = 0; j < MR; ++j) {
            const int gi = 64 * j + lane;
            int info = 0, off = 0;
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
                off = (k * NZ + (c0 < 0 ? 0 : c0)) | ((k * NZ + (c1 < 0 ? 0 : c1)) << 16);
            }
            rinf[j] = info, roff[j] = off;
            Fr[j] = f, Sv[j] = 1.0, Lam[j] = 0.0, Rp[j] = 0.0, Dd[j] = 0.0;
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
        auto row_dot = [&](int info, int off, const double* v) -> double {
            const int t = (info >> 8) & 0xFFFFF;
            return L.TT[2 * t] * v[off & 0xFFFF] + L.TT[2 * t + 1] * v[(off >> 16) & 0xFFFF]; // (absent coefficients carry value 0)
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
        // this lane's element (r, q) of -inverse of the symmetric 3 x 3 block m(a, b) = base[20 a + b]: adjugate over determinant,
        // ONE reciprocal (every lane computes the determinant from six wave-uniform reads and ITS cofactor from four reads at its
        // own addresses); lanes outside the block return 0.  Positive definite <=> m22, C00, det > 0 (Sylvester).
        const int ai1 = (r4 + 1) % 3, ai2 = (r4 + 2) % 3, aj1 = (q4 + 1) % 3, aj2 = (q4 + 2) % 3;
        const int ad1 = qr3 ? 20 * ai1 + aj1 : 0, ad2 = qr3 ? 20 * ai2 + aj2 : 0, ad3 = qr3 ? 20 * ai1 + aj2 : 0, ad4 = qr3 ? 20 * ai2 + aj1 : 0;
        auto neg_inv3 = [&](const double* base, bool& bad) -> double {
            const double m00 = base[0], m01 = base[1], m02 = base[2], m11 = base[21], m12 = base[22], m22 = base[42];
            const double x1 = base[ad1], x2 = base[ad2], x3 = base[ad3], x4 = base[ad4];
            const double c00 = m11 * m22 - m12 * m12, c01 = m12 * m02 - m01 * m22, c02 = m01 * m12 - m11 * m02;
            const double det = m00 * c00 + (m01 * c01 + m02 * c02);
            bad = bad || !(m22 > 0.0) || !(c00 > 0.0) || !(det > 0.0);
            const double v = (x1 * x2 - x3 * x4) * (-rf_rcp(det));
            return qr3 ? v : 0.0;
        };

        // ---- tables of the stage class in flight (registers): the entries of H = W + A' D A the rows touch (one per lane), and for
        //      the lanes 0 .. 17 the non-zeros of row `lane` of W and the rows that touch component `lane`
        int cur_cls = -1;
        int t_ent = 0, t_r0 = 0, t_r1 = 0, t_r2 = 0, t_r3 = 0, t_cnt = 0;
        double t_w = 0.0, t_v0 = 0.0, t_v1 = 0.0, t_v2 = 0.0, t_v3 = 0.0;
        int w_c0 = 0, w_c1 = 0, w_c2 = 0, w_c3 = 0, g_r0 = 0, g_r1 = 0, g_r2 = 0, g_r3 = 0;
        double w_v0 = 0.0, w_v1 = 0.0, w_v2 = 0.0, w_v3 = 0.0, g_v0 = 0.0, g_v1 = 0.0, g_v2 = 0.0, g_v3 = 0.0;
        auto load_class = [&](int c) {
            if (c == cur_cls) return;
            cur_cls = c;
            wave_sync();
            const double* Wp = S.blob + S.f_Wp[c];
            for (int e = lane; e < NZ * NZ; e += kWave) L.Hb[e] = Wp[e];
            const int t0 = S.f_tptr[c];
            t_cnt = S.f_tptr[c + 1] - t0;
            if (lane < t_cnt) {
                const int t = t0 + lane;
                t_ent = S.f_tent[t];
                t_w = Wp[t_ent];
                t_r0 = S.f_trow[4 * t], t_r1 = S.f_trow[4 * t + 1], t_r2 = S.f_trow[4 * t + 2], t_r3 = S.f_trow[4 * t + 3];
                t_v0 = S.f_tval[4 * t], t_v1 = S.f_tval[4 * t + 1], t_v2 = S.f_tval[4 * t + 2], t_v3 = S.f_tval[4 * t + 3];
                t_r0 = t_r0 < 0 ? 0 : t_r0, t_r1 = t_r1 < 0 ? 0 : t_r1, t_r2 = t_r2 < 0 ? 0 : t_r2, t_r3 = t_r3 < 0 ? 0 : t_r3;
            }
            if (lane < NZ) {
                const size_t at = ((size_t)c * NZ + lane) * 4;
                w_c0 = S.f_wcol[at], w_c1 = S.f_wcol[at + 1], w_c2 = S.f_wcol[at + 2], w_c3 = S.f_wcol[at + 3];
                w_v0 = S.f_wval[at], w_v1 = S.f_wval[at + 1], w_v2 = S.f_wval[at + 2], w_v3 = S.f_wval[at + 3];
                w_c0 = w_c0 < 0 ? 0 : w_c0, w_c1 = w_c1 < 0 ? 0 : w_c1, w_c2 = w_c2 < 0 ? 0 : w_c2, w_c3 = w_c3 < 0 ? 0 : w_c3;
                g_r0 = S.f_grow[at], g_r1 = S.f_grow[at + 1], g_r2 = S.f_grow[at + 2], g_r3 = S.f_grow[at + 3];
                g_v0 = S.f_gval[at], g_v1 = S.f_gval[at + 1], g_v2 = S.f_gval[at + 2], g_v3 = S.f_gval[at + 3];
            }
            wave_sync();
        };
        // gradient of stage k at the z in X:  g = W z + q (+ the InitialStateLMPC terms at stage 0) (+ A' c, c in Cb) -> gk.
        // q_k = - sum_rows w p a (costFunctions.cpp: the p-dependent part of the gradient; references shared by the batch) comes
        // from the plan's table, one stage ahead (`qn`: this stage's, fetched while the previous one ran)
        auto stage_gradient = [&](int k, double qk, bool with_rows, bool with_x0_terms) {
            if (lane < NZ) {
                const double* Xk = L.X + k * NZ;
                double g = qk + ((w_v0 * Xk[w_c0] + w_v1 * Xk[w_c1]) + (w_v2 * Xk[w_c2] + w_v3 * Xk[w_c3]));
                if (with_rows) {
                    const double* Ck = L.Cb + S.stage_row0[k];
                    g += (g_v0 * Ck[g_r0] + g_v1 * Ck[g_r1]) + (g_v2 * Ck[g_r2] + g_v3 * Ck[g_r3]);
                }
                if (with_x0_terms && k == 0 && lane < NX) {
                    g += L.G0[lane];
#pragma unroll
                    for (int l = 0; l < NX; ++l) g += L.H0[lane + NX * l] * L.X[l];
                }
                L.gk[lane] = g;
            }
        };
        auto fetch_q = [&](int k) -> double { return (lane < NZ && k >= 0) ? S.f_q[k * NZ + lane] : 0.0; };

        // ---- sweep 1: backward factorisation.  Before: X holds z, Cb the rows' gradient coefficients, Y their weights D (with_rows).
        //      After: stage records in KF / KV, P_0 in Pb, p_0 in pv0.  Returns false when a control block is not positive definite.
        // where this lane's initial values of the accumulators come from: an entry of H, the gradient of the stage, or zero
        const double* hxp[5];
        const double* hup[2];
#pragma unroll
        for (int I = 0; I < 5; ++I) {
            const int row = I < 3 ? 4 * I + q4 : (q4 < 3 ? NX + 3 * (I - 3) + q4 : -1);
            hxp[I] = L.Zs;
            if (row >= 0) {
                if (hb < 3)
                    hxp[I] = L.Hb + row + NZ * (4 * hb + r4);
                else if (r4 == 0)
                    hxp[I] = L.gk + row;
            }
        }
#pragma unroll
        for (int J = 0; J < 2; ++J) {
            hup[J] = L.Zs;
            if (qr3 && hb < 2) hup[J] = L.Hb + (NX + 3 * J + q4) + NZ * (NX + 3 * hb + r4);
        }
        double* const dummy = L.Zs + 2;
        auto sweep1 = [&](bool with_rows, bool with_x0_terms) -> bool {
            bool bad = false;
            double PX[3] = { 0.0, 0.0, 0.0 };
            double qn = fetch_q(N);
            for (int k = N; k >= 0; --k) {
                const int gi0 = S.stage_row0[k];
                load_class(S.cls_of_stage[k]); // (the stage Hessian starts from the padded W of its class; the entries the rows touch are rebuilt per stage)
                const double qk = qn;
                qn = fetch_q(k - 1);
                if (with_rows && lane < t_cnt) {
                    const double* Yk = L.Y + gi0;
                    L.Hb[t_ent] = t_w + ((t_v0 * Yk[t_r0] + t_v1 * Yk[t_r1]) + (t_v2 * Yk[t_r2] + t_v3 * Yk[t_r3]));
                }
                stage_gradient(k, qk, with_rows, with_x0_terms);
                wave_sync();
                double HX[5], HU[2];
#pragma unroll
                for (int I = 0; I < 5; ++I) HX[I] = *hxp[I];
#pragma unroll
                for (int J = 0; J < 2; ++J) HU[J] = *hup[J];
                if (k == N) { // P_N = H_xx, p_N = g_x
#pragma unroll
                    for (int I = 0; I < 3; ++I) PX[I] = HX[I];
                } else {
                    // P as the left factor: through LDS, which replicates it over the hardware blocks
#pragma unroll
                    for (int I = 0; I < 3; ++I) *(hb < 3 ? L.Pb + (4 * I + q4) + NX * (4 * hb + r4) : dummy) = PX[I];
                    wave_sync();
                    double aP[3][3];
#pragma unroll
                    for (int I = 0; I < 3; ++I)
#pragma unroll
                        for (int K = 0; K < 3; ++K) aP[I][K] = L.Pb[(4 * I + r4) + NX * (4 * K + q4)];
                    // T = P [A B]  (+ p in the gradient column)
                    double TX[3], TU[3];
#pragma unroll
                    for (int I = 0; I < 3; ++I) TX[I] = hb == 3 ? PX[I] : 0.0, TU[I] = 0.0;
#pragma unroll
                    for (int K = 0; K < 3; ++K)
#pragma unroll
                        for (int I = 0; I < 3; ++I) {
                            TX[I] = mfma_f64_4x4x4(aP[I][K], bA[K], TX[I]);
                            TU[I] = mfma_f64_4x4x4(aP[I][K], bBu[K], TU[I]);
                        }
                    // M = H + [A B]' T
                    double MX[5], MU[2];
#pragma unroll
                    for (int I = 0; I < 5; ++I) MX[I] = HX[I];
                    MU[0] = HU[0], MU[1] = HU[1];
#pragma unroll
                    for (int K = 0; K < 3; ++K) {
#pragma unroll
                        for (int I = 0; I < 3; ++I) MX[I] = mfma_f64_4x4x4(aMx[I][K], TX[K], MX[I]);
                        MX[3] = mfma_f64_4x4x4(aMu[0][K], TX[K], MX[3]);
                        MX[4] = mfma_f64_4x4x4(aMu[1][K], TX[K], MX[4]);
                        MU[0] = mfma_f64_4x4x4(aMu[0][K], TU[K], MU[0]);
                        MU[1] = mfma_f64_4x4x4(aMu[1][K], TU[K], MU[1]);
                    }
                    // ---- eliminate u_b (controls 3 .. 5): rows u_b of M -> LDS (left factor of the Schur update, the 3 x 3 block)
                    {
                        double* w1 = dummy;
                        if (q4 < 3) w1 = hb < 3 ? L.Rb + 20 * q4 + 4 * hb + r4 : (r4 == 0 ? L.Rb + 20 * q4 + 18 : dummy);
                        *w1 = MX[4];
                        double* w2 = dummy;
                        if (qr3 && hb < 2) w2 = L.Rb + 20 * q4 + NX + 3 * hb + r4;
                        *w2 = MU[1];
                    }
                    wave_sync();
                    double aRb[4];
#pragma unroll
                    for (int I = 0; I < 3; ++I) aRb[I] = *(q4 < 3 ? L.Rb + 20 * q4 + 4 * I + r4 : L.Zs);
                    aRb[3] = *(qr3 ? L.Rb + 20 * q4 + NX + r4 : L.Zs);
                    const double nB = neg_inv3(L.Rb + NX + 3, bad);
                    const double KbX = mfma_f64_4x4x4(nB, MX[4], 0.0);
                    const double KbU = mfma_f64_4x4x4(nB, MU[1], 0.0);
#pragma unroll
                    for (int I = 0; I < 4; ++I) MX[I] = mfma_f64_4x4x4(aRb[I], KbX, MX[I]);
                    MU[0] = mfma_f64_4x4x4(aRb[3], KbU, MU[0]);
                    double* const Fk = L.KF + k * KS;
                    {
                        double* w1 = dummy;
                        if (q4 < 3) w1 = hb < 3 ? Fk + oKb + 12 * q4 + 4 * hb + r4 : (r4 == 0 ? L.KV + 8 * k + 4 + q4 : dummy);
                        *w1 = KbX;
                        *((qr3 && hb == 0) ? Fk + oKba + 3 * q4 + r4 : dummy) = KbU;
                        *((qr3 && hb == 0) ? Fk + oNb + 3 * r4 + q4 : dummy) = nB;
                    }
                    // ---- eliminate u_a (controls 0 .. 2)
                    {
                        double* w1 = dummy;
                        if (q4 < 3) w1 = hb < 3 ? L.Ra + 20 * q4 + 4 * hb + r4 : (r4 == 0 ? L.Ra + 20 * q4 + 18 : dummy);
                        *w1 = MX[3];
                        *((qr3 && hb == 0) ? L.Ra + 20 * q4 + NX + r4 : dummy) = MU[0];
                    }
                    wave_sync();
                    double aRa[3];
#pragma unroll
                    for (int I = 0; I < 3; ++I) aRa[I] = *(q4 < 3 ? L.Ra + 20 * q4 + 4 * I + r4 : L.Zs);
                    const double nA = neg_inv3(L.Ra + NX, bad);
                    const double KaX = mfma_f64_4x4x4(nA, MX[3], 0.0);
#pragma unroll
                    for (int I = 0; I < 3; ++I) PX[I] = mfma_f64_4x4x4(aRa[I], KaX, MX[I]);
                    {
                        double* w1 = dummy;
                        if (q4 < 3) w1 = hb < 3 ? Fk + oKa + 12 * q4 + 4 * hb + r4 : (r4 == 0 ? L.KV + 8 * k + q4 : dummy);
                        *w1 = KaX;
                        *((qr3 && hb == 0) ? Fk + oNa + 3 * r4 + q4 : dummy) = nA;
                    }
                }
            }
            // P_0, p_0 for the step in x_0
            wave_sync();
#pragma unroll
            for (int I = 0; I < 3; ++I) *(hb < 3 ? L.Pb + (4 * I + q4) + NX * (4 * hb + r4) : (r4 == 0 ? L.pv0 + 4 * I + q4 : dummy)) = PX[I];
            wave_sync();
            return !bad;
        };
        // ---- sweep 3: backward VECTOR sweep through the stored factors.  Before: X holds z, Cb the rows' gradient coefficients.
        //      After: kv in KV, p_0 in pv0.
        auto sweep3 = [&]() {
            double pB[3] = { 0.0, 0.0, 0.0 };
            double qn = fetch_q(N);
            for (int k = N; k >= 0; --k) {
                load_class(S.cls_of_stage[k]);
                const double qk = qn;
                qn = fetch_q(k - 1);
                stage_gradient(k, qk, true, x0_free);
                wave_sync();
                if (k == N) {
#pragma unroll
                    for (int K = 0; K < 3; ++K) pB[K] = L.gk[4 * K + q4];
                    continue;
                }
                const double* Fk = L.KF + k * KS;
                // h = g + [A B]' p : hardware block b = row block (x_0..3 | x_4..7 | x_8..11 | u_a), u_b on its own (replicated)
                double hv = r4 == 0 ? *(hb < 3 ? L.gk + 4 * hb + q4 : (q4 < 3 ? L.gk + NX + q4 : L.Zs)) : 0.0;
                double hbv = (r4 == 0 && q4 < 3) ? L.gk[NX + 3 + q4] : 0.0;
                const double nBop = qr3 ? Fk[oNb + 3 * r4 + q4] : 0.0, nAop = qr3 ? Fk[oNa + 3 * r4 + q4] : 0.0;
                const double aKbT = *(q4 < 3 ? (hb < 3 ? Fk + oKb + 12 * q4 + 4 * hb + r4 : (r4 < 3 ? Fk + oKba + 3 * q4 + r4 : L.Zs)) : L.Zs);
                const double aKaT = *((q4 < 3 && hb < 3) ? Fk + oKa + 12 * q4 + 4 * hb + r4 : L.Zs);
#pragma unroll
                for (int K = 0; K < 3; ++K) {
                    hv = mfma_f64_4x4x4(hM[K], pB[K], hv);
                    hbv = mfma_f64_4x4x4(hMb[K], pB[K], hbv);
                }
                const double kvb = mfma_f64_4x4x4(nBop, hbv, 0.0); // kv_b = -M_bb^-1 h_b
                const double hp = mfma_f64_4x4x4(aKbT, hbv, hv); // h' = h + K_b' h_b  (x and u_a)
                const double ha = row_bcast_f64<12>(hp); // h'_a to every hardware block
                const double kva = mfma_f64_4x4x4(nAop, ha, 0.0); // kv_a = -M'_aa^-1 h'_a
                const double pn = mfma_f64_4x4x4(aKaT, ha, hp); // p = h'_x + K_a' h'_a
                if (r4 == 0 && q4 < 3 && hb < 2) L.KV[8 * k + 4 * hb + q4] = hb == 0 ? kva : kvb;
                pB[0] = row_bcast_f64<0>(pn), pB[1] = row_bcast_f64<4>(pn), pB[2] = row_bcast_f64<8>(pn);
            }
            wave_sync();
            if (r4 == 0 && hb < 3) L.pv0[4 * hb + q4] = hb == 0 ? pB[0] : hb == 1 ? pB[1] : pB[2];
            wave_sync();
        };
        // ---- forward sweep: dz_k into Y, from dx_0 in dx0[]; the controls through the stored gains, the states through [A B]
        auto forward = [&]() {
            double xB[3];
#pragma unroll
            for (int K = 0; K < 3; ++K) xB[K] = L.dx0[4 * K + q4];
            if (r4 == 0 && hb < 3) L.Y[4 * hb + q4] = hb == 0 ? xB[0] : hb == 1 ? xB[1] : xB[2];
            for (int k = 0; k < N; ++k) {
                const double* Fk = L.KF + k * KS;
                double fKa[3], fKb[3];
#pragma unroll
                for (int K = 0; K < 3; ++K) {
                    fKa[K] = *(r4 < 3 ? Fk + oKa + 12 * r4 + 4 * K + q4 : L.Zs);
                    fKb[K] = *(r4 < 3 ? Fk + oKb + 12 * r4 + 4 * K + q4 : L.Zs);
                }
                const double fKba = *(qr3 ? Fk + oKba + 3 * r4 + q4 : L.Zs);
                double ua = (r4 == 0 && q4 < 3) ? L.KV[8 * k + q4] : 0.0;
                double ub = (r4 == 0 && q4 < 3) ? L.KV[8 * k + 4 + q4] : 0.0;
                double xn = 0.0;
#pragma unroll
                for (int K = 0; K < 3; ++K) {
                    ua = mfma_f64_4x4x4(fKa[K], xB[K], ua);
                    ub = mfma_f64_4x4x4(fKb[K], xB[K], ub);
                    xn = mfma_f64_4x4x4(fA[K], xB[K], xn);
                }
                ub = mfma_f64_4x4x4(fKba, ua, ub);
                xn = mfma_f64_4x4x4(fA[3], ua, xn);
                xn = mfma_f64_4x4x4(fA[4], ub, xn);
                if (r4 == 0 && q4 < 3 && hb < 2) L.Y[k * NZ + NX + 3 * hb + q4] = hb == 0 ? ua : ub;
                if (r4 == 0 && hb < 3) L.Y[(k + 1) * NZ + 4 * hb + q4] = xn;
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
            double* GJ = L.GJ;
            const int w1 = nx + 1;
            for (int e = lane; e < nx * w1; e += kWave) {
                const int r = e / w1, cc = e - r * w1;
                GJ[e] = (cc < nx) ? L.Pb[r + NX * cc] + L.H0[r + NX * cc] : -L.pv0[r];
            }
            wave_sync();
            bool ok = true;
            for (int p = 0; p < nx; ++p) {
                const double piv = GJ[p * w1 + p];
                if (!(piv > 0.0)) ok = false;
                const double ip = 1.0 / piv;
                wave_sync();
                for (int e = lane; e < nx * w1; e += kWave) {
                    const int r = e / w1, cc = e - r * w1;
                    if (r == p || cc == p) continue;
                    GJ[e] -= GJ[r * w1 + p] * ip * GJ[p * w1 + cc];
                }
                wave_sync();
                for (int e = lane; e < nx * w1; e += kWave) {
                    const int r = e / w1, cc = e - r * w1;
                    if (r == p)
                        GJ[e] *= ip;
                    else if (cc == p)
                        GJ[e] = 0.0;
                }
                wave_sync();
            }
            if (lane < NX) L.dx0[lane] = lane < nx ? GJ[lane * w1 + nx] : 0.0;
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
            double qn = fetch_q(N);
            for (int k = N; k >= 0; --k) {
                load_class(S.cls_of_stage[k]);
                const double qk = qn;
                qn = fetch_q(k - 1);
                stage_gradient(k, qk, false, false);
                wave_sync();
                double acc = 0.0;
                if (lane < NX) {
                    acc = L.gk[lane];
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
            if (gi < m && (rinf[j] >> 28) == kRfIneq) {
                Sv[j] = fmax(Fr[j] - row_dot(rinf[j], roff[j], L.X), 1.0);
                Lam[j] = 1.0;
            }
        }
        stamp(0);
        // ------------------------------------------------------------------ 2. Newton iterations
        int it = 0;
        bool converged = false;
        for (it = 1; it <= S.max_iter && good; ++it) {
            // ---- bulk phase: residuals, barrier weights D (-> Y) and gradient coefficients c (-> Cb) of every row
            double musum = 0.0, maxr = 0.0;
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                const int gi = 64 * j + lane, fl = rinf[j] >> 28;
                double Dv = 0.0, Cv = 0.0, rp = 0.0;
                if (gi < m && fl != kRfOff) {
                    const double az = row_dot(rinf[j], roff[j], L.X);
                    if (fl == kRfEq) {
                        rp = az - Fr[j];
                        Dv = 1.0 / delta;
                        Cv = Lam[j] + rp / delta;
                    } else {
                        rp = az + Sv[j] - Fr[j];
                        Dv = Lam[j] / Sv[j];
                        Cv = Dv * rp;
                        maxr = fmax(maxr, fabs(rp));
                        musum += Sv[j] * Lam[j];
                    }
                }
                Rp[j] = rp;
                if (gi < m) L.Y[gi] = Dv, L.Cb[gi] = Cv;
            }
            wave_sync();
            const double mu = wave_sum(musum) * inv_mi;
            const double maxres = wave_max(maxr);
            stamp(1);
            // ---- sweep 1 (backward): factorisation and the predictor's right-hand side
            good = sweep1(true, x0_free) && good;
            stamp(2);
            if (!good) break;
            // ---- predictor: forward sweep, then the rows in bulk
            good = solve_x0() && good;
            forward();
            stamp(3);
            double amin = 1.0e300;
            // mu_aff = sum (s + a ds)(lam + a dl) / n  needs the step length a of the whole wave first: its three coefficients in a
            // are summed in the same pass (no second pass over the directions, no registers to keep them in)
            double q0 = 0.0, q1 = 0.0, q2 = 0.0;
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                const int gi = 64 * j + lane, fl = rinf[j] >> 28;
                double dsdl = 0.0;
                if (gi < m && fl == kRfIneq) {
                    const double adz = row_dot(rinf[j], roff[j], L.Y);
                    const double ds = -Rp[j] - adz;
                    const double dl = (-Lam[j] * Sv[j] - Lam[j] * ds) / Sv[j];
                    if (ds < 0.0) amin = fmin(amin, -Sv[j] / ds);
                    if (dl < 0.0) amin = fmin(amin, -Lam[j] / dl);
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
                const int gi = 64 * j + lane, fl = rinf[j] >> 28;
                double Cv = 0.0;
                if (fl == kRfEq)
                    Cv = Lam[j] + Rp[j] / delta; // as in the predictor
                else if (fl == kRfIneq)
                    Cv = (sigma_mu - Dd[j]) / Sv[j] + (Lam[j] / Sv[j]) * Rp[j];
                if (gi < m) L.Cb[gi] = Cv;
            }
            wave_sync();
            stamp(4);
            sweep3();
            good = solve_x0() && good;
            forward();
            stamp(5);
            // ---- final direction of the rows, step length, update.  The direction takes the place of values that are dead by now:
            //      inequality rows ds -> Rp, dl -> Dd; equality rows keep their residual in Rp, a' dz -> Dd
            amin = 1.0e300;
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                const int gi = 64 * j + lane, fl = rinf[j] >> 28;
                if (gi < m && fl != kRfOff) {
                    const double adz = row_dot(rinf[j], roff[j], L.Y);
                    if (fl == kRfEq) {
                        Dd[j] = adz;
                    } else {
                        const double ds = -Rp[j] - adz;
                        const double dl = ((sigma_mu - Dd[j]) - Lam[j] * Sv[j] - Lam[j] * ds) / Sv[j];
                        if (ds < 0.0) amin = fmin(amin, -Sv[j] / ds);
                        if (dl < 0.0) amin = fmin(amin, -Lam[j] / dl);
                        Rp[j] = ds, Dd[j] = dl;
                    }
                }
            }
            amin = -wave_max(-amin);
            const double tau = mu > 1e-10 ? 0.995 : 0.9999;
            const double alpha = amin < 1.0 ? fmin(1.0, tau * amin) : 1.0;
            double z_inf = 0.0, step_inf = 0.0;
            for (int e = lane; e < NE; e += kWave) {
                const double dz = L.Y[e];
                const double zn = L.X[e] + alpha * dz;
                L.X[e] = zn;
                step_inf = fmax(step_inf, fabs(dz));
                z_inf = fmax(z_inf, fabs(zn));
            }
            step_inf = wave_max(step_inf) * alpha;
            z_inf = wave_max(z_inf);
            double musum2 = 0.0, maxe = 0.0;
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                const int fl = rinf[j] >> 28;
                if (64 * j + lane < m) {
                    if (fl == kRfIneq) {
                        Sv[j] += alpha * Rp[j];
                        Lam[j] += alpha * Dd[j];
                        musum2 += Sv[j] * Lam[j];
                    } else if (fl == kRfEq) {
                        const double re = Rp[j] + alpha * Dd[j]; // residual of the row at the new point
                        Lam[j] += re / delta;
                        maxe = fmax(maxe, fabs(re));
                    }
                }
            }
            wave_sync();
            const double mu_new = wave_sum(musum2) * inv_mi;
            const double res_new = fmax((1.0 - alpha) * maxres, wave_max(maxe));
            if (!(mu_new == mu_new) || !(step_inf == step_inf)) {
                good = false;
                break;
            }
            stamp(6);
            if (res_new <= 1e-9 && ((step_inf <= 1e-10 * (1.0 + z_inf) && mu_new <= 1e-8) || mu_new <= 1e-15)) {
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
            stamp(6);
            prof[7] = cycle_counter() - tstart;
            for (int q = 0; q < 8; ++q) P.prof[8 * (size_t)inst + q] = prof[q];
        }
        wave_sync();
    }
}

} // namespace copra_hip
