// Riccati form of the factor of the condensed LMPC Hessian (factor-only tier of the fused kernel, lmpc_fused_ric.hpp).
//
// The condensed Hessian Q = Psi' W Psi + N'WN + 1e-6 I (LMPC.cpp:228-230, 252-255 after costFunctions.cpp:63-215) of a
// controller whose costs are all per-step entries is the Hessian of a stage-wise LQ problem.  One backward Riccati sweep
// over the N stages (O(N nx^3) instead of the O(n^3) Cholesky of the n x n matrix) gives, per stage k,
//     Lam_k Lam_k' = M_uu,k            (Cholesky of the nu x nu control block)
//     K_k = -M_uu,k^-1 M_ux,k ,  Acl_k = A + B K_k ,  Bt_k = B Lam_k^-T
// and with them a factor Q^-1 = Rinv Rinv' whose two products are closed-loop recursions over the stages:
//     z = Rinv v   :  xi_0 = 0;   z_k = K_k xi_k + Lam_k^-T v_k;   xi_{k+1} = Acl_k xi_k + Bt_k v_k          (forward)
//     w = Rinv' n  :  mu_N = 0;   w_k = Lam_k^-1 n_k + Bt_k' mu_{k+1};   mu_k = Acl_k' mu_{k+1} + K_k' n_k   (backward)
// These are the ONLY two operations the factor-only Goldfarb-Idnani iteration applies to the factor (gi_core.hpp, TRI:
// w = R^-T n+, z = R^-1 (w - Q1 d1)); qpgen2's J = R^-1 [Q1 Q2] differs from this choice of R by an orthogonal factor that
// cancels in every quantity the method looks at, so the steps, the active sets and the iterates are the same up to rounding.
// (Check of the algebra in numpy: tools/exp/ric_factor_proto.py.)
//
// Bt is never stored: with t_k = Lam_k^-T v_k formed beforehand the forward recursion is  z_k = K_k xi_k + t_k,
// xi_{k+1} = Acl_k xi_k + B t_k  (the CONSTANT B in place of Bt_k), and the backward one yields s_k = n_k + B' mu_{k+1} with
// w_k = Lam_k^-1 s_k afterwards -- 18 doubles less per stage, which is what lets a ninth instance share a CU.
//
// Stage record (RicRec<NX, NU>::SZ doubles, N of them in the J region of the LDS layout):
//     Acl (NX x NX, column-major) | K (NU x NX, column-major) | Li = Lam^-1 (lower triangular, packed by rows)
//     (the feed-forward terms kv of the unconstrained minimiser: a block of their own, see RicRec)
// followed, once, by the constant block (RicRec::CST doubles):  B (NX x NU, column-major) | d (NX) | a zero | a spare double | a one
// (the one: the identity block that passes t_k / n_k through to the output rows of the backward recursion; the forward one and
//  the roll-out keep their whole K-block 2, [B; I] resp. [B d; I 0], in a register -- it is the same at every stage)
#pragma once

#ifndef COPRA_RIC_UNROLL
#define COPRA_RIC_UNROLL 2 // stages per loop body of the fixed-length recursions (forward, roll-out)
#endif

namespace copra_hip {

// 1 / x: v_rcp_f64 seed + two Newton steps
COPRA_DEV double ric_rcp(double x)
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

template <int NX, int NU>
struct RicRec {
    static constexpr int oAcl = 0;
    static constexpr int oK = oAcl + NX * NX;
    static constexpr int oLi = oK + NU * NX; // (Lam^-1 packed: entry (r, c), c <= r, at r (r + 1) / 2 + c)
    // (Round 5: the feed-forward terms kv left the record -- only the roll-out reads them, ONCE, and behind the one-instance-per-lane pass
    //  nobody does.  They wait in a block of their own, NU per stage: LdsLayout::ricKv in LDS -- in the compact variant the tail of the
    //  trajectory's place, which the roll-out overwrites only behind its own reads --, FusedPlan::ric_model + oBk in the shared-model
    //  buffer.  Four doubles less per stage at the headline shape: 13.3 instead of 13.9 KB of LDS, the twelfth instance per CU.)
    static constexpr int SZ = (oLi + NU * (NU + 1) / 2 + 1) & ~1;
    // the constant block behind the N records
    static constexpr int cB = 0;
    static constexpr int cD = cB + NX * NU;
    static constexpr int cZ = cD + NX; // holds 0.0: what structurally zero operands read
    static constexpr int cS = cZ + 1; // nobody reads it: what lanes with nothing to store write
    static constexpr int cO = cS + 1; // holds 1.0: the identity block of the stacked matrix
    static constexpr int CST = (cO + 1 + 1) & ~1;
};

// ---- the two products on the matrix cores ------------------------------------------------------------------------
// One stage of either recursion is a small matrix-vector product,  [xi+; z_k] = [Acl Bt; K Lam^-T] [xi; v_k]  (and its
// transpose for w), and the MFMA instructions have the property that makes a CHAIN of them free of data movement: the
// result of one product is laid out exactly as the B operand of the next one.  So the recursion state never leaves the
// accumulator registers -- no v_readlane broadcasts (18 per stage in a vector-ALU version), no multiply-adds on the vector
// ALU (18 per stage).  (History: a vector-ALU version and one on v_mfma_f64_16x16x4 -- 64 cycles of the matrix pipe per
// instruction on MI355X, no faster than the vector ALU -- preceded the one below; DESIGN.md 3.9b has their numbers.)
// Stacked index:  0 .. NX-1 state components (NX <= 8) | 8 .. 8+NU-1 input / output components.
// -> offset (doubles from the first record) of element (s_out, s_in) of the stacked matrix [Acl B; K I] of the forward
// recursion (its identity block excepted); kmul = RicRec::SZ if the element belongs to the stage record (add kmul * stage),
// 0 for the constant block.
// NH == 0: the horizon is a run-time value, nh_rt (the builds the library ships for EVERY horizon of a shape; NH > 0: the builds of the
// BASELINE shapes, whose loops have compile-time trip counts).
template <int NX, int NU, int NH, bool IDENT = false>
COPRA_DEV int ric_stack_offset(int s_out, int s_in, int& kmul, int nh_rt = 0)
{
    using RR = RicRec<NX, NU>;
    static_assert(NX <= 8 && NU >= 1 && NU <= 4, "stacked layout of the MFMA recursions");
    const int cbase = (NH ? NH : nh_rt) * RR::SZ;
    kmul = 0;
    const int to = (s_out < NX) ? 0 : (s_out >= 8 && s_out < 8 + NU) ? 1 : 2;
    const int ti = (s_in < NX) ? 0 : (s_in >= 8 && s_in < 8 + NU) ? 1 : 2;
    if (IDENT && to == 1 && s_out == s_in) return cbase + RR::cO; // the identity block: a constant 1.0
    if (to == 2 || ti == 2 || (to == 1 && ti == 1)) return cbase + RR::cZ;
    const int a = to == 0 ? s_out : s_out - 8, b = ti == 0 ? s_in : s_in - 8;
    if (ti == 1) return cbase + RR::cB + a + NX * b; // B(a, b)
    kmul = RR::SZ;
    return to == 0 ? RR::oAcl + a + NX * b : RR::oK + a + NU * b; // Acl(a, b) | K(a, b)
}

// ---- on v_mfma_f64_4x4x4_4b_f64 ----------------------------------------------------------------------------------
// (A 16 x 16 x 4 FP64 MFMA occupies the matrix pipe of its SIMD for 64 cycles: profiles/r02/mfma_f64_4x4x4_probe.txt.)  The
// 4 x 4 x 4 instruction computes four independent 4 x 4 blocks in 16 cycles: block b of instruction J multiplies rows 4b .. 4b+3 of the stacked matrix by K-block
// J of the stacked vector, the three instructions of a stage accumulate in place.  Result rows 4b + i sit in quad b of lane
// row i; the next stage needs K-block J in EVERY quad of row k -- one DPP row broadcast of lane 4 J per block.
// lane = 16 q + 4 b + r:  A operand (row 4b + r, column 4J + q), B operand: component 4J + q of the vector in all lanes of row q.
// nstages (TR only): the backward recursion starts at stage nstages - 1 -- the normal of a constraint at step k has no
// component beyond stage k - 1, mu is zero until then and so is w beyond it (X must hold zeros there: see the caller).
// in / X (may be the same NU NH doubles of LDS): the input vector (v for the forward recursion, n for the backward one) and
// the hand-over buffer; returns component `lane` of z = Rinv v (TR = false) or of w = Rinv' n (TR = true).
// XI (forward recursion only): the closed-loop states xi_0 .. xi_NH are stored there, NX per stage.
// dummy: one double of LDS nobody reads -- the lanes that have nothing to store write there, so that the loop body has no
// branches (with the stores in exec-masked blocks the compiler waits for ALL outstanding LDS operations at every stage,
// stores included, instead of just for the prefetched operands).
template <int NX, int NU, int NH, bool TR>
// inj_stage > 0 (backward recursion only): the normal has a state part  inj_val e_comp  at step inj_stage -- a row of Psi.  Its
// Psi' e is never formed: with lambda the adjoint of the OPEN loop (lambda_k = e, lambda_j = A' lambda_{j+1}, n_j = B' lambda_{j+1})
// the sums nu = mu + lambda obey the same recursion, nu_j = Acl_j' nu_{j+1}, s_j = B' nu_{j+1}: the unit vector simply joins the
// state that enters stage inj_stage - 1 (the caller makes nstages >= inj_stage).
COPRA_DEV double ric_apply_mfma4(const double* F, const double* in, double* X, double* dummy, int nstages = NH, double* XI = nullptr,
    int inj_stage = 0, int inj_comp = 0, double inj_val = 0.0, int nh_rt = 0)
{
    using RR = RicRec<NX, NU>;
    const int nh = NH ? NH : nh_rt; // (NH == 0: the horizon is a run-time value)
    const int lane = lane_id(), q = lane >> 4, b = (lane >> 2) & 3, r = lane & 3;
    const int kl = lane / NU, cl = lane - kl * NU; // this lane's (stage, component) of the NU NH vectors
    const bool mine = lane < NU * nh;
    if (!TR) { // t_k = Lam_k^-T v_k, in place of v
        double t = 0.0;
        if (mine) {
#pragma unroll
            for (int c = 0; c < NU; ++c) // (Lam^-T)(cl, c) = Lam^-1(c, cl): zero above the diagonal
                t += (c >= cl ? F[kl * RR::SZ + RR::oLi + c * (c + 1) / 2 + (c >= cl ? cl : 0)] : 0.0) * in[NU * kl + c];
        }
        wave_sync();
        if (mine) X[lane] = t;
        wave_sync();
        in = X;
    }
    int off[3], km[3];
#pragma unroll
    for (int J = 0; J < 3; ++J)
        off[J] = TR ? ric_stack_offset<NX, NU, NH, true>(4 * J + q, 4 * b + r, km[J], nh) : ric_stack_offset<NX, NU, NH, true>(4 * b + r, 4 * J + q, km[J], nh);
    // The loop is bound by instruction ISSUE, not latency (several instances share a SIMD): every operand has a pointer of its
    // own that moves by a per-lane stride (one addition per stage; a constant operand has stride 0), the outputs of a stage --
    // rows 8.. (block 2) and, for the forward recursion, the new state (blocks 0 and 1: the lanes that own a result row hold it
    // before the broadcast) -- leave through ONE store, the injection is a wave-uniform branch, and the operands of the next
    // stage are fetched into the registers of the current one right after their last use (no copies).  The fetches behind the
    // last stage read one record past the end (inside the instance's LDS; unused).
    const int count = TR ? uniform_i32(nstages) : nh;
    const int kfirst = TR ? count - 1 : 0;
    const bool writer = q < NU && b == 2 && r == 0; // rows 8 + q: the outputs
    const bool xwriter = !TR && XI && r == 0 && b < 2 && 4 * b + q < NX; // the state: component 4 b + q sits in lane row q
    const double* p0 = F + off[0] + km[0] * kfirst;
    const double* p1 = F + off[1] + km[1] * kfirst;
    const double* p2 = F + off[2] + km[2] * kfirst;
    const double* pv = in + (q < NU ? q : 0) + NU * kfirst;
    const int d0 = TR ? -km[0] : km[0], d1 = TR ? -km[1] : km[1], d2 = TR ? -km[2] : km[2], dv = TR ? -NU : NU;
    double* sp = writer ? X + q + NU * kfirst : xwriter ? XI + NX + 4 * b + q : dummy;
    const int sst = writer ? dv : xwriter ? NX : 0;
    if (xwriter) XI[4 * b + q] = 0.0; // xi_0
    double s0 = 0.0, s1 = 0.0; // K-blocks 0 and 1 of the state (stacked components 0..3 and 4..7), one per lane row
    const double inj0 = (q == inj_comp) ? inj_val : 0.0, inj1 = (4 + q == inj_comp) ? inj_val : 0.0;
    const int tinj = TR ? count - uniform_i32(inj_stage) : -1; // (stage inj_stage - 1)
    double a0 = *p0, a1 = *p1, a2 = *p2, vk = *pv;
    auto stage = [&]() {
        double y = mfma_f64_4x4x4(a2, vk, 0.0); // (does not wait for the previous stage)
        pv += dv;
        vk = *pv;
        if (TR) { // (forward recursion: K-block 2 of the matrix is [B; I], the same at every stage)
            p2 += d2;
            a2 = *p2;
        }
        y = mfma_f64_4x4x4(a0, s0, y);
        p0 += d0;
        a0 = *p0;
        y = mfma_f64_4x4x4(a1, s1, y);
        p1 += d1;
        a1 = *p1;
        *sp = y;
        sp += sst;
        s0 = row_bcast_f64<0>(y);
        s1 = row_bcast_f64<4>(y);
    };
    if (TR) { // two loops around the injection: nothing of it inside them
        for (int t = 0; t < tinj; ++t) stage();
        if (tinj < count) {
            s0 += inj0;
            s1 += inj1;
        }
        for (int t = tinj; t < count; ++t) stage();
    } else {
#pragma unroll COPRA_RIC_UNROLL
        for (int t = 0; t < count; ++t) stage();
    }
    wave_sync();
    if (!TR) return mine ? X[lane] : 0.0;
    // w_k = Lam_k^-1 s_k  (s = n where the recursion did not go: zero there, and so is w)
    double w = 0.0;
    if (mine) {
#pragma unroll
        for (int c = 0; c < NU; ++c) w += (c <= cl ? F[kl * RR::SZ + RR::oLi + cl * (cl + 1) / 2 + (c <= cl ? c : 0)] : 0.0) * X[NU * kl + c];
    }
    wave_sync(); // (every lane has read s before the caller reuses the buffer)
    return w;
}

} // namespace copra_hip
