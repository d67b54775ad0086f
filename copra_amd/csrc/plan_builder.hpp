// plan_builder.hpp -- host-side compilation of copra cost / constraint descriptors into a FusedPlan.
// Pure C++ (no HIP): used by the C-ABI library and by the CPU wave-emulator harness under tests/emu/.
//
// Mirrors the dimension checks of the reference's initializeCost / initializeConstraint
// (src/costFunctions.cpp:44-61, 88-98, 122-137, 173-193; src/constraints.cpp:45-64, 106-135, 171-195, 263-282,
// 333-357) -> COPRA_ERR_DOMAIN, and the stacking order of LMPC::makeQPForm (src/LMPC.cpp:250-280).
#pragma once

#include "../../include/copra_hip.h"
#include "plan.hpp"

#include <cfloat>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>

namespace copra_hip {

// the process-wide default options (copra_set_default_options); built-in: everything zero = the engine decides
inline copra_options_t& default_options()
{
    static copra_options_t d = [] {
        copra_options_t o {};
        o.struct_size = (int)sizeof(copra_options_t);
        return o;
    }();
    return d;
}
// a caller's struct may be shorter (built against an older header): the fields it does not have keep the defaults
inline copra_options_t resolve_options(const copra_options_t* given)
{
    copra_options_t o = default_options();
    if (given) {
        size_t len = given->struct_size > 0 ? (size_t)given->struct_size : sizeof(copra_options_t);
        if (len > sizeof(copra_options_t)) len = sizeof(copra_options_t);
        std::memcpy(&o, given, len);
        o.struct_size = (int)sizeof(copra_options_t);
    }
    return o;
}

struct HostPlan {
    copra_options_t opt = default_options(); // engine options of this controller (set BEFORE build_plan; include/copra_hip.h)
    FusedPlan plan {}; // pointers refer to the vectors below (host addresses)
    std::vector<double> params;
    std::vector<int> row_step, row_ekind, row_eoff, row_gkind, row_goff;
    std::vector<double> row_f;
    std::vector<int> row_prev; // the same row one step earlier, -1: none (warm start of the receding-horizon path)
    std::vector<double> lb, ub;
    std::vector<double> isR, isr; // InitialStateLMPC: R (nx x nx), r (nx)
    std::string error;
    size_t lds_bytes = 0; // of plan.lds (the layout of the first launch)
    // two-tier execution: plan.lds is the compact layout (4 waves per CU) when it fits, lds_full the always-sufficient
    // one used by the second launch for the instances that overflowed R
    bool two_tier = false;
    LdsLayout lds_full {};
    size_t lds_full_bytes = 0;
    // run-time shapes: plan.lds may be a DENSER compact layout than the safe choice below; copra_batch_solve falls back
    // to the safe one when too many instances of the first solves overflow into the second tier
    bool dense = false;
    LdsLayout lds_safe {};
    bool safe_two_tier = false;
    bool large = false; // more than 64 decision variables: workgroup-per-instance kernel (lmpc_large.hpp)
    bool ric_only = false; // beyond the condensed kernels' sizes (more than 512 decision variables; InitialStateLMPC with xDim > 16):
                           // only the stage-wise Riccati interior-point kernels cover it, and only if the controller is stage-wise
    // where the rows of constraint k (position in the user's array) sit in the stacked order: row = row0 + s * per_step
    // + i for its steps s and lines i (steps == 1 for a full-size entry); row0 < 0: bound constraint, no rows
    std::vector<int> cstr_row0, cstr_per_step, cstr_steps;
    std::vector<int> cost_slot; // position of the user's cost k among the kernel-evaluated cost terms, -1: a dense cost
    // the (instance, axis)-per-lane solver's tables for AXIS-MAJOR state order (state i on axis i / nxa: x = (p_x, v_x, p_y, v_y, ..)); the plan's
    // own fields hold the set for component-major order (state i on axis i % nu: x = (p, v)), what FusedPlan::axis_order = 0 means
    int axis1_tab = -1, axis1_cref = -1, axis1_rpa = 0, axis1_const = 0;
};

// Which order are the states of a system of decoupled chains in?  0: state i on axis i % nu (the benchmark's CoM model: x = (p, v)), 1: state i on
// axis i / (nx / nu) (x = (p_x, v_x, p_y, v_y, ..)), -1: neither -- from the zero pattern of ONE system (column-major A: nx x nx, B: nx x nu).
inline int axis_order_of(const double* A, const double* B, int nx, int nu)
{
    if (nu < 1 || nx % nu != 0) return -1;
    const int nxa = nx / nu;
    for (int ord = 0; ord < 2; ++ord) {
        auto axis_of = [&](int i) { return ord ? i / nxa : i % nu; };
        bool ok = true;
        for (int j = 0; j < nx && ok; ++j)
            for (int i = 0; i < nx && ok; ++i)
                if (A[(size_t)i + (size_t)nx * j] != 0.0 && axis_of(i) != axis_of(j)) ok = false;
        for (int c = 0; c < nu && ok; ++c)
            for (int i = 0; i < nx && ok; ++i)
                if (B[(size_t)i + (size_t)nx * c] != 0.0 && axis_of(i) != c) ok = false;
        if (ok) return ord;
    }
    return -1;
}

// qpgen2's "vsmall": smallest 1e-60 * 2^k with 1 + 0.1 vsmall > 1 and 1 + 0.2 vsmall > 1
inline double qpgen2_vsmall()
{
    volatile double vsmall = 1.0e-60, ta, tb;
    do {
        vsmall = vsmall + vsmall;
        ta = 1.0 + 0.1 * vsmall;
        tb = 1.0 + 0.2 * vsmall;
    } while (ta <= 1.0 || tb <= 1.0);
    return vsmall;
}

inline int align2(int v) { return (v + 1) & ~1; }

// LDS carve-up for a solver working on n variables with mgen general rows.
//   compact == false: every region has its own space, R holds n columns (always sufficient).
//   compact == true : the preview-phase tables (A, B, d, x0, Phi, xi) borrow the not-yet-used J region, the cost-phase
//                     tables (Y, We, cost parameters) borrow the not-yet-used solver vectors, and R gets whatever is
//                     left of `budget` doubles (at least 1 column).  Returns false if even that does not fit.
//   tri (with compact): factor-only layout (LdsLayout::tri) -- the packed triangle instead of the n x ldj square, no
//                     dv / Givens coefficients, and rcap columns of Q1 (64 doubles each) next to R; the cost-phase
//                     tables may run over Q1 / R as well (nothing of the active set exists yet).
//   xcur_late (with tri): the plan never reads the trajectory during the active-set loop (FusedPlan::rows_direct), so
//                     the trajectory written at the very end takes the place of the then dead factor.
//   q1regs (with tri): the first-tier kernel keeps that many columns of Q1 in registers (gi_core.hpp, QR): no Q1 region,
//                     rcap = q1regs.
inline bool layout_lds(LdsLayout& L, int nx, int nu, int N, int n, int X, int rmax, int mgen, int meq, int mtotal,
    bool fused, bool compact = false, int budget = 0, int rfull = 0, bool tri = false, bool xcur_late = false, int q1regs = 0,
    bool no_y = false)
{
    // no_y: every state cost of the controller has M = I (CostTerm::ident): the cost phase reads the blocks G_k instead of
    // forming Y_k = M G_k, so the Y table does not exist (the high-water mark of the factor-only layout is the cost phase)
    int o = 0;
    auto take = [&](int count) {
        int at = o;
        o += align2(count);
        return at;
    };
    L.ldj = (n % 2 == 0) ? n + 1 : n; // odd leading dimension: row-per-lane and column-per-lane reads conflict-free
    const int sizeJ = tri ? align2(n * (n + 1) / 2) : align2(n * L.ldj);
    L.tri = tri ? 1 : 0;
    L.Q1 = 0;
    L.q1regs = (tri && q1regs > 0) ? q1regs : 0;
    const int sizePrev = fused ? align2(nx * nx) + align2(nx * nu) + 2 * align2(nx) + align2((N + 1) * nx * nx) + align2(X) : 0;
    const int sizeFull = rfull > 0 ? align2(rfull) : 0; // weighted residuals (the 16 x 64 tile of tmp stays in registers: full_size_cost_term)
    const int sizeY = no_y ? 0 : align2(N * rmax * nu);
    const int sizeCost = fused ? sizeY + align2((N + 1) * rmax) + align2(rmax * (nx + nu + 2)) + sizeFull : 0;
    if (fused) {
        L.G = take(N * nx * nu);
        L.Xbar = take(X);
        L.Xcur = (tri && xcur_late) ? -1 : take(X);
    } else {
        L.G = L.Xbar = L.Xcur = 0;
    }
    L.J = take(sizeJ > 0 ? sizeJ : 2);
    if (L.Xcur < 0) L.Xcur = L.J;
    if (compact && sizePrev > sizeJ) take(sizePrev - sizeJ); // (never for the shapes of interest)
    int prev0 = L.J; // preview tables alias J in the compact layout
    if (fused && !compact) prev0 = take(sizePrev);
    if (fused) {
        int q = prev0;
        L.A = q, q += align2(nx * nx);
        L.B = q, q += align2(nx * nu);
        L.D = q, q += align2(nx);
        L.X0 = q, q += align2(nx);
        L.BldPhi = q, q += align2((N + 1) * nx * nx);
        L.BldXi = q, q += align2(X);
    } else {
        L.A = L.B = L.D = L.X0 = L.BldPhi = L.BldXi = 0;
    }
    // solver vectors (the cost tables alias them in the compact layout)
    const int vec0 = o;
    L.xs = take(n);
    L.dv = tri ? L.xs : take(n);
    L.zv = L.dv; // (unused scratch name kept for the carve helper)
    if (!compact) L.uv = take(n + 2); // (compact: sized by rcap, below)
    L.ap = take(n);
    L.cvec = compact ? L.ap : take(n); // c is consumed by the factorisation before ap is first written
    L.coef = tri ? L.xs : take(4 * n); // (factor-only: 1 / R(i,i) sits on the diagonal of the packed factor; never touched)
    L.nb = take(mgen > 0 ? mgen : 1);
    L.eqsgn = take(meq > 0 ? meq : 1);
    L.scal = take(2);
    L.act = take((mtotal + 7) / 8 + 1); // one byte per row
    if (!compact) L.iact = take((n + 2) / 2 + 1);
    if (compact && !tri && sizeCost > o - vec0) take(sizeCost - (o - vec0));
    int cost0 = vec0;
    if (fused && !compact) cost0 = take(sizeCost);
    if (fused) {
        int q = cost0;
        L.BldY = q, q += sizeY;
        L.BldWe = q, q += align2((N + 1) * rmax);
        L.BldCp = q, q += align2(rmax * (nx + nu + 2));
        L.BldFull = q;
    } else {
        L.BldY = L.BldWe = L.BldCp = L.BldFull = 0;
    }
    // R: n columns, or what the budget leaves
    int rcap = n;
    if (compact) {
        const int left = budget - o - 2;
        const int per_col = (tri && q1regs == 0) ? kWave : 0;
        // a column of R, its multiplier (uv), its row index (iact) and, factor-only, its column of Q1
        auto need = [&](int r) { return r * per_col + r * (r + 1) / 2 + align2(r + 2) + align2((r + 2) / 2 + 1); };
        rcap = 0;
        while (rcap < n && need(rcap + 1) <= left) ++rcap;
        if (q1regs > 0) {
            if (rcap < q1regs) return false;
            rcap = q1regs;
        }
        if (rcap < 1) return false;
        L.uv = take(rcap + 2);
        L.iact = take((rcap + 2) / 2 + 1);
    }
    L.rcap = rcap;
    if (tri && q1regs == 0) L.Q1 = take(rcap * kWave);
    L.R = take(rcap * (rcap + 1) / 2 + 2);
    if (tri && o < vec0 + sizeCost) o = vec0 + sizeCost;
    L.total = o;
    return true;
}

// Layout of the Riccati-factor tier (lmpc_fused_ric.hpp): the J region holds N stage records (ric_factor.hpp) and, before
// them, nothing (the lean preview of that body writes G and Xbar in place); A / B / d / x0 keep their own slots because the
// sweep reads them while it fills the records; no cost tables; the sweep's scratch aliases the solver vectors.
// q1regs > 0: that many columns of Q1 in registers (rcap = q1regs);  q1regs == 0: Q1 in LDS, as many columns as `budget` leaves.
// shapes lmpc_fused_ric_body<NX, NU, NH> can be instantiated for (its static_asserts)
inline bool ric_shape_ok(int nx, int nu, int N)
{
    const int nz = nx + nu, nxx = nx * (nx + 1) / 2, nux = nu * nx, nuu = nu * (nu + 1) / 2;
    return nu >= 1 && nu <= 3 && nx >= 1 && nx <= 7 && nx * (nz + 1) <= kWave && nxx + nux + nuu + nz <= kWave && nu * N <= kWave
        && nxx + nux + nuu >= nx * (nx + 1) && N >= 2 && nx >= nu; // (nx >= nu: where the roll-out keeps kv, layout_lds_ric)
}

// Shapes whose Riccati-factor tier (lmpc_fused_ric.hpp) and one-instance-per-lane pass (lmpc_lane.hpp) the LIBRARY holds for every horizon
// (run-time value, NU N <= 64; copra_hip_ric.hip): the double integrators in one, two and three dimensions -- the reference's falling
// mass (tests/systems.h:42-229), a planar point mass, the CoM system of binding/python/tests/pyTests.py:342-359.  (6, 3) at N = 10, 15, 20
// additionally has builds with a compile-time horizon (copra_hip.hip).  Every other shape: copra_batch_specialise.
// Shapes and horizons the library holds the one-(instance, axis)-per-lane solver for (lmpc_axis.hpp): chains of two states and one control --
// the double integrators in two and three dimensions (in ONE dimension the horizons its lanes have registers for are the packed kernels'
// and the factor-only tiers': 32 variables and fewer).  -> the build's largest horizon, 0: none.
constexpr int kAxisQmax = 6; // active constraints per (instance, axis) its lanes have room for
constexpr int kAxisQmaxBig = 16; // ... and the lanes of the second chance of what it lists (copra_lmpc_axis_list_kernel)
inline int axis_solver_nmax(int nx, int nu, int N)
{
    // chains of two states per control in two and three dimensions (the CoM model, planar point masses) at horizons up to 20 (two dimensions: 31);
    // since late round 6 also chains of THREE states per control (the jerk-controlled CoM model: position, velocity, acceleration per axis)
    // and of ONE (kinematic models) in two and three dimensions, up to 20 steps
    if (nu < 2 || nu > 3 || N < 1 || nx % nu != 0) return 0;
    const int nxa = nx / nu;
    if (nxa == 2) return N <= 20 ? 20 : (nu == 2 && N <= 31) ? 31 : (nu == 3 && N == 21) ? 21 : 0; // (three axes: 3 N <= 64 variables -- N = 21 is the last horizon of the one-wave kernels)
    if (nxa == 1) return N <= 20 ? 20 : (nu == 2 && N <= 31) ? 31 : 0; // (one state per control: kinematic models, x+ = a x + b u per axis)
    if (nxa == 3) return N <= 20 ? 20 : 0;
    return 0;
}
inline bool ric_aot_shape(int nx, int nu) { return (nx == 6 && nu == 3) || (nx == 4 && nu == 2) || (nx == 2 && nu == 1); }
inline bool ric_aot_exact(int nx, int nu, int N) { return nx == 6 && nu == 3 && (N == 10 || N == 15 || N == 20); }

inline bool layout_lds_ric(LdsLayout& L, int nx, int nu, int N, int n, int X, int mgen, int meq, int mtotal, bool xcur_late,
    int q1regs, int budget)
{
    int o = 0;
    auto take = [&](int count) {
        int at = o;
        o += align2(count);
        return at;
    };
    L = LdsLayout {};
    const int rec = (nx * nx + nx * nu + nu * (nu + 1) / 2 + 1) & ~1; // RicRec<NX, NU>::SZ
    const int cst = (nx * nu + nx + 3 + 1) & ~1; // RicRec<NX, NU>::CST
    L.ldj = (n % 2 == 0) ? n + 1 : n;
    L.tri = 1;
    L.ric = 1;
    L.q1regs = q1regs;
    L.rcap = q1regs;
    // compact variant (xcur_late <=> FusedPlan::rows_direct): trajectory and closed-loop states inside the G region (LdsLayout::ricC)
    // (there the blocks G are never stored: their block-row norms are taken by the preview steps, nothing else needs them;
    //  and A | B | d | x0 share the solver vectors' place behind the sweep's scratch -- they are dead before those are written)
    const bool compact = xcur_late; // (callers pass rows_pure && !copra_options_t::ric_general)
    const int scratch = align2(nx * nx) + align2(nx) + align2(nu * 12) + 2; // P | p | rows u of M | zero, spare
    L.ricC = compact ? 1 : 0;
    L.G = take(compact ? 2 * align2(X) : N * nx * nu);
    L.Xbar = compact ? L.G + align2(X) : take(X);
    L.J = take(N * rec + cst > X ? N * rec + cst : X);
    L.Xcur = xcur_late ? L.J : take(X);
    if (!compact) {
        L.A = take(nx * nx);
        L.B = take(nx * nu);
        L.D = take(nx);
        L.X0 = take(nx);
    }
    if (compact) {
        L.ricX = L.G;
        L.ricD = L.J + N * rec + nx * nu + nx + 1; // (RicRec::cS: the spare double of the constant block)
        // kv: the tail of the trajectory's place.  The roll-out writes x_{k+1} at nx (k + 1) .. and has read kv_{k+1} at X - n + nu (k + 1) ..
        // by then: with nx >= nu no state ever lands on a kv that is still to be read (lmpc_fused_ric.hpp, the roll-out).
        L.ricKv = L.G + X - n;
    } else {
        L.ricX = take(kWave); // (directly after A | B | d | x0: together they hold the unconstrained trajectory between the roll-out
                              //  and the first scan -- lmpc_fused_ric.hpp, StageRows::xu)
        L.ricD = L.ricX + kWave - 2;
        L.ricKv = take(n);
    }
    L.BldPhi = L.BldXi = L.J; // (unused by the body)
    const int vec0 = o;
    L.ricS = vec0;
    int sys_end = vec0 + scratch;
    if (compact) {
        L.A = sys_end, sys_end += align2(nx * nx);
        L.B = sys_end, sys_end += align2(nx * nu);
        L.D = sys_end, sys_end += align2(nx);
        L.X0 = sys_end, sys_end += align2(nx);
    }
    L.xs = take(n);
    L.dv = L.zv = L.coef = L.xs;
    L.ap = take(n);
    L.cvec = L.ap;
    L.nb = take(compact ? (mgen > kWave ? mgen - kWave : 1) : (mgen > 0 ? mgen : 1)); // (compact: the first 64 norms live in registers)
    L.eqsgn = take(meq > 0 ? meq : 1);
    L.scal = take(2);
    L.act = take((mtotal + 7) / 8 + 1);
    if (q1regs == 0) { // a column of Q1, its column of R, its multiplier and its row index per active constraint
        auto need = [&](int r) { return r * kWave + align2(r * (r + 1) / 2 + 2) + align2(r + 2) + align2((r + 2) / 2 + 1); };
        int rcap = 0;
        while (rcap < n && o + need(rcap + 1) <= budget) ++rcap;
        if (rcap < 1) return false;
        L.rcap = rcap;
        L.Q1 = take(rcap * kWave);
    }
    L.uv = take(L.rcap + 2);
    L.iact = take((L.rcap + 2) / 2 + 1);
    L.R = take(L.rcap * (L.rcap + 1) / 2 + 2);
    if (o < sys_end) o = sys_end;
    L.BldY = L.BldWe = L.BldCp = L.BldFull = vec0;
    L.total = o;
    return o <= budget;
}

// Stage-cost tables of the Riccati-factor tier (FusedPlan::ric_tab): what lane `l` of lmpc_fused_ric_body owns in the sweep
// is entry (a, b) of  M = Hin + [A B]' P+ [A B]  over z = (x, u) -- x-x upper triangle | u-x | u-u upper triangle | the affine
// column b == nz -- and entry (l % nx, l / nx) of the terminal [HN | hN].  Hin = sum_t [M_t N_t]' W_t [M_t N_t] (+ 1e-6 I on
// u: LMPC.cpp:228-229) does not depend on the instance; the affine entries are linear in the references p_t, which may be
// per-instance (copra_batch_set_cost_reference): the tables hold their coefficients.
inline void build_lane_tables(HostPlan& hp);
inline int build_ric_tables(HostPlan& hp, int rp)
{
    FusedPlan& P = hp.plan;
    const int nx = P.nx, nu = P.nu, nz = nx + nu;
    const int nxx = nx * (nx + 1) / 2, nux = nu * nx, nuu = nu * (nu + 1) / 2;
    std::vector<double> tab((size_t)kWave * (2 + kRicMaxCosts * rp + 3), 0.0);
    std::vector<double> Hin((size_t)nz * nz, 0.0); // the quadratic part, for the second view below
    auto coef = [&](const CostTerm& ct, int r, int a) -> double { // entry (r, a) of [M_t N_t]
        if (a < nx) return (ct.offM >= 0 && ct.kind != kCostControl) ? hp.params[(size_t)ct.offM + r + ct.rows * a] : 0.0;
        return (ct.offN >= 0 && (ct.kind == kCostControl || ct.kind == kCostMixed)) ? hp.params[(size_t)ct.offN + r + ct.rows * (a - nx)] : 0.0;
    };
    for (int lane = 0; lane < kWave; ++lane) {
        int ma = 0, mb = 0;
        bool on = true;
        auto tri = [](int t, int& lo, int& hi) {
            hi = 0;
            while ((hi + 1) * (hi + 2) / 2 <= t) ++hi;
            lo = t - hi * (hi + 1) / 2;
        };
        if (lane < nxx) {
            tri(lane, ma, mb);
        } else if (lane < nxx + nux) {
            ma = nx + (lane - nxx) % nu;
            mb = (lane - nxx) / nu;
        } else if (lane < nxx + nux + nuu) {
            tri(lane - nxx - nux, ma, mb);
            ma += nx;
            mb += nx;
        } else if (lane < nxx + nux + nuu + nz) {
            ma = lane - nxx - nux - nuu;
            mb = nz;
        } else {
            on = false;
        }
        const int ti = lane % nx, tj = lane / nx; // terminal: tj < nx entry of HN, tj == nx entry ti of hN
        double h0 = 0.0, t0 = 0.0;
        for (int t = 0; t < P.ncost; ++t) {
            const CostTerm& ct = P.cost[t];
            const bool in_stage = ct.kind != kCostTarget; // TargetCost: the last state only (costFunctions.cpp:107-120)
            const bool in_term = ct.kind == kCostTrajectory || ct.kind == kCostTarget; // MixedCost stops at x_{N-1} (:207)
            for (int r = 0; r < ct.rows && r < rp; ++r) {
                const double w = hp.params[(size_t)ct.offW + r];
                if (on && in_stage) {
                    if (mb < nz)
                        h0 += (coef(ct, r, ma) * w) * coef(ct, r, mb);
                    else
                        tab[(size_t)kWave * (2 + t * rp + r) + lane] = -(coef(ct, r, ma) * w);
                }
                if (in_term && tj <= nx) {
                    if (tj < nx)
                        t0 += (coef(ct, r, ti) * w) * coef(ct, r, tj);
                    else
                        tab[(size_t)kWave * (2 + t * rp + r) + lane] = -(coef(ct, r, ti) * w); // (lanes nx nx .. nx nx + nx - 1: never an affine stage lane)
                }
            }
        }
        if (on && ma == mb && ma >= nx) {
            double one = 1.0;
            one *= 1e-6; // Q_.setIdentity(); Q_ *= 1e-6;  (LMPC.cpp:228-229)
            h0 += one;
        }
        tab[lane] = h0;
        tab[kWave + lane] = t0;
        if (on && mb < nz) Hin[(size_t)ma + nz * mb] = Hin[(size_t)mb + nz * ma] = h0;
    }
    // second view of Hin for the MFMA sweep: accumulator layout -- lane 16 q + 4 b + r of row block I holds the entry (stacked row
    // 4 I + q, stacked column 4 b + r), stacked index: 0 .. nu-1 = u | 4 .. 4+nx-1 = x (the affine column comes from the lanes above)
    auto unstack = [&](int s) { return s < nu ? nx + s : (s >= 4 && s < 4 + nx) ? s - 4 : -1; }; // -> index in z = (x, u)
    for (int I = 0; I < 3; ++I)
        for (int lane = 0; lane < kWave; ++lane) {
            const int a = unstack(4 * I + (lane >> 4)), b = unstack(4 * ((lane >> 2) & 3) + (lane & 3));
            tab[(size_t)kWave * (2 + kRicMaxCosts * rp + I) + lane] = (a >= 0 && b >= 0) ? Hin[(size_t)a + nz * b] : 0.0;
        }
    const int at = (int)hp.params.size();
    hp.params.insert(hp.params.end(), tab.begin(), tab.end());
    build_lane_tables(hp); // (behind the tier's own tables: the pass in front of it, lmpc_lane.hpp)
    return at;
}

// Tables of the one-instance-per-lane pass (lmpc_lane.hpp) for a controller on the Riccati-factor tier: the stage cost as plain dense
// matrices and the constraint rows grouped by step.  Eligible: inequality rows only, every row a per-step entry (state part on x_k,
// control part on u_k of the same step).  Per-instance cost references and right-hand sides are checked when a solve is launched
// (they can be set at any time).  Sets P.lane_tab (-1: not eligible) and P.lane_rps.
inline void build_lane_tables(HostPlan& hp)
{
    FusedPlan& P = hp.plan;
    P.lane_tab = -1;
    P.lane_rps = 0;
    P.lane_tlds = 0;
    P.lane_cref = -1;
    P.lane_axes = 0;
    P.axis_tab = -1;
    P.axis_cref = -1;
    P.axis_rpa = 0;
    P.axis_const = 0;
    const int nx = P.nx, nu = P.nu, nz = nx + nu, N = P.N;
    const bool lane_ok = nx <= 7; // (the one-instance-per-lane pass: the lane's registers; the (instance, axis)-per-lane solver takes chains of up to 3 states per control)
    if (P.meq > 0 || P.initial_state || nu > 3 || nx > 9 || P.denseQ >= 0 || P.rfull > 0 || P.n > kWave) return;
    std::vector<int> per_step((size_t)N + 1, 0);
    for (int i = 0; i < P.mgen; ++i) {
        const int k = hp.row_step[i], ek = hp.row_ekind[i], gk = hp.row_gkind[i];
        if (ek == kEFull || gk == kGFull || k < 0 || k > N || (k == N && gk != kGNone)) return;
        per_step[k] += 1;
    }
    int rps = 0;
    for (int k = 0; k <= N; ++k) rps = per_step[k] > rps ? per_step[k] : rps;
    if (rps > 32) return;
    int oh, oHN, ohN, oRows;
    lane_tab_offsets(nx, nu, oh, oHN, ohN, oRows);
    const int rw = nz + 2; // E | G | f | index
    const int oCref = (oRows + (N + 1) * rps * rw + 1) & ~1, crw = nz + nx; // reference coefficients: [cost][row (6)][nz | nx]
    std::vector<double> tab((size_t)oCref + (size_t)kRicMaxCosts * 6 * crw, 0.0);
    auto coef = [&](const CostTerm& ct, int r, int a) -> double { // entry (r, a) of [M_t N_t]  (as build_ric_tables)
        if (a < nx) return (ct.offM >= 0 && ct.kind != kCostControl) ? hp.params[(size_t)ct.offM + r + ct.rows * a] : 0.0;
        return (ct.offN >= 0 && (ct.kind == kCostControl || ct.kind == kCostMixed)) ? hp.params[(size_t)ct.offN + r + ct.rows * (a - nx)] : 0.0;
    };
    for (int t = 0; t < P.ncost; ++t) {
        const CostTerm& ct = P.cost[t];
        if (ct.full) return;
        const bool in_stage = ct.kind != kCostTarget; // TargetCost: the last state only (costFunctions.cpp:107-120)
        const bool in_term = ct.kind == kCostTrajectory || ct.kind == kCostTarget; // MixedCost stops at x_{N-1} (:207)
        for (int r = 0; r < ct.rows; ++r) {
            const double w = hp.params[(size_t)ct.offW + r], pr = hp.params[(size_t)ct.offP + r];
            for (int a = 0; a < nz; ++a) {
                if (in_stage) {
                    for (int b = 0; b < nz; ++b) tab[(size_t)a + nz * b] += (coef(ct, r, a) * w) * coef(ct, r, b);
                    tab[(size_t)oh + a] += -(coef(ct, r, a) * w) * pr;
                    if (t < kRicMaxCosts && r < 6) tab[(size_t)oCref + ((size_t)t * 6 + r) * crw + a] = -(coef(ct, r, a) * w);
                }
                if (in_term && a < nx) {
                    for (int b = 0; b < nx; ++b) tab[(size_t)oHN + a + nx * b] += (coef(ct, r, a) * w) * coef(ct, r, b);
                    tab[(size_t)ohN + a] += -(coef(ct, r, a) * w) * pr;
                    if (t < kRicMaxCosts && r < 6) tab[(size_t)oCref + ((size_t)t * 6 + r) * crw + nz + a] = -(coef(ct, r, a) * w);
                }
            }
        }
    }
    for (int c = nx; c < nz; ++c) {
        double one = 1.0;
        one *= 1e-6; // Q_.setIdentity(); Q_ *= 1e-6;  (LMPC.cpp:228-229)
        tab[(size_t)c + nz * c] += one;
    }
    std::vector<int> filled((size_t)N + 1, 0);
    for (int k = 0; k <= N; ++k)
        for (int r = 0; r < rps; ++r) {
            tab[(size_t)oRows + ((size_t)k * rps + r) * rw + nz] = HUGE_VAL;
            tab[(size_t)oRows + ((size_t)k * rps + r) * rw + nz + 1] = -1.0;
        }
    for (int i = 0; i < P.mgen; ++i) {
        const int k = hp.row_step[i];
        double* row = tab.data() + oRows + ((size_t)k * rps + filled[k]++) * rw;
        if (e_onehot(hp.row_ekind[i])) row[hp.row_eoff[i]] = e_sign(hp.row_ekind[i]);
        if (hp.row_ekind[i] == kEDense)
            for (int c = 0; c < nx; ++c) row[c] = hp.params[(size_t)hp.row_eoff[i] + c];
        if (hp.row_gkind[i] == kGStep)
            for (int c = 0; c < nu; ++c) row[nx + c] = hp.params[(size_t)hp.row_goff[i] + c];
        row[nz] = hp.row_f[i];
        row[nz + 1] = (double)i;
    }
    // axis-decoupled costs and rows (FusedPlan::lane_axes for order 0): H(a, b) = 0 and HN(a, b) = 0 wherever a and b belong to different axes ...
    auto decoupled = [&](int ord) {
        const int nxa_ = nx / nu;
        auto axis = [&](int a) { return a < nx ? (ord ? a / nxa_ : a % nu) : a - nx; };
        bool ok = nu > 1 && nx % nu == 0 && !hp.opt.no_lane_axes;
        for (int a = 0; a < nz && ok; ++a)
            for (int b = 0; b < nz && ok; ++b) {
                if (axis(a) == axis(b)) continue;
                if (tab[(size_t)a + nz * b] != 0.0) ok = false;
                if (a < nx && b < nx && tab[(size_t)oHN + a + nx * b] != 0.0) ok = false;
            }
        // ... and no constraint row does: the pass takes its speculative steps axis by axis there (lmpc_lane.hpp: lane_spec_axes) -- right as long
        // as the iterates BETWEEN the axes' steps, which it never forms, cannot violate anything, i.e. as long as every row looks at one axis
        // (a row u_x + u_y <= 1 can hold before and after both steps and fail in between: qpgen2 would add it there)
        for (int k = 0; k <= N && ok; ++k)
            for (int r = 0; r < rps && ok; ++r) {
                const double* row = tab.data() + oRows + ((size_t)k * rps + r) * rw;
                int ax = -1;
                for (int a = 0; a < nz; ++a)
                    if (row[a] != 0.0) {
                        if (ax >= 0 && ax != axis(a)) ok = false;
                        ax = axis(a);
                    }
            }
        return ok;
    };
    const bool axes_decoupled = decoupled(0);
    P.lane_axes = axes_decoupled ? 1 : 0;
    const bool cref_ok = P.ncost <= kRicMaxCosts && P.rmax <= 6;
    if (lane_ok) {
        if (hp.params.size() & 1) hp.params.push_back(0.0);
        P.lane_tab = (int)hp.params.size();
        P.lane_rps = rps;
        P.lane_cref = cref_ok ? oCref : -1; // (-1: per-instance references keep the first tier alone)
        {
            int oHl = 0;
            const int base = lane_lds_doubles(nx, nu, oHl), tl = (N + 1) * rps * rw + 2 * P.n;
            P.lane_tlds = ((size_t)(base + tl) * sizeof(double) <= 40u * 1024u) ? tl : 0; // (four waves per CU)
        }
        hp.params.insert(hp.params.end(), tab.begin(), tab.end());
    } else {
        P.lane_axes = 0; // (what the pass reads; the solver below checks the same on its own)
    }
    // Tables of the one-(instance, axis)-per-lane solver (lmpc_axis.hpp): the same stage cost and rows, cut up axis by axis.  Eligible: the
    // costs and every row look at one axis each (lane_axes above; a system with ONE control is one axis), at most kAxisMaxRpa rows per axis
    // and step.  (Whether the SYSTEMS couple two axes is checked per instance by the kernel.)
    P.axis_tab = -1;
    P.axis_cref = -1;
    P.axis_rpa = 0;
    P.axis_order = 0;
    hp.axis1_tab = hp.axis1_cref = -1;
    hp.axis1_rpa = hp.axis1_const = 0;
    // (two sets: for state i on axis i % nu -- the plan's own fields -- and for state i on axis i / nxa -- HostPlan::axis1_*; which one a
    //  controller's systems are in is seen when they are set, axis_order_of above)
    for (int ord = 0; ord < 2; ++ord) {
        int set_tab = -1, set_cref = -1, set_rpa = 0, set_const = 0;
        if (!hp.opt.no_axis_solver && nx % nu == 0 && nx / nu <= 3 && nu > 1 && (ord == 0 ? axes_decoupled : (nx / nu > 1 && decoupled(1)))) {
        const int nxa = nx / nu, nza = nxa + 1, arw = nxa + 3;
        auto axis = [&](int a) { return a < nx ? (ord ? a / nxa : a % nu) : a - nx; };
        // rows per axis and step
        int rpa = 0;
        bool ok = true;
        std::vector<int> row_axis((size_t)(N + 1) * rps, -1);
        for (int k = 0; k <= N; ++k) {
            std::vector<int> cnt((size_t)nu, 0);
            for (int r = 0; r < rps; ++r) {
                const double* row = tab.data() + oRows + ((size_t)k * rps + r) * rw;
                if (row[nz + 1] < 0.0) continue; // (a row that is not there)
                int ax = 0;
                for (int a = 0; a < nz; ++a)
                    if (row[a] != 0.0) ax = axis(a);
                row_axis[(size_t)k * rps + r] = ax;
                cnt[(size_t)ax] += 1;
                if (cnt[(size_t)ax] > rpa) rpa = cnt[(size_t)ax];
            }
        }
        if (rpa > kAxisMaxRpa) ok = false;
        if (rpa < 1) rpa = 1;
        if (ok) {
            int aoh, aoHN, aohN, aoRows;
            axis_tab_offsets(nxa, aoh, aoHN, aohN, aoRows);
            const int TA = axis_tab_doubles(nxa, N, rpa);
            std::vector<double> at((size_t)nu * TA, 0.0);
            for (int c = 0; c < nu; ++c) {
                double* t = at.data() + (size_t)c * TA;
                auto zi = [&](int a) { return a < nxa ? (ord ? c * nxa + a : c + nu * a) : nx + c; }; // axis index -> index in z = (x, u) of the system
                for (int a = 0; a < nza; ++a) {
                    for (int b = 0; b < nza; ++b) t[a + nza * b] = tab[(size_t)zi(a) + nz * zi(b)];
                    t[aoh + a] = tab[(size_t)oh + zi(a)];
                }
                for (int a = 0; a < nxa; ++a) {
                    for (int b = 0; b < nxa; ++b) t[aoHN + a + nxa * b] = tab[(size_t)oHN + zi(a) + nx * zi(b)];
                    t[aohN + a] = tab[(size_t)ohN + zi(a)];
                }
                for (int k = 0; k <= N; ++k) {
                    for (int j = 0; j < rpa; ++j) {
                        t[aoRows + ((size_t)k * rpa + j) * arw + nxa + 1] = HUGE_VAL;
                        t[aoRows + ((size_t)k * rpa + j) * arw + nxa + 2] = -1.0;
                    }
                    int filled_c = 0;
                    for (int r = 0; r < rps; ++r) {
                        if (row_axis[(size_t)k * rps + r] != c) continue;
                        const double* row = tab.data() + oRows + ((size_t)k * rps + r) * rw;
                        double* dst = t + aoRows + ((size_t)k * rpa + filled_c++) * arw;
                        for (int a = 0; a < nxa; ++a) dst[a] = row[zi(a)];
                        dst[nxa] = row[nx + c];
                        dst[nxa + 1] = row[nz];
                        dst[nxa + 2] = row[nz + 1];
                    }
                }
            }
            // the same tables at every step?  (FusedPlan::axis_const)
            bool cst = true;
            for (int c = 0; c < nu && cst; ++c) {
                const double* t = at.data() + (size_t)c * TA;
                for (int j = 0; j < rpa && cst; ++j) {
                    const double* r0 = t + aoRows + (size_t)j * arw;
                    const double* r1 = t + aoRows + ((size_t)rpa + j) * arw;
                    const double stride = N >= 1 ? r1[nxa + 2] - r0[nxa + 2] : 0.0;
                    for (int k = 0; k <= N && cst; ++k) {
                        const double* rk = t + aoRows + ((size_t)k * rpa + j) * arw;
                        for (int a = 0; a < nxa; ++a) cst = cst && rk[a] == r0[a];
                        cst = cst && rk[nxa] == 0.0 && rk[nxa + 1] == r0[nxa + 1];
                        cst = cst && ((r0[nxa + 2] < 0.0 && rk[nxa + 2] < 0.0) || (r0[nxa + 2] >= 0.0 && rk[nxa + 2] == r0[nxa + 2] + k * stride));
                    }
                }
                for (int k = 0; k < N && cst; ++k)
                    cst = cst && hp.ub[(size_t)k * nu + c] == hp.ub[(size_t)c] && hp.lb[(size_t)k * nu + c] == hp.lb[(size_t)c];
            }
            set_const = cst ? 1 : 0;
            if (hp.params.size() & 1) hp.params.push_back(0.0);
            set_tab = (int)hp.params.size();
            set_rpa = rpa;
            hp.params.insert(hp.params.end(), at.begin(), at.end());
            // the coefficients of the cost references (oCref above), axis by axis: what a lane whose instance has its OWN references
            // (copra_batch_set_cost_reference) rebuilds h and hN of its axis from
            { // (straight from the costs: any number of rows per cost -- the jerk-controlled CoM model's TrajectoryCost has nine)
                const int ew = 5 + nza + nxa, aw = 1 + kAxisMaxRef * ew; // [cost | row | length of its reference | stride per step | where the controller-wide one sits | coefficients]
                std::vector<double> ac((size_t)nu * aw, 0.0);
                bool fits = P.ncost <= kMaxCosts;
                for (int c = 0; c < nu && fits; ++c) {
                    auto zi = [&](int a) { return a < nxa ? (ord ? c * nxa + a : c + nu * a) : nx + c; };
                    int nref = 0;
                    for (int t = 0; t < P.ncost && fits; ++t) {
                        const CostTerm& ct = P.cost[t];
                        const bool in_stage = ct.kind != kCostTarget, in_term = ct.kind == kCostTrajectory || ct.kind == kCostTarget;
                        for (int r = 0; r < ct.rows; ++r) {
                            const double w = hp.params[(size_t)ct.offW + r];
                            double ch[4] = { 0, 0, 0, 0 }, cn[3] = { 0, 0, 0 };
                            bool any = false;
                            for (int a = 0; a < nza; ++a) {
                                ch[a] = in_stage ? -(coef(ct, r, zi(a)) * w) : 0.0;
                                any = any || ch[a] != 0.0;
                            }
                            for (int a = 0; a < nxa; ++a) {
                                cn[a] = in_term ? -(coef(ct, r, zi(a)) * w) : 0.0;
                                any = any || cn[a] != 0.0;
                            }
                            if (!any) continue;
                            if (nref == kAxisMaxRef) {
                                fits = false;
                                break;
                            }
                            double* dst = ac.data() + (size_t)c * aw + 1 + (size_t)nref * ew;
                            dst[0] = (double)t;
                            dst[1] = (double)r;
                            dst[2] = (double)ct.prows;
                            dst[3] = (double)ct.pstride;
                            dst[4] = (double)ct.offP;
                            for (int a = 0; a < nza; ++a) dst[5 + a] = ch[a];
                            for (int a = 0; a < nxa; ++a) dst[5 + nza + a] = cn[a];
                            nref += 1;
                        }
                    }
                    ac[(size_t)c * aw] = (double)nref;
                }
                if (fits) {
                    set_cref = (int)hp.params.size();
                    hp.params.insert(hp.params.end(), ac.begin(), ac.end());
                    if (hp.params.size() & 1) hp.params.push_back(0.0);
                }
            }
        }
        }
        if (ord == 0)
            P.axis_tab = set_tab, P.axis_cref = set_cref, P.axis_rpa = set_rpa, P.axis_const = set_const;
        else
            hp.axis1_tab = set_tab, hp.axis1_cref = set_cref, hp.axis1_rpa = set_rpa, hp.axis1_const = set_const;
    }
}

// Factor-only layouts trade columns of Q1 for instances per CU.  The next layout down the ladder from `cur`: one
// instance per CU fewer (at least four) and more room for active constraints; false when there is none.
inline bool next_tri_layout(const FusedPlan& P, const LdsLayout& cur, LdsLayout& out, bool no_ladder = false)
{
    if (!cur.tri) return false;
    if (no_ladder) return false; // (copra_options_t::no_ladder -- tests: the fall-back that follows an exhausted ladder, reachable at once)
    const int rp = specialised_cost_rows(P.nx, P.nu, P.N, P.rmax, P.rfull);
    const int rows = rp > P.rmax ? rp : P.rmax;
    const int kcur = (160 * 1024) / (cur.total * (int)sizeof(double));
    if (cur.ric) { // the Riccati-factor tier keeps its factor
        // Q1 moves to LDS, one instance per CU fewer per step.  (A first step with TEN register columns on a 256-VGPR build of the kernel,
        //  eight instances per CU, was measured in round 3 and dropped: 1.39 ms against 1.30 ms for seven LDS columns at nine per CU on a
        //  mid-constrained workload, no better anywhere: profiles/r03/README.md)
        for (int k = kcur - 1; k >= 4; --k) {
            const int budget = ((160 * 1024 / k) & ~511) / (int)sizeof(double);
            LdsLayout t {};
            if (layout_lds_ric(t, P.nx, P.nu, P.N, P.n, P.X, P.mgen, P.meq, P.mtotal, cur.ricC != 0, 0, budget) && t.rcap > cur.rcap) {
                out = t;
                return true;
            }
        }
        return false;
    }
    for (int k = kcur - 1; k >= 4; --k) {
        const int budget = ((160 * 1024 / k) & ~511) / (int)sizeof(double);
        LdsLayout t {};
        if (layout_lds(t, P.nx, P.nu, P.N, P.n, P.X, rows, P.mgen, P.meq, P.mtotal, true, true, budget, P.rfull, true, P.rows_direct != 0)
            && t.total <= budget && t.rcap > cur.rcap) {
            out = t;
            return true;
        }
    }
    return false;
}

// The layout with Q1 in LDS that is closest to a register-Q1 layout (kernels without the QR instantiation -- the
// shared-model path -- cannot run the latter)
inline bool tri_layout_with_lds_q1(const FusedPlan& P, const LdsLayout& cur, LdsLayout& out)
{
    if (!cur.tri || (cur.q1regs == 0 && !cur.ric)) return false; // (a Riccati-factor layout is never run by those kernels)
    const int rp = specialised_cost_rows(P.nx, P.nu, P.N, P.rmax, P.rfull);
    const int rows = rp > P.rmax ? rp : P.rmax;
    const int kfirst = (160 * 1024) / (cur.total * (int)sizeof(double)) < 8 ? (160 * 1024) / (cur.total * (int)sizeof(double)) : 8;
    for (int k = kfirst; k >= 4; --k) {
        const int budget = ((160 * 1024 / k) & ~511) / (int)sizeof(double);
        LdsLayout t {};
        if (layout_lds(t, P.nx, P.nu, P.N, P.n, P.X, rows, P.mgen, P.meq, P.mtotal, true, true, budget, 0, true, P.rows_direct != 0)
            && t.total <= budget && (t.rcap >= cur.rcap || k == 4)) {
            out = t;
            return true;
        }
    }
    return false;
}

inline bool is_neg_inf(double v) { return std::isinf(v) && v < 0; }
inline bool is_pos_inf(double v) { return std::isinf(v) && v > 0; }

// Move a controller that build_plan has laid out onto the Riccati-factor tier (lmpc_fused_ric.hpp) if its shape and its costs allow:
// the layout with the stage records in the factor's place, Q1 in registers, and the stage-cost tables (appended to hp.params: the
// caller uploads them again).  What build_plan does itself for the shapes the library instantiates; copra_batch_specialise calls it
// for every other shape once the kernel for it is compiled.
inline bool take_ric_layout(HostPlan& hp)
{
    FusedPlan& P = hp.plan;
    const int nx = P.nx, nu = P.nu, N = P.N;
    if (P.lds.ric) return true;
    if (hp.large || P.initial_state || !ric_shape_ok(nx, nu, N)) return false;
    if (P.rmax > 6 || P.rfull != 0 || P.denseQ >= 0 || P.ncost > kRicMaxCosts) return false;
    for (int t = 0; t < P.ncost; ++t)
        if (P.cost[t].full) return false;
    // (copra_options_t::ric_k = instances per CU: start on the LDS-Q1 step of the ladder with that budget -- tests pin a ladder level with it)
    const int rk = hp.opt.ric_k > 0 ? hp.opt.ric_k : 0;
    for (int k = rk ? rk : 16; k >= 6; --k) { // (small shapes: as many instances per CU as the LDS granule allows)
        const int budget = ((160 * 1024 / k) & ~511) / (int)sizeof(double);
        LdsLayout t {};
        if (layout_lds_ric(t, nx, nu, N, P.n, P.X, P.mgen, P.meq, P.mtotal, P.rows_pure != 0 && !hp.opt.ric_general, rk ? 0 : kFusedQ1Regs, budget)) {
            hp.lds_safe = P.lds; // (what the controller falls back to when the tier's layout ladder is exhausted: adapt_layout)
            hp.safe_two_tier = hp.two_tier;
            hp.two_tier = true;
            hp.dense = true;
            P.lds = t;
            P.ric_tab = build_ric_tables(hp, 6);
            hp.lds_bytes = (size_t)P.lds.total * sizeof(double);
            return true;
        }
    }
    return false;
}

inline copra_status_t build_plan(HostPlan& hp, const copra_dims_t& dims, int n_costs, const copra_cost_desc_t* costs,
    int n_cstrs, const copra_cstr_desc_t* cstrs, const copra_initial_state_desc_t* is = nullptr)
{
    FusedPlan& P = hp.plan;
    const int nx = dims.nx, nu = dims.nu, N = dims.N;
    if (nx <= 0 || nu <= 0 || N <= 0 || dims.batch < 0) { // PreviewSystem.cpp:19-33
        hp.error = "PreviewSystem: dimensions and number of steps must be positive";
        return COPRA_ERR_DOMAIN;
    }
    const int X = nx * (N + 1), U = nu * N;
    P.nx = nx;
    P.nu = nu;
    P.N = N;
    P.n = U;
    P.X = X;
    P.batch = dims.batch;
    P.dump_instance = -1;
    P.initial_state = is ? 1 : 0;
    if (is) {
        if (!is->R || !is->r) return hp.error = "InitialStateLMPC: R / r missing", COPRA_ERR_DOMAIN;
        hp.isR.assign(is->R, is->R + (size_t)nx * nx);
        hp.isr.assign(is->r, is->r + nx);
    }
    auto push = [&](const double* src, int count) {
        int at = (int)hp.params.size();
        hp.params.insert(hp.params.end(), src, src + count);
        if (hp.params.size() & 1) hp.params.push_back(0.0);
        return at;
    };

    // ---------------- costs ----------------
    // COPRA_COST_DENSE terms (host-evaluated user cost functions) are summed into one dense block; the others become
    // cost terms the kernels evaluate
    std::vector<double> dQ, dc, dE, df;
    std::vector<copra_cost_desc_t> builtin;
    for (int k = 0; k < n_costs; ++k) {
        const copra_cost_desc_t& c = costs[k];
        hp.cost_slot.push_back(c.kind != COPRA_COST_DENSE ? (int)builtin.size() : -1);
        if (c.kind != COPRA_COST_DENSE) {
            builtin.push_back(c);
            continue;
        }
        if (!c.Q || (!is && !c.c) || (is && (!c.E || !c.f))) // LMPC.cpp:252-255 / InitialStateLMPC.cpp:80-84
            return hp.error = "dense cost: Q and c (LMPC) or Q, E and f (InitialStateLMPC) are needed", COPRA_ERR_DOMAIN;
        if (dQ.empty()) dQ.assign((size_t)U * U, 0.0), dc.assign((size_t)U, 0.0), dE.assign((size_t)nx * U, 0.0), df.assign((size_t)U, 0.0);
        for (size_t e = 0; e < (size_t)U * U; ++e) dQ[e] += c.Q[e];
        if (c.c)
            for (int e = 0; e < U; ++e) dc[(size_t)e] += c.c[e];
        if (c.E)
            for (size_t e = 0; e < (size_t)nx * U; ++e) dE[e] += c.E[e];
        if (c.f)
            for (int e = 0; e < U; ++e) df[(size_t)e] += c.f[e];
    }
    P.denseQ = P.densec = P.denseE = P.densef = -1;
    P.ric_tab = -1;
    P.lane_tab = -1;
    P.axis_tab = -1;
    P.axis_rpa = 0;
    P.axis_const = 0;
    if (!dQ.empty()) {
        P.denseQ = push(dQ.data(), U * U);
        P.densec = push(dc.data(), U);
        P.denseE = push(dE.data(), nx * U);
        P.densef = push(df.data(), U);
    }
    n_costs = (int)builtin.size();
    costs = builtin.data();
    if (n_costs > kMaxCosts) {
        hp.error = "too many cost functions for the fused kernel";
        return COPRA_ERR_UNSUPPORTED;
    }
    P.ncost = n_costs;
    for (int k = 0; k < kMaxCosts; ++k) P.cost_p[k] = nullptr, P.model_ref_off[k] = -1;
    P.rmax = 1;
    P.rfull = 0;
    P.stage_refs = 0;
    for (int k = 0; k < n_costs; ++k) {
        const copra_cost_desc_t& c = costs[k];
        CostTerm& t = P.cost[k];
        t.kind = c.kind;
        t.rows = c.rows;
        t.offM = t.offN = -1;
        t.offMask = -1;
        if (c.rows <= 0 || !c.p || !c.weights) {
            hp.error = "cost: empty p / weights";
            return COPRA_ERR_DOMAIN;
        }
        bool full = false;
        switch (c.kind) {
        case COPRA_COST_TRAJECTORY: // costFunctions.cpp:44-61
            if (!c.M) return hp.error = "TrajectoryCost: M missing", COPRA_ERR_DOMAIN;
            if (c.m_cols == nx)
                full = false;
            else if (c.m_cols == X)
                full = true;
            else
                return hp.error = "TrajectoryCost: M has neither xDim nor fullXDim columns", COPRA_ERR_DOMAIN;
            break;
        case COPRA_COST_TARGET: // costFunctions.cpp:88-98
            if (!c.M) return hp.error = "TargetCost: M missing", COPRA_ERR_DOMAIN;
            if (c.m_cols != nx) return hp.error = "TargetCost: M must have xDim columns", COPRA_ERR_DOMAIN;
            break;
        case COPRA_COST_CONTROL: // costFunctions.cpp:122-137
            if (!c.N) return hp.error = "ControlCost: N missing", COPRA_ERR_DOMAIN;
            if (c.n_cols == nu)
                full = false;
            else if (c.n_cols == U)
                full = true;
            else
                return hp.error = "ControlCost: N has neither uDim nor fullUDim columns", COPRA_ERR_DOMAIN;
            break;
        case COPRA_COST_MIXED: // costFunctions.cpp:173-193
            if (!c.M || !c.N) return hp.error = "MixedCost: M / N missing", COPRA_ERR_DOMAIN;
            if (c.m_cols == nx && c.n_cols == nu)
                full = false;
            else if (c.m_cols == X && c.n_cols == U)
                full = true;
            else
                return hp.error = "MixedCost: M / N column mismatch", COPRA_ERR_DOMAIN;
            break;
        default:
            return hp.error = "unknown cost kind", COPRA_ERR_DOMAIN;
        }
        t.pstride = 0;
        t.prows = c.rows;
        // A REFERENCE TRAJECTORY: the reference's API can only express a reference that changes along the horizon as a full-size entry
        // (costFunctions.cpp:63-82, 139-156): M = blkdiag(M0, .., M0) over the N + 1 states (N = blkdiag(N0, .., N0) over the N
        // controls), the same weights in every block, p the stacked references.  That is a PER-STEP entry whose reference is that of
        // the step -- the same sums over the same blocks in the same order, without the dense contraction of a full-size entry
        // (lmpc_fused.hpp: 7.6 M solves/s at the headline shape).  One-wave LMPC controllers only; the kernels that do not evaluate
        // costs step by step with the step's reference refuse such a controller (FusedPlan::stage_refs).
        if (full && !is && U <= kWave && (c.kind == COPRA_COST_TRAJECTORY || c.kind == COPRA_COST_CONTROL || c.kind == COPRA_COST_MIXED)
            && !hp.opt.no_stage_refs) {
            // (MixedCost, costFunctions.cpp:173-210: M x_k + N u_k - p_k over the N steps with a control -- M has fullXDim columns, the
            //  ones of x_N zero; both matrices must repeat their block)
            const bool traj = c.kind == COPRA_COST_TRAJECTORY, mixed = c.kind == COPRA_COST_MIXED;
            const int S = traj ? N + 1 : N, R = c.rows;
            // `Mx`, column-major R x (cols), is blkdiag(its first block, ...) over S row blocks of r rows and column blocks of cwd (columns
            // past the S-th block: zero)
            auto repeats = [&](const double* Mx, int cols, int cwd, int r) {
                for (int sblk = 0; sblk < S; ++sblk)
                    for (int i = 0; i < r; ++i)
                        for (int j = 0; j < cols; ++j) {
                            const double v = Mx[(size_t)j * R + (size_t)sblk * r + i];
                            const int jb = j / cwd, jc = j - jb * cwd;
                            if (v != (jb == sblk ? Mx[(size_t)jc * R + i] : 0.0)) return false;
                        }
                return true;
            };
            if (R % S == 0 && R / S <= 9) { // (nine: the jerk-controlled CoM model in three dimensions; the Riccati-factor tier and the pass take six, plan checks below)
                const int r = R / S;
                bool ok = true;
                for (int sblk = 0; sblk < S && ok; ++sblk)
                    for (int i = 0; i < r && ok; ++i) ok = c.weights[(size_t)sblk * r + i] == c.weights[i];
                if (ok && (traj || mixed)) ok = repeats(c.M, X, nx, r);
                if (ok && (!traj)) ok = repeats(c.N, U, nu, r);
                if (ok) {
                    auto block = [&](const double* Mx, int cwd) {
                        std::vector<double> blk((size_t)r * cwd);
                        for (int j = 0; j < cwd; ++j)
                            for (int i = 0; i < r; ++i) blk[(size_t)i + (size_t)r * j] = Mx[(size_t)j * R + i];
                        return blk;
                    };
                    t.rows = r;
                    t.full = 0;
                    t.ident = 0;
                    if (traj || mixed) {
                        const std::vector<double> blk = block(c.M, nx);
                        if (traj && r == nx) {
                            bool id = true;
                            for (int j = 0; j < nx && id; ++j)
                                for (int i = 0; i < nx && id; ++i) id = blk[(size_t)j * nx + i] == ((i == j) ? 1.0 : 0.0);
                            t.ident = id ? 1 : 0;
                        }
                        t.offM = push(blk.data(), r * nx);
                    }
                    if (!traj) {
                        const std::vector<double> blk = block(c.N, nu);
                        t.offN = push(blk.data(), r * nu);
                    }
                    if (r > P.rmax) P.rmax = r;
                    t.offP = push(c.p, R);
                    t.offW = push(c.weights, r);
                    t.pstride = r;
                    t.prows = R;
                    P.stage_refs = 1;
                    continue;
                }
            }
        }
        t.full = full ? 1 : 0;
        t.ident = 0;
        if (!full && c.kind != COPRA_COST_CONTROL && c.rows == nx) {
            bool id = true;
            for (int j = 0; j < nx && id; ++j)
                for (int i = 0; i < nx && id; ++i) id = c.M[(size_t)j * nx + i] == ((i == j) ? 1.0 : 0.0);
            t.ident = id ? 1 : 0;
        }
        if (full) {
            // full-size entry: keep M (rows x fullXDim) and N (rows x fullUDim) ROW-major, one contiguous row per cost row
            auto push_rowmajor = [&](const double* Mx, int rows, int cols) {
                std::vector<double> tmp((size_t)rows * cols);
                for (int i = 0; i < rows; ++i)
                    for (int j = 0; j < cols; ++j) tmp[(size_t)i * cols + j] = Mx[(size_t)j * rows + i];
                return push(tmp.data(), rows * cols);
            };
            if (c.kind != COPRA_COST_CONTROL) t.offM = push_rowmajor(c.M, c.rows, X);
            if (c.kind != COPRA_COST_TRAJECTORY) t.offN = push_rowmajor(c.N, c.rows, U);
            if (c.kind != COPRA_COST_CONTROL) { // the K-steps of the dense contraction in which a block of sixteen rows of M is not all zero
                const int KS = (X + 3) / 4, NW = (KS + 63) / 64, RB = (c.rows + 15) / 16;
                std::vector<double> words((size_t)2 * RB * NW, 0.0);
                for (int rb = 0; rb < RB; ++rb)
                    for (int ks = 0; ks < KS; ++ks) {
                        bool nz = false;
                        for (int i = 16 * rb; i < 16 * rb + 16 && i < c.rows && !nz; ++i)
                            for (int j = 4 * ks; j < 4 * ks + 4 && j < X && !nz; ++j) nz = c.M[(size_t)j * c.rows + i] != 0.0;
                        if (nz) {
                            const size_t at = (size_t)2 * (rb * NW + ks / 64) + ((ks % 64) >= 32 ? 1 : 0);
                            words[at] += std::ldexp(1.0, ks % 32); // (bit ks % 32 of that half: an exactly representable integer below 2^32)
                        }
                    }
                t.offMask = push(words.data(), (int)words.size());
            }
            if (c.rows > P.rfull) P.rfull = c.rows;
        } else {
            if (c.kind != COPRA_COST_CONTROL) t.offM = push(c.M, c.rows * nx);
            if (c.kind == COPRA_COST_CONTROL || c.kind == COPRA_COST_MIXED) t.offN = push(c.N, c.rows * nu);
            if (c.rows > P.rmax) P.rmax = c.rows;
        }
        t.offP = push(c.p, c.rows);
        t.offW = push(c.weights, c.rows);
    }

    // ---------------- constraints: two passes (equalities first, then inequalities) ----------------
    hp.lb.assign(U, -DBL_MAX); // LMPC.cpp:207-208
    hp.ub.assign(U, DBL_MAX);
    int bound_line = 0;
    P.any_state_rows = 0;
    P.rows_direct = 1;
    P.rows_pure = 1;
    // validate + bounds
    for (int k = 0; k < n_cstrs; ++k) {
        const copra_cstr_desc_t& c = cstrs[k];
        if (c.rows <= 0) return hp.error = "constraint: no rows", COPRA_ERR_DOMAIN;
        switch (c.kind) {
        case COPRA_CSTR_TRAJECTORY:
            if (!c.E || !c.f) return hp.error = "TrajectoryConstraint: E / f missing", COPRA_ERR_DOMAIN;
            if (c.e_cols != nx && c.e_cols != X)
                return hp.error = "TrajectoryConstraint: E has neither xDim nor fullXDim columns", COPRA_ERR_DOMAIN;
            break;
        case COPRA_CSTR_CONTROL:
            if (!c.G || !c.f) return hp.error = "ControlConstraint: G / f missing", COPRA_ERR_DOMAIN;
            if (c.g_cols != nu && c.g_cols != U)
                return hp.error = "ControlConstraint: G has neither uDim nor fullUDim columns", COPRA_ERR_DOMAIN;
            break;
        case COPRA_CSTR_MIXED:
            if (!c.E || !c.G || !c.f) return hp.error = "MixedConstraint: E / G / f missing", COPRA_ERR_DOMAIN;
            if (!((c.e_cols == nx && c.g_cols == nu) || (c.e_cols == X && c.g_cols == U)))
                return hp.error = "MixedConstraint: E / G column mismatch", COPRA_ERR_DOMAIN;
            break;
        case COPRA_CSTR_TRAJECTORY_BOUND:
            if (!c.lower || !c.upper) return hp.error = "TrajectoryBoundConstraint: bounds missing", COPRA_ERR_DOMAIN;
            if (c.rows != nx && c.rows != X)
                return hp.error = "TrajectoryBoundConstraint: bounds have neither xDim nor fullXDim rows",
                       COPRA_ERR_DOMAIN;
            break;
        case COPRA_CSTR_DENSE:
            if (!c.A || (!is && !c.b) || (is && (!c.Y || !c.z)))
                return hp.error = "dense constraint: A and b (LMPC) or Y, A and z (InitialStateLMPC) are needed", COPRA_ERR_DOMAIN;
            break;
        case COPRA_CSTR_CONTROL_BOUND: {
            if (!c.lower || !c.upper) return hp.error = "ControlBoundConstraint: bounds missing", COPRA_ERR_DOMAIN;
            if (c.rows != nu && c.rows != U)
                return hp.error = "ControlBoundConstraint: bounds have neither uDim nor fullUDim rows",
                       COPRA_ERR_DOMAIN;
            if (bound_line + U > U) // LMPC.cpp:274-279 writes consecutively: a second bound constraint overflows
                return hp.error = "more than one ControlBoundConstraint does not fit lb/ub", COPRA_ERR_RUNTIME;
            for (int i = 0; i < U; ++i) { // constraints.cpp:359-367
                const int src = (c.rows == nu) ? (i % nu) : i;
                hp.lb[bound_line + i] = c.lower[src];
                hp.ub[bound_line + i] = c.upper[src];
            }
            bound_line += U;
            break;
        }
        default:
            return hp.error = "unknown constraint kind", COPRA_ERR_DOMAIN;
        }
    }
    auto add_row = [&](int step, int ekind, int eoff, int gkind, int goff, double f) {
        hp.row_step.push_back(step);
        hp.row_ekind.push_back(ekind);
        hp.row_eoff.push_back(eoff);
        hp.row_gkind.push_back(gkind);
        hp.row_goff.push_back(goff);
        hp.row_f.push_back(f);
        if (ekind != kENone) P.any_state_rows = 1;
        if (ekind != kENone && !e_onehot(ekind)) P.rows_direct = 0;
        if (ekind != kENone && (!e_onehot(ekind) || gkind != kGNone)) P.rows_pure = 0;
    };
    // row-major copy of one row of a column-major (rows x cols) matrix into the blob
    auto push_row = [&](const double* Mx, int rows, int cols, int r) {
        std::vector<double> tmp((size_t)cols);
        for (int j = 0; j < cols; ++j) tmp[(size_t)j] = Mx[(size_t)j * rows + r];
        return push(tmp.data(), cols);
    };
    // A row of a FULL-SIZE entry (E over the whole trajectory, G over all controls: constraints.cpp:66-84, 137-148, 197-226) whose
    // non-zeros lie inside ONE step -- a terminal constraint written as a full-size matrix, the usual way to get one from the reference --
    // is a per-step row of that step: the same coefficients, the same arithmetic over them, none of the full-row machinery (sums over
    // the whole trajectory at every slack evaluation, the general variants of the tiers, no lane pass).
    const bool step_rows = !hp.opt.no_step_rows;
    auto single_block = [&](const double* Mx, int rows, int blk, int nblk, int i, int& at) { // false: more than one block of row i is non-zero
        at = -1;
        for (int b = 0; b < nblk; ++b)
            for (int j = 0; j < blk; ++j)
                if (Mx[(size_t)(b * blk + j) * rows + i] != 0.0) {
                    if (at >= 0 && at != b) return false;
                    at = b;
                }
        return true;
    };
    auto push_block = [&](const double* Mx, int rows, int blk, int b, int i) {
        std::vector<double> tmp((size_t)blk);
        for (int j = 0; j < blk; ++j) tmp[(size_t)j] = Mx[(size_t)(b * blk + j) * rows + i];
        return push(tmp.data(), blk);
    };
    // state part of a per-step row from nx coefficients: +- one component (selection row) or dense
    auto state_kind = [&](const double* coef, bool ineq, int& eoff_or_comp) {
        int nnz = 0, at = -1;
        for (int j = 0; j < nx; ++j)
            if (coef[j] != 0.0) ++nnz, at = j;
        const double ev = nnz == 1 ? coef[at] : 0.0;
        if ((ev == 1.0 || ev == -1.0) && ineq && !hp.opt.no_selection_rows) {
            eoff_or_comp = at;
            return ev < 0.0 ? (int)kEOneHotNeg : (int)kEOneHot;
        }
        eoff_or_comp = push(coef, nx);
        return (int)kEDense;
    };
    P.meq = P.mineq = 0;
    P.row_f_inst = nullptr;
    P.lb_inst = P.ub_inst = nullptr;
    hp.cstr_row0.assign((size_t)(n_cstrs > 0 ? n_cstrs : 1), -1);
    hp.cstr_per_step.assign((size_t)(n_cstrs > 0 ? n_cstrs : 1), 0);
    hp.cstr_steps.assign((size_t)(n_cstrs > 0 ? n_cstrs : 1), 0);
    for (int pass = 0; pass < 2; ++pass) { // pass 0: equalities, pass 1: inequalities (LMPC.cpp:257-271)
        for (int k = 0; k < n_cstrs; ++k) {
            const copra_cstr_desc_t& c = cstrs[k];
            if (c.kind == COPRA_CSTR_CONTROL_BOUND) continue;
            const bool ineq = (c.kind == COPRA_CSTR_TRAJECTORY_BOUND) || c.is_inequality;
            if ((pass == 1) != ineq) continue;
            const int before = (int)hp.row_f.size();
            const int r = c.rows;
            switch (c.kind) {
            case COPRA_CSTR_TRAJECTORY: // constraints.cpp:66-84
                if (c.e_cols == nx) {
                    std::vector<int> eo((size_t)r), hot((size_t)r, -1), neg((size_t)r, 0);
                    for (int i = 0; i < r; ++i) {
                        eo[(size_t)i] = push_row(c.E, r, nx, i);
                        // a row of E that SELECTS one component (one entry, equal to 1 or -1: a velocity limit written as +-x_c <= f) is, up to its sign, the
                        // row of Psi a TrajectoryBoundConstraint would give -- the same arithmetic (1 x_c, sums of zeros), and the
                        // controller keeps the compact variant of the Riccati-factor tier and the hand-over from the lane pass
                        int nnz = 0, at = -1;
                        for (int j = 0; j < nx; ++j)
                            if (c.E[(size_t)j * r + i] != 0.0) ++nnz, at = j;
                        const double ev = nnz == 1 ? c.E[(size_t)at * r + i] : 0.0;
                        if ((ev == 1.0 || ev == -1.0) && c.is_inequality && !hp.opt.no_selection_rows) {
                            hot[(size_t)i] = at;
                            neg[(size_t)i] = ev < 0.0; // (-x_c <= -l: a lower limit)
                        }
                    }
                    for (int s = 0; s <= N; ++s)
                        for (int i = 0; i < r; ++i) {
                            if (hot[(size_t)i] >= 0)
                                add_row(s, neg[(size_t)i] ? kEOneHotNeg : kEOneHot, hot[(size_t)i], kGNone, -1, c.f[i]);
                            else
                                add_row(s, kEDense, eo[(size_t)i], kGNone, -1, c.f[i]);
                        }
                } else {
                    for (int i = 0; i < r; ++i) {
                        int sx = -1;
                        if (step_rows && single_block(c.E, r, nx, N + 1, i, sx) && sx >= 0) {
                            std::vector<double> coef((size_t)nx);
                            for (int j = 0; j < nx; ++j) coef[(size_t)j] = c.E[(size_t)(sx * nx + j) * r + i];
                            int eoc = -1;
                            const int ek = state_kind(coef.data(), c.is_inequality != 0, eoc);
                            add_row(sx, ek, eoc, kGNone, -1, c.f[i]);
                        } else {
                            add_row(0, kEFull, push_row(c.E, r, X, i), kGNone, -1, c.f[i]);
                        }
                    }
                }
                break;
            case COPRA_CSTR_CONTROL: // constraints.cpp:137-148
                if (c.g_cols == nu) {
                    std::vector<int> go((size_t)r);
                    for (int i = 0; i < r; ++i) go[(size_t)i] = push_row(c.G, r, nu, i);
                    for (int s = 0; s < N; ++s)
                        for (int i = 0; i < r; ++i) add_row(s, kENone, -1, kGStep, go[(size_t)i], c.f[i]);
                } else {
                    for (int i = 0; i < r; ++i) {
                        int su = -1;
                        if (step_rows && single_block(c.G, r, nu, N, i, su) && su >= 0)
                            add_row(su, kENone, -1, kGStep, push_block(c.G, r, nu, su, i), c.f[i]);
                        else
                            add_row(0, kENone, -1, kGFull, push_row(c.G, r, U, i), c.f[i]);
                    }
                }
                break;
            case COPRA_CSTR_MIXED: // constraints.cpp:197-226
                if (c.e_cols == nx) {
                    std::vector<int> eo((size_t)r), go((size_t)r);
                    for (int i = 0; i < r; ++i) {
                        eo[(size_t)i] = push_row(c.E, r, nx, i);
                        go[(size_t)i] = push_row(c.G, r, nu, i);
                    }
                    for (int s = 0; s < N; ++s)
                        for (int i = 0; i < r; ++i) add_row(s, kEDense, eo[(size_t)i], kGStep, go[(size_t)i], c.f[i]);
                } else {
                    for (int i = 0; i < r; ++i) {
                        int sx = -1, su = -1;
                        const bool one = step_rows && single_block(c.E, r, nx, N + 1, i, sx) && single_block(c.G, r, nu, N, i, su);
                        if (one && sx >= 0 && su >= 0 && sx == su) { // E x_s + G u_s: a per-step mixed row
                            add_row(sx, kEDense, push_block(c.E, r, nx, sx, i), kGStep, push_block(c.G, r, nu, su, i), c.f[i]);
                        } else if (one && sx >= 0 && su < 0) { // no control part at all
                            std::vector<double> coef((size_t)nx);
                            for (int j = 0; j < nx; ++j) coef[(size_t)j] = c.E[(size_t)(sx * nx + j) * r + i];
                            int eoc = -1;
                            const int ek = state_kind(coef.data(), c.is_inequality != 0, eoc);
                            add_row(sx, ek, eoc, kGNone, -1, c.f[i]);
                        } else if (one && sx < 0 && su >= 0) { // no state part at all
                            add_row(su, kENone, -1, kGStep, push_block(c.G, r, nu, su, i), c.f[i]);
                        } else {
                            add_row(0, kEFull, push_row(c.E, r, X, i), kGFull, push_row(c.G, r, U, i), c.f[i]);
                        }
                    }
                }
                break;
            case COPRA_CSTR_DENSE:
                // row i:  Y_i x_0 + A_i U  <=|=  z_i  (InitialStateLMPC.cpp:88-102), or  A_i U <=|= b_i  with the
                // host-evaluated b = z - Y x0 (LMPC.cpp:257-271)
                for (int i = 0; i < r; ++i) {
                    if (is)
                        add_row(0, kEDense, push_row(c.Y, r, nx, i), kGFull, push_row(c.A, r, U, i), c.z[i]);
                    else
                        add_row(0, kENone, -1, kGFull, push_row(c.A, r, U, i), c.b[i]);
                }
                break;
            case COPRA_CSTR_TRAJECTORY_BOUND: { // constraints.h:248-255, constraints.cpp:284-315
                // reference quirk Q1 reproduced: lower rows keep the SAME orientation as upper rows (x <= lower)
                const bool full = (c.rows == X);
                for (int bpass = 0; bpass < 2; ++bpass) {
                    const double* bound = bpass == 0 ? c.lower : c.upper;
                    for (int s = 0; s <= N; ++s) {
                        for (int line = 0; line < c.rows; ++line) {
                            if (bpass == 0 ? is_neg_inf(bound[line]) : is_pos_inf(bound[line])) continue;
                            const int row = line + nx * s;
                            add_row(row / nx, kEOneHot, row % nx, kGNone, -1, bound[line]);
                        }
                        if (full) break;
                    }
                }
                break;
            }
            default:
                break;
            }
            const int added = (int)hp.row_f.size() - before;
            if (c.kind != COPRA_CSTR_TRAJECTORY_BOUND && r > 0) { // (bound rows skip infinite components: no map)
                hp.cstr_row0[(size_t)k] = before;
                hp.cstr_per_step[(size_t)k] = r;
                hp.cstr_steps[(size_t)k] = added / r;
            }
            if (ineq)
                P.mineq += added;
            else
                P.meq += added;
        }
    }
    P.mgen = P.meq + P.mineq;
    { // row i one step earlier: same coefficients, row_step - 1 (per-step entries only)
        hp.row_prev.assign((size_t)(P.mgen > 0 ? P.mgen : 1), -1);
        for (int i = 0; i < P.mgen; ++i) {
            if (hp.row_ekind[(size_t)i] == kEFull || hp.row_gkind[(size_t)i] == kGFull) continue;
            for (int j = i - 1; j >= 0; --j) // (rows of one constraint are stacked step-major: the match is close by)
                if (hp.row_step[(size_t)j] == hp.row_step[(size_t)i] - 1 && hp.row_ekind[(size_t)j] == hp.row_ekind[(size_t)i]
                    && hp.row_eoff[(size_t)j] == hp.row_eoff[(size_t)i] && hp.row_gkind[(size_t)j] == hp.row_gkind[(size_t)i]
                    && hp.row_goff[(size_t)j] == hp.row_goff[(size_t)i] && (i < P.meq) == (j < P.meq)) {
                    hp.row_prev[(size_t)i] = j;
                    break;
                }
        }
    }
    P.warm_set = nullptr;
    P.row_prev = nullptr;
    P.n_full_rows = 0;
    for (int i = 0; i < kMaxFullRows; ++i) P.full_row[i] = -1;
    for (int i = 0; i < P.mgen && P.n_full_rows >= 0; ++i) {
        if (hp.row_ekind[(size_t)i] == kEFull || hp.row_gkind[(size_t)i] == kGFull) {
            if (P.n_full_rows == kMaxFullRows)
                P.n_full_rows = -1;
            else
                P.full_row[P.n_full_rows++] = i;
        }
    }
    const int nvar = is ? nx + U : U; // InitialStateLMPC optimises [x0; U] (InitialStateLMPC.cpp:52-75)
    P.mtotal = P.mgen + 2 * nvar; // QuadProgSolver.cpp:51
    if (hp.params.empty()) hp.params.assign(2, 0.0);

    P.vsmall = qpgen2_vsmall();
    P.max_iter = 50 * (nvar + P.mtotal) + 100;

    // fused-kernel limits
    P.use_large = 0;
    if (nvar > kLargeMaxN || (is && nx > 16)) {
        // The condensed Goldfarb-Idnani kernels stop here (one thread per row of a 512 x 512 inverse factor; xDim x xDim scratch of
        // InitialStateLMPC).  The stage-wise Riccati interior-point method has no object of that size: such a controller is
        // accepted if it is stage-wise (stage_plan.hpp decides when the handle is created) and then runs on that path alone --
        // instances it does not converge on keep status 3 instead of being re-queued.
        if (nu > kMaxNu) return hp.error = "uDim > 8 is not covered", COPRA_ERR_UNSUPPORTED;
        hp.large = true;
        hp.ric_only = true;
        P.use_large = 0;
        return COPRA_OK;
    }
    if (nvar > kWave) { // workgroup-per-instance kernel: J / R in HBM (lmpc_large.hpp)
        if (nu > kMaxNu) return hp.error = "uDim > 8 is not covered", COPRA_ERR_UNSUPPORTED;
        LargeLayout& L = P.large;
        int o = 0;
        auto take = [&](int count) {
            int at = o;
            o += align2(count);
            return at;
        };
        L.A = take(nx * nx);
        L.B = take(nx * nu);
        L.D = take(nx);
        L.X0 = take(nx);
        L.G = take(N * nx * nu);
        L.Xi = take(X);
        L.Xbar = is ? L.Xi : take(X); // InitialStateLMPC never uses the free response Phi x0 + xi
        L.Xcur = take(X);
        const int plimit = 6144;
        L.nparams = ((int)hp.params.size() <= plimit) ? (int)hp.params.size() : 0; // (dropped below if LDS gets too tight)
        L.Params = take(L.nparams);
        L.FullS = take(kMaxFullRows);
        const int sol0 = o;
        const int sol_end = layout_large_solver(L.sol, sol0, nvar, P.mgen, P.meq, P.mtotal);
        L.TL = L.sol.coef; // 4 nvar >= 256 >= nx^2 doubles; free between the two factorisations
        o = sol0; // the preview ping-pong blocks and the cost tables alias the solver regions
        L.PhiPP = take(2 * nx * nx);
        o = sol0;
        L.Y = take(N * P.rmax * nu);
        L.We = take((N + 1) * P.rmax);
        L.Cp = take(P.rmax * (nx + nu + 2));
        L.total = o > sol_end ? o : sol_end;
        L.threads = (nvar + kWave - 1) & ~(kWave - 1);
        L.ld = large_ld(nvar);
        long long w = 0;
        auto wtake = [&](long long count) {
            long long at = w;
            w += (count + 7) & ~7LL;
            return at;
        };
        L.wsF = wtake((long long)nvar * L.ld);
        L.wsJ = wtake((long long)nvar * L.ld);
        L.wsPhi = wtake((long long)(N + 1) * nx * nx);
        L.wsMPhi = wtake(is ? (long long)(N + 1) * P.rmax * nx : 0);
        L.wsE = wtake(is ? (long long)nx * U : 0);
        L.wsT = wtake(is ? (long long)nx * U : 0);
        L.ws_total = w;
        // the LDS copy of the parameters goes first when LDS is tight: when the layout does not fit at all, and when
        // dropping it brings the footprint under 80 KiB, i.e. lets a second workgroup share the CU
        const size_t with_p = (size_t)L.total * sizeof(double);
        const size_t without_p = (size_t)(L.total - align2(L.nparams)) * sizeof(double);
        if (L.nparams > 0 && (with_p > 160u * 1024u || (with_p > 80u * 1024u && without_p <= 80u * 1024u))) {
            L.total -= align2(L.nparams);
            L.nparams = 0;
            const int shift = align2((int)hp.params.size());
            L.sol.xs -= shift, L.sol.cv -= shift, L.sol.np -= shift, L.sol.dv -= shift, L.sol.rv -= shift;
            L.sol.uv -= shift, L.sol.hv -= shift, L.sol.coef -= shift, L.sol.nb -= shift, L.sol.eqsgn -= shift;
            L.sol.red -= shift, L.sol.stage -= shift, L.sol.dblk -= shift, L.sol.act -= shift, L.sol.iact -= shift;
            L.sol.total -= shift;
            L.TL -= shift, L.PhiPP -= shift, L.Y -= shift, L.We -= shift, L.Cp -= shift, L.FullS -= shift;
        }
        hp.lds_bytes = hp.lds_full_bytes = (size_t)L.total * sizeof(double);
        if (hp.lds_bytes > 160u * 1024u) return hp.error = "problem does not fit the 160 KiB LDS of one CU", COPRA_ERR_UNSUPPORTED;
        hp.two_tier = false;
        hp.large = true;
        P.use_large = 1;
        return COPRA_OK;
    }
    if (is) {
        layout_lds(hp.lds_full, nx, nu, N, nvar, X, P.rmax, P.mgen, P.meq, P.mtotal, true);
        LdsLayout& L = hp.lds_full;
        int o = L.total;
        P.isl.ldq = (U % 2 == 0) ? U + 1 : U;
        P.isl.Jq = o, o += align2(U * P.isl.ldq);
        P.isl.E = o, o += align2(nx * U);
        P.isl.MPhi = o, o += align2((N + 1) * P.rmax * nx);
        L.total = o;
        hp.lds_full_bytes = (size_t)o * sizeof(double);
        if (hp.lds_full_bytes > 160u * 1024u) return hp.error = "problem does not fit the 160 KiB LDS of one CU", COPRA_ERR_UNSUPPORTED;
        hp.two_tier = false;
        P.lds = hp.lds_full;
        hp.lds_bytes = hp.lds_full_bytes;
        return COPRA_OK;
    }
    if (nu > kMaxNu) {
        hp.error = "uDim > 8 is not covered by the one-wave fused kernel";
        return COPRA_ERR_UNSUPPORTED;
    }
    {
        const int rp = specialised_cost_rows(nx, nu, N, P.rmax, P.rfull); // the specialised kernels pad every cost to rp rows
        const int rows = rp > P.rmax ? rp : P.rmax;
        layout_lds(hp.lds_full, nx, nu, N, U, X, rows, P.mgen, P.meq, P.mtotal, true, false, 0, P.rfull);
        hp.lds_full_bytes = (size_t)hp.lds_full.total * sizeof(double);
        if (hp.lds_full_bytes > 160u * 1024u) {
            hp.error = "problem does not fit the 160 KiB LDS of one CU";
            return COPRA_ERR_UNSUPPORTED;
        }
        // Occupancy: the 160 KiB of a CU admit 4 waves (one per SIMD) at <= 40 KiB each.  If the full layout does not
        // reach that but a compact one (aliased build tables, R capped) does, run two-tier.
        const int quarter = (160 * 1024 / 4) / (int)sizeof(double);
        LdsLayout compact {};
        hp.two_tier = false;
        P.lds = hp.lds_full;
        if (hp.lds_full.total > quarter
            && layout_lds(compact, nx, nu, N, U, X, rows, P.mgen, P.meq, P.mtotal, true, true, quarter, P.rfull)
            && compact.total <= quarter && compact.rcap >= 8 && compact.rcap < U) {
            hp.two_tier = true;
            P.lds = compact;
        }
        // Run-time shapes (no compile-time instantiation): the same trade at finer grain -- the densest compact layout
        // (160 KiB / k per instance) that still leaves R room for a quarter of the variables (at least 8 columns);
        // instances that need more finish in the second tier.  Measured at 30 variables (CoM preview, N = 10):
        // 7 instances per CU in the full layout 17.4 M solves/s, k = 12 with the 128-VGPR build of the kernel 28.8 M.
        hp.lds_safe = P.lds;
        hp.safe_two_tier = hp.two_tier;
        if (rp == 0 && !hp.opt.no_dense_layout) {
            static const int ks[] = { 16, 14, 12, 10, 8, 7, 6, 5 };
            const int need = U < 8 ? U : ((U + 3) / 4 > 8 ? (U + 3) / 4 : 8);
            for (int k : ks) {
                const int budget = (160 * 1024 / k) / (int)sizeof(double);
                if (budget >= P.lds.total) continue; // no denser than what is already chosen
                LdsLayout c2 {};
                if (layout_lds(c2, nx, nu, N, U, X, rows, P.mgen, P.meq, P.mtotal, true, true, budget, P.rfull)
                    && c2.total <= budget && c2.rcap >= need) {
                    hp.two_tier = c2.rcap < U;
                    hp.dense = hp.two_tier; // (with room for every column there is nothing to fall back from)
                    P.lds = c2;
                    break;
                }
            }
        }
        // Factor-only first tier (LdsLayout::tri): without the n x n inverse factor the instance is about half the
        // size.  Taken when it lets more instances share a CU than the layout chosen so far and still leaves the
        // active set `need` columns; the instances that outgrow them finish in the second tier as above.
        // Measured (M solves/s, square layout -> factor-only): headline shape 13.4 -> 21.3; run-time shapes with 45
        // variables 10.0 -> 19.0, 48: 10.1 -> 13.9, 64: 4.1 -> 9.8.  Up to 32 variables the packed kernels and the dense
        // square layouts already fill the wave slots, so those shapes stay as they are.
        // The Riccati-factor tier (lmpc_fused_ric.hpp) is instantiated for the CoM shape at three horizons; N = 20 is chosen
        // inside the factor-only block below (it competes with the layouts there), the shorter ones here -- they would
        // otherwise run on the square layouts (30 variables) or the run-time-shape factor-only kernel (45)
        bool ric_short = nx == 6 && nu == 3 && (N == 10 || N == 15) && P.rmax <= 6 && P.rfull == 0 && P.denseQ < 0 && !P.initial_state
            && P.ncost <= kRicMaxCosts && !hp.opt.no_ric && !hp.opt.no_tri;
        // (every other shape the body of that tier can be instantiated for gets there through copra_batch_specialise, which
        //  compiles the kernel and calls take_ric_layout)
        for (int t = 0; t < P.ncost; ++t) ric_short = ric_short && !P.cost[t].full;
        bool ric_taken = false;
        for (int k = 8; ric_short && !ric_taken && k >= 6; --k) {
            const int budget = ((160 * 1024 / k) & ~511) / (int)sizeof(double);
            LdsLayout t {};
            if (layout_lds_ric(t, nx, nu, N, U, X, P.mgen, P.meq, P.mtotal, P.rows_pure != 0 && !hp.opt.ric_general, kFusedQ1Regs, budget)) {
                hp.lds_safe = P.lds;
                hp.safe_two_tier = hp.two_tier;
                hp.two_tier = true;
                hp.dense = true;
                P.lds = t;
                P.ric_tab = build_ric_tables(hp, 6);
                ric_taken = true;
            }
        }
        if (!ric_taken && U > 32 && !hp.opt.no_tri) {
            const int kenv = 0;
            const int need = rp > 0 ? 5 : ((U + 7) / 8 > 5 ? (U + 7) / 8 : 5);
            // the headline instantiation keeps five columns of Q1 in registers (kFusedQ1Regs): 8 instances per CU
            const int qregs = (nx == 6 && rp == 6 && !hp.opt.no_q1regs) ? kFusedQ1Regs : 0;
            bool all_ident = true; // (state costs with M = I padded to nx rows: lmpc_fused.hpp reads G instead of Y)
            for (int t = 0; t < P.ncost; ++t)
                all_ident = all_ident && (P.cost[t].kind == kCostControl || (P.cost[t].ident && rows == nx));
            // Riccati form of the factor (lmpc_fused_ric.hpp): every cost a per-step entry, the headline instantiation
            bool ric_ok = qregs > 0 && nu == 3 && N == 20 && P.rfull == 0 && P.denseQ < 0 && !P.initial_state && P.ncost <= kRicMaxCosts
                && !hp.opt.no_ric;
            for (int t = 0; t < P.ncost; ++t) ric_ok = ric_ok && !P.cost[t].full;
            for (int k = kenv > 0 ? kenv : 8; k >= 2; --k) {
                const int budget = ((160 * 1024 / k) & ~511) / (int)sizeof(double); // (LDS is granted in 512-byte units)
                if (budget >= P.lds.total && !kenv) break; // no denser than what is already chosen
                LdsLayout t {};
                // (COPRA_RIC_K = instances per CU: start on the LDS-Q1 step of the ladder that adapt_layout would reach -- experiments, tests)
                const int rk = hp.opt.ric_k > 0 ? hp.opt.ric_k : 0;
                const int rbudget = rk ? ((160 * 1024 / rk) & ~511) / (int)sizeof(double) : budget;
                if (ric_ok && k >= 6 // (general state rows keep their own trajectory buffer: seven instances per CU)
                    && layout_lds_ric(t, nx, nu, N, U, X, P.mgen, P.meq, P.mtotal, P.rows_pure != 0 && !hp.opt.ric_general, rk ? 0 : qregs, rbudget)) {
                    hp.lds_safe = P.lds;
                    hp.safe_two_tier = hp.two_tier;
                    hp.two_tier = true;
                    hp.dense = true;
                    P.lds = t;
                    P.ric_tab = build_ric_tables(hp, rp);
                    break;
                }
                t = LdsLayout {};
                if (qregs > 0
                    && layout_lds(t, nx, nu, N, U, X, rows, P.mgen, P.meq, P.mtotal, true, true, budget, 0, true, P.rows_direct != 0, qregs,
                        all_ident)
                    && t.total <= budget) {
                    hp.lds_safe = P.lds;
                    hp.safe_two_tier = hp.two_tier;
                    hp.two_tier = true;
                    hp.dense = true;
                    P.lds = t;
                    break;
                }
                // full-size cost entries at the headline shape: the same five register columns (the tile of the dense contraction and
                // they fit the 256 VGPRs of two waves per SIMD) -- one instance per CU more than with Q1 in LDS
                if (P.rfull > 0 && nx == 6 && nu == 3 && N == 20 && !hp.opt.no_q1regs
                    && layout_lds(t, nx, nu, N, U, X, rows, P.mgen, P.meq, P.mtotal, true, true, budget, P.rfull, true, P.rows_direct != 0,
                        kFusedQ1Regs)
                    && t.total <= budget) {
                    hp.lds_safe = P.lds;
                    hp.safe_two_tier = hp.two_tier;
                    hp.two_tier = true;
                    hp.dense = true;
                    P.lds = t;
                    break;
                }
                t = LdsLayout {};
                if (layout_lds(t, nx, nu, N, U, X, rows, P.mgen, P.meq, P.mtotal, true, true, budget, P.rfull, true, P.rows_direct != 0)
                    && t.total <= budget && t.rcap >= need) {
                    hp.lds_safe = P.lds; // what the adaptive fall-back steps to
                    hp.safe_two_tier = hp.two_tier;
                    hp.two_tier = true;
                    hp.dense = true;
                    P.lds = t;
                    break;
                }
                if (kenv) break;
            }
        }
        hp.lds_bytes = (size_t)P.lds.total * sizeof(double);
    }
    // the tables of the one-instance-per-lane pass (lmpc_lane.hpp) for a controller that did not get them with the Riccati-factor tier:
    // in front of the other one-wave first tiers the pass only filters (the instances at their unconstrained minimiser end in it)
    // Shapes the library holds the Riccati-factor tier for at ANY horizon (ric_aot_shape): taken here, as copra_batch_specialise does for
    // the shapes it compiles.  (Single-control systems below 48 variables stay on the packed / factor-only kernels: the reference's
    // falling-mass problems hold most of their control bounds active, far beyond the tier's five register columns -- measured, §3.7.)
    if (!P.lds.ric && !hp.large && ric_aot_shape(nx, nu) && ric_shape_ok(nx, nu, N) && !hp.opt.no_ric && !hp.opt.no_tri
        && (nu >= 2 || U >= 48))
        (void)take_ric_layout(hp);
    if (P.lane_tab < 0 && !hp.large && !P.initial_state) build_lane_tables(hp);
    return COPRA_OK;
}

// fix up the pointers of hp.plan to the host vectors (emulator) -- the product path re-points them to HBM copies
inline void point_plan_to_host(HostPlan& hp)
{
    FusedPlan& P = hp.plan;
    P.row_step = hp.row_step.data();
    P.row_ekind = hp.row_ekind.data();
    P.row_eoff = hp.row_eoff.data();
    P.row_gkind = hp.row_gkind.data();
    P.row_goff = hp.row_goff.data();
    P.row_f = hp.row_f.data();
    P.row_prev = hp.row_prev.data();
    P.params = hp.params.data();
    P.lb = hp.lb.data();
    P.ub = hp.ub.data();
    P.is_R = hp.isR.empty() ? nullptr : hp.isR.data();
    P.is_r = hp.isr.empty() ? nullptr : hp.isr.data();
}

} // namespace copra_hip
