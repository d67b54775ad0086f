// plan.hpp -- the "compiled" form of one batched LMPC controller, shared by host and device code.
//
// copra_batch_create() turns the user's cost / constraint descriptors (the arguments of LMPC::addCost /
// addConstraint, reference src/LMPC.cpp:118-128) into
//   * a parameter blob (doubles) holding every M / N / p / w / E / G matrix, column-major,
//   * a list of cost terms (offsets into the blob),
//   * a table of constraint ROWS in the order LMPC::makeQPForm stacks them (reference src/LMPC.cpp:257-271:
//     equalities in insertion order, then inequalities in insertion order), each row being
//         E_row . x_k  (+ full-horizon E_row . X)  +  G_row . u_k  (+ full-horizon G_row . U)   <=|=   f
//     so that A = E Psi + G never has to be materialised (reference src/constraints.cpp:66-84, 137-148, 197-226,
//     284-315 all reduce to this form),
//   * lb / ub (reference src/constraints.cpp:359-367, src/LMPC.cpp:274-279).
// The device kernel interprets this plan once per instance; control flow over the plan is wave-uniform.
#pragma once

namespace copra_hip {

constexpr int kMaxCosts = 8;
constexpr int kMaxFullRows = 16; // full-size constraint rows the workgroup-per-instance kernel evaluates cooperatively
constexpr int kMaxNu = 8; // register arrays in the Hessian recursion (uDim <= 8 on the fused path)
constexpr int kWarmCap = 32; // active rows remembered per instance for the warm start
constexpr int kRicMaxCosts = 3; // cost terms the Riccati-factor tier takes (lmpc_fused_ric.hpp keeps the loads of all of them in flight)
constexpr int kFusedQ1Regs = 5; // columns of Q1 the headline first-tier kernel keeps in registers (gi_core.hpp, QR)
// lanes one instance works with: the 64-lane wavefront, or a 16-lane DPP row of it in the packed small-problem build
// (copra_hip_packed.hip compiles the same kernel bodies with COPRA_WAVE_WIDTH = 16: four instances per wavefront)
#ifndef COPRA_WAVE_WIDTH
#define COPRA_WAVE_WIDTH 64
#endif
constexpr int kWave = COPRA_WAVE_WIDTH;

// cost kinds (== copra_cost_kind_t)
enum { kCostTrajectory = 0, kCostTarget = 1, kCostControl = 2, kCostMixed = 3 };

// state part of a constraint row
enum {
    kENone = 0, // no state term
    kEDense = 1, // nx coefficients at params[eoff..] applied to x_step
    kEOneHot = 2, // coefficient 1 on component `eoff` of x_step (rows of Psi: TrajectoryBoundConstraint)
    kEFull = 3, // fullXDim coefficients at params[eoff..] applied to the whole trajectory X (full-size entry)
    kEOneHotNeg = 4 // coefficient -1 on component `eoff` of x_step (a lower limit written as -x_c <= -l: TrajectoryConstraint with E = -e_c')
};
// a row whose state part is +- one component of one state, and that sign
#if defined(__HIPCC__)
__host__ __device__
#endif
inline bool e_onehot(int ek) { return ek == kEOneHot || ek == kEOneHotNeg; }
#if defined(__HIPCC__)
__host__ __device__
#endif
inline double e_sign(int ek) { return ek == kEOneHotNeg ? -1.0 : 1.0; }
// control part of a constraint row
enum {
    kGNone = 0,
    kGStep = 1, // nu coefficients at params[goff..] applied to u_step
    kGFull = 2 // fullUDim coefficients at params[goff..] applied to U (full-size entry)
};

// Shapes that get their own compile-time instantiation of the fused kernel (the BASELINE.json configs).  Returns the
// padded cost-row count RP of the specialisation, or 0 when (nx, nu, N, rmax) runs on the generic instantiation.
// Used by the plan builder (LDS sizing), the HIP launcher and the CPU emulator so that all three agree.
inline int specialised_cost_rows(int nx, int nu, int N, int rmax, int rfull = 0)
{
    if (rfull > 0) return 0; // full-size cost entries run on the generic instantiation
    if (nx == 6 && nu == 3 && N == 20 && rmax <= 6) return 6;
    if (nx == 2 && nu == 1 && N == 10 && rmax <= 2) return 2;
    return 0;
}

struct CostTerm {
    int kind;
    int rows; // r
    int offM; // per-step: r x nx column-major; full-size: r x fullXDim ROW-major (one contiguous row per cost row); or -1
    int offN; // per-step: r x nu column-major; full-size: r x fullUDim ROW-major; or -1
    int offP; // r
    int offW; // r
    int full; // 1: full-size entry (costFunctions.cpp:65-71, 141-146, 197-203) -> dense MFMA contraction
    int offMask; // full-size entry with M: which K-steps (four columns of M) of each block of sixteen rows are not all zero -- per row block
                 // ceil(ceil(X / 4) / 64) 64-bit words, each as two doubles (low | high 32 bits); -1: none (every K-step is visited)
    int pstride; // 0: one reference p for every step.  r (= rows): a REFERENCE TRAJECTORY -- the reference of step k sits at offP + k r.
                 // (A full-size entry whose M / N is block-diagonal with identical blocks and identical weights per step -- the only way the
                 // reference's API can express a reference that changes along the horizon, costFunctions.cpp:63-82, 139-156 -- is such a
                 // per-step entry: plan_builder.hpp.)
    int prows; // length of the whole reference: rows, or rows x steps with pstride (what a per-instance reference holds per instance)
    int ident; // 1: per-step entry whose M is the xDim x xDim identity (the usual "track the whole state" cost): the
               // products M G_k ARE the blocks G_k, bit for bit, so the cost phase reads G instead of forming them
};

// LDS carve-up, offsets in doubles from the dynamic-LDS base (all multiples of 2 doubles = 16 bytes)
struct LdsLayout {
    int A, B, D, X0; // system matrices of this instance
    int G; // N blocks G_k = A^k B (nx x nu, column-major): Psi_{i,j} = G_{i-1-j}  (PreviewSystem.cpp:57-74)
    int Xbar; // free response  Phi x0 + xi               (fullXDim)
    int Xcur; // current trajectory Xbar + Psi U          (fullXDim)
    int J, ldj; // n x ldj: Hessian (upper) -> Cholesky factor -> J = R^-1 (row i at J + i*ldj)
    int tri; // 1: factor-only layout (gi_core.hpp, TRI): J holds the PACKED upper triangle (entry (i, c), i <= c, at
             // c (c + 1) / 2 + i) of the Hessian -> its Cholesky factor, J = R^-1 is never formed, Q1 holds the
             // orthonormal basis of the active normals (rcap columns of 64), dv / the 4n coefficients do not exist and
             // the diagonal slots of the factor carry 1 / R(i,i)
    int Q1;
    int q1regs; // > 0 (factor-only layout): that many columns of Q1 live in registers, there is no Q1 region in LDS
    int ric; // 1 (with tri and q1regs): the J region holds the RICCATI form of the factor (ric_factor.hpp: N stage records
             // instead of the packed triangle), A / B / d / x0 keep their own slots and there are no cost tables
    int ricX; // general variant: 64 doubles behind A | B | d | x0 -- together the trajectory of the roll-out (lmpc_fused_ric.hpp)
    int ricD; // two doubles nobody reads (what the lanes with nothing to store write to)
    int ricKv; // the feed-forward terms kv of the unconstrained minimiser, nu per stage: written by the sweep, read once by the roll-out (RicRec)
    int ricC; // 1: compact variant (every state term of a row is one component of one state): once the row norms are known the
              // blocks G are dead -- the normal of a state row enters w = R^-T n as a unit injection into the recursion's state --
              // and their region holds the maintained trajectory (at G) and the closed-loop states of z = R^-1 v (at Xbar)
    int ricS; // scratch of the Riccati sweep (aliases the solver vectors, which are written after it)
    int R; // packed upper-triangular R of the active set (rcap columns)
    int rcap; // number of active constraints R has room for: n in the full layout, fewer in the compact (tier-1) one
    int xs, dv, zv, uv, ap, coef, cvec; // solver vectors (n; uv n+1; coef 4n)
    int nb; // norms of the general rows (m_gen)
    int eqsgn; // current orientation of each equality row (meq)
    int scal; // 8 scalars
    int act; // unsigned char[m_total] active flags   (offset in doubles)
    int iact; // int[n+1]
    // build-phase scratch (aliases R and beyond; sized on the host)
    int BldPhi; // (N+1) blocks nx x nx
    int BldXi; // fullXDim
    int BldY; // N blocks r x nu   (M G_k)
    int BldWe; // (N+1) blocks r    (w .* (M xbar_k - p))
    int BldCp; // parameters of the cost being processed: M (r x nx) | N (r x nu) | p (r) | w (r)
    int BldFull; // full-size costs: weighted residuals (rfull)
    int total; // total doubles
};

// extra LDS regions of the InitialStateLMPC variant (islmpc_fused.hpp)
struct IsLayout {
    int Jq, ldq; // n x ldq: Hessian of the U block -> its Cholesky factor -> inverse factor
    int E; // nx x n: top-right block of the Hessian
    int MPhi; // (N+1) blocks r x nx
};

// LDS regions of the workgroup-per-instance solver (gi_large.hpp), offsets in doubles
struct LargeLds {
    int xs, cv, np, dv, rv, uv, hv, coef, nb, eqsgn, red, stage, dblk, act, iact, total;
};

constexpr int kNB = 8; // panel width of the blocked Cholesky factorisation / triangular inversion
constexpr int kLargeMaxN = 512; // thread = row: workgroup size = n rounded up to a wave, at most 512 threads
constexpr int kLargeMaxWaves = kLargeMaxN / 64;

inline int large_ld(int n) { return (n + 7) & ~7; }

// `o` = first free double; returns the first free double after the solver regions
inline int layout_large_solver(LargeLds& L, int o, int n, int mgen, int meq, int mtotal)
{
    auto take = [&](int count) {
        int at = o;
        o += (count + 1) & ~1;
        return at;
    };
    L.xs = take(n);
    L.cv = take(n);
    L.np = take(n);
    L.dv = take(n);
    L.rv = take(n);
    L.uv = take(n + 2);
    L.hv = take(n);
    L.coef = take(4 * n);
    L.nb = take(mgen > 0 ? mgen : 1);
    L.eqsgn = take(meq > 0 ? meq : 1);
    L.red = take(4 * kLargeMaxWaves + 4);
    L.stage = L.np; // kNB * n doubles over np | dv | rv | uv | hv | coef (9n + 2): dead while a factorisation runs
    L.dblk = take(kNB * kNB + 2); // + the "pivot not positive" flag
    L.act = take((mtotal + 7) / 8 + 1); // one byte per row
    L.iact = take((n + 2) / 2 + 1);
    L.total = o;
    return o;
}

// Workgroup-per-instance LMPC / InitialStateLMPC kernel (lmpc_large.hpp; 64 < decision variables <= 512).
// LDS offsets in doubles; the cost-phase tables (Y, We, Cp) alias the solver regions, which are not live yet.
struct LargeLayout {
    int A, B, D, X0; // system matrices of this instance
    int G; // N blocks G_k = A^k B
    int Xi, Xbar, Xcur; // fullXDim each
    int PhiPP; // two nx x nx blocks: Phi_{s-1}, Phi_s of the preview recursion (aliases the solver regions)
    int TL; // nx x nx: top-left Hessian block of the InitialStateLMPC variant (aliases sol.coef)
    int FullS; // kMaxFullRows doubles: left-hand sides of the full-size rows at the current iterate
    int Params, nparams; // LDS copy of the parameter blob (nparams == 0: too large, stays in HBM)
    int Y, We, Cp; // cost tables (see LdsLayout)
    LargeLds sol;
    int total; // LDS doubles
    int threads; // workgroup size: number of QP variables rounded up to a wave
    int ld; // leading dimension of F / J
    // per-workgroup HBM workspace, offsets in doubles
    long long wsF, wsJ; // nv x ld each
    long long wsPhi; // (N+1) blocks nx x nx
    long long wsMPhi; // (N+1) blocks rmax x nx   (InitialStateLMPC)
    long long wsE, wsT; // nx x n each              (InitialStateLMPC)
    long long ws_total;
};

#if defined(__HIPCC__)
#define COPRA_HD __host__ __device__
#else
#define COPRA_HD
#endif
// offsets (doubles) of the parts of FusedPlan::ric_model; returns the total
COPRA_HD inline int ric_model_offsets(int nx, int nu, int N, int mgen, int& oBk, int& oG, int& oNb)
{
    const int rec = (nx * nx + nx * nu + nu * (nu + 1) / 2 + 1) & ~1; // RicRec<NX, NU>::SZ
    const int cst = (nx * nu + nx + 3 + 1) & ~1; // RicRec<NX, NU>::CST (B | d | zero | spare | one behind the records)
    oBk = N * rec + cst; // the feed-forward terms kv of the unconstrained minimiser, nu per stage (RicRec: they are not part of the records)
    oG = oBk + ((N * nu + 1) & ~1);
    oNb = oG + N * nx * nu;
    return oNb + ((mgen + 1) & ~1) + ((nx * nx + 1) & ~1); // (+ the system's A behind the norms: ric_model_A below)
}
// ... and of the fifth: A of the shared system (nx x nx, column-major) -- what an instance whose rows go through the free response of the
// preview (dense state rows: StageRows::refresh_trajectory) rebuilds  xbar_{k+1} = A xbar_k + d  from its own x0 with
COPRA_HD inline int ric_model_A(int nx, int nu, int N, int mgen)
{
    int a, b, oNb;
    (void)ric_model_offsets(nx, nu, N, mgen, a, b, oNb);
    return oNb + ((mgen + 1) & ~1);
}

// offsets (doubles) of the tables at FusedPlan::lane_tab:  H (nz x nz, column-major, z = (x, u)) | h (nz) | HN (nx x nx) | hN (nx) |
// rows: (N + 1) steps x lane_rps rows of [E (nx) | G (nu) | f | index of the row in the stacked order]  (a row that is not there: zeros,
// f = +inf, index -1) |
// per cost t < kRicMaxCosts and row r < 6: the coefficients of the reference p_t[r] in h (nz) and in hN (nx) -- lane_cref: what h and hN are
// rebuilt from, per lane, when a cost has per-instance references (copra_batch_set_cost_reference)
COPRA_HD inline void lane_tab_offsets(int nx, int nu, int& oh, int& oHN, int& ohN, int& oRows)
{
    const int nz = nx + nu;
    oh = nz * nz;
    oHN = oh + nz;
    ohN = oHN + nx * nx;
    oRows = (ohN + nx + 1) & ~1;
}

// rows of its LANE-MAJOR workspace per stage: K (nu x nx, column-major) | kv (nu) -- what its own roll-out reads back (and the first tier
// gathers K from: lmpc_fused_ric.hpp, from_lane)
COPRA_HD inline int lane_ws_rows(int nx, int nu) { return nu * nx + nu; }
// ... and doubles per instance of its INSTANCE-MAJOR hand-over block (FusedPlan::lane_ws2; round 5): what only the first tier reads --
// Lam^-1 (packed by rows, as RicRec) and kv of every stage ([N][nu (nu + 1) / 2 + nu]), then the running sums of the squared block-row
// norms of G_s = A^s B ([N][nx]).  Until round 5 these were twelve more lane-major rows per stage: the tier fetched each of the 240 values of an
// instance from a 64-byte sector of its own (the gather was bound by the sector requests of the CU's address unit: 704 per instance);
// as one contiguous block they are 30 sectors.
COPRA_HD inline int lane_ws2_doubles(int nx, int nu, int N) { return N * (nu * (nu + 1) / 2 + nu + nx); }
// LDS of that pass (doubles): the staging area of the transpositions (64 lanes x the widest array, odd stride), then H | h
constexpr int kLaneGroup = 4; // stages per group of its roll-out (results leave through LDS once per group)
constexpr int kLaneAhead = 1; // stages whose gains are in flight (round 5: two -- a stage of the roll-out carries two trajectories now and takes twice as long, and the 42 doubles of the other two buffers were what pushed the roll-out into scratch memory)
constexpr int kLaneHistBins = 32; // bins of its violated-row histogram (FusedPlan::lane_hist), the last one open
COPRA_HD inline int lane_lds_doubles(int nx, int nu, int& oH)
{
    int w = (nx * nx) | 1;
    if (((nx * nu) | 1) > w) w = (nx * nu) | 1;
    // the roll-out's group of stages: states | controls | norm sums (the hand-over block), all at once
    if (2 * ((kLaneGroup * nx) | 1) + ((kLaneGroup * nu) | 1) > w) w = 2 * ((kLaneGroup * nx) | 1) + ((kLaneGroup * nu) | 1);
    // the sweep: h of the lane | Lam^-1 of a group of stages (the hand-over block)
    { // (+ x0, d and M_uu,0^-1, parked)
        const int nl = nu * (nu + 1) / 2, hs = (3 * nx + nu + kLaneGroup * (nl + nu) + nl) | 1;
        if (hs > w) w = hs;
    }
    oH = 64 * w;
    const int nz = nx + nu;
    return (oH + nz * nz + nz + 1) & ~1;
}
// Tables of the one-(instance, AXIS)-per-lane solver (lmpc_axis.hpp; round 6) at FusedPlan::axis_tab, one block of axis_tab_doubles() per
// axis c -- state i of the system belongs to axis i % nu, control c to axis c; nxa = nx / nu states and ONE control per axis:
//     H (nz x nz, column-major, z = (x_axis, u_axis)) | h (nz) | HN (nxa x nxa) | hN (nxa) |
//     rows: (N + 1) steps x axis_rpa rows of [E (nxa) | g | f | index of the row in the stacked order]   (a row that is not there: zeros, f = +inf, index -1)
COPRA_HD inline void axis_tab_offsets(int nxa, int& oh, int& oHN, int& ohN, int& oRows)
{
    const int nz = nxa + 1;
    oh = nz * nz;
    oHN = oh + nz;
    ohN = oHN + nxa * nxa;
    oRows = (ohN + nxa + 1) & ~1;
}
COPRA_HD inline int axis_tab_doubles(int nxa, int N, int rpa)
{
    int oh, oHN, ohN, oRows;
    axis_tab_offsets(nxa, oh, oHN, ohN, oRows);
    return (oRows + (N + 1) * rpa * (nxa + 3) + 1) & ~1;
}
constexpr int kAxisGroup = 4; // steps per group of its result staging (U and X leave through LDS as contiguous segments per instance)
constexpr int kAxisMaxRpa = 2; // constraint rows per axis and step it takes
constexpr int kAxisMaxRef = 4; // cost rows with a reference per axis it takes (FusedPlan::axis_cref: the CoM model's TrajectoryCost has two per axis)
// ... and its LDS (doubles): the tables of every axis | the bounds of every axis (ub, lb: N each) -- read there by the builds whose tables change
// along the horizon -- | per lane (odd stride): the sparse coefficient / response array of the two recursions (N controls + (N + 1) rpa rows + a
// spare), the matrix S of its active set and the multipliers
COPRA_HD inline int axis_lds_doubles(int nx, int nu, int N, int rpa, int qmax, int& oBnd, int& oRC, int& rcs)
{
    const int nxa = nx / nu;
    oBnd = nu * axis_tab_doubles(nxa, N, rpa);
    oRC = oBnd + nu * 2 * N;
    // per lane: the sparse array (+ a spare entry) | S = N' Q^-1 N of its active set, lower triangle | the multipliers
    rcs = (N + (N + 1) * rpa + 1 + qmax * (qmax + 1) / 2 + qmax + (qmax > 8 ? 2 * qmax : 0)) | 1; // (QMAX > 8: the FACTOR of S instead of S, + g and r)
    int w = 64 * rcs;
    // (chains of three states: X and U leave one after the other through the same place -- 21 x 249 doubles would cost the fourth wave of a CU)
    const int stage_out = (64 / nu) * (nx * (N + 1) + (nxa >= 3 ? 0 : nu * N)) + (N + 1) * (nx / nu) + N + 2; // (+ the axis of an instance on a spare lane)
    // ... whose place the results of the wave's instances take at the end, as they lie in memory
    if (stage_out > w) w = stage_out;
    return (oRC + w + 1) & ~1;
}
// ... its launch: waves for `batch` instances of nu axes -- 64 / nu instances per wave on regular lanes, and where nu does not divide 64 one more
// instance per nu waves on their spare lanes (lmpc_axis.hpp) -- and how many instances sit on spare lanes
COPRA_HD inline int axis_grid(int nu, int batch, int& on_spare)
{
    const int ipw = 64 / nu, sp = 64 - ipw * nu;
    on_spare = 0;
    if (batch <= 0) return 0;
    if (sp == 0) return (batch + ipw - 1) / ipw;
    // the smallest w with ipw w + floor(w sp / nu) >= batch
    long long w = ((long long)batch * nu) / ((long long)ipw * nu + sp);
    while (ipw * w + (w * sp) / nu < batch) ++w;
    while (w > 0 && ipw * (w - 1) + ((w - 1) * sp) / nu >= batch) --w;
    on_spare = batch - ipw * (int)w > 0 ? batch - ipw * (int)w : 0;
    return (int)w;
}
struct FusedPlan {
    // dimensions
    int nx, nu, N, n, X; // n = fullUDim, X = fullXDim
    int batch;
    // costs
    int ncost;
    CostTerm cost[kMaxCosts];
    const double* cost_p[kMaxCosts]; // per-instance references p of cost t: [batch][rows], or nullptr = the shared one
    // sum of the COPRA_COST_DENSE terms (host-evaluated user cost functions): offsets into `params`, -1 = none.
    // Q (n x n, column-major, both triangles), c (n), E (nx x n), f (n)
    int denseQ, densec, denseE, densef;
    int rmax; // max rows over the per-step costs
    // Riccati-factor tier (lmpc_fused_ric.hpp): per-lane stage-cost tables in `params`, built by the plan builder --
    //   [0, 64): Hin entry of the lane | [64, 128): HN entry | then per cost t and row r < 6 one vector of 64: the
    //   coefficient of p_t[r] in the lane's affine entry (of the stage cost OR of the terminal cost: no lane has both) | then
    //   three vectors of 64: Hin again, in the accumulator layout of the MFMA sweep (row blocks 0, 1, 2).  -1: none.
    int ric_tab;
    // Shared-model mode of that tier (copra_batch_set_shared_system): the stage records do not depend on x0, so ONE prepare
    // launch sweeps (ric_model_out, instance dump_instance) and every instance of the batch copies the result (ric_model):
    //   records [N x RicRec::SZ] + constant block | kv [N x nu] | G [N x nx x nu] | row norms [mgen] | A [nx x nx]      (ric_model_offsets below)
    const double* ric_model;
    double* ric_model_out;
    int rfull; // max rows over the full-size costs (0 if none)
    int stage_refs; // 1: some cost follows a reference trajectory (CostTerm::pstride): only the kernels that evaluate costs step by step
                    // with the reference of the step (lmpc_fused.hpp and its packed builds) take the controller
    // One-instance-per-LANE pass in front of the Riccati-factor tier (lmpc_lane.hpp): tables in `params` (-1: the controller is not
    // eligible): H | h | HN | hN | (N + 1) x lane_rps rows [E | G | f], offsets from lane_tab_offsets().  The pass appends every instance
    // it does not finish to lane_list (lane_count entries; lane_zero: the next solve's counter, zeroed on the way); the first tier
    // then runs with lane_from_list = 1: workgroup w takes instance lane_list[w], workgroups beyond the count leave at once.
    int lane_tab, lane_rps;
    int lane_axes; // 1: the stage and terminal cost couple no two AXES -- state i belongs to axis i % nu, control c to axis c (nx a multiple of nu: the
                   // double integrators in nu dimensions, the CoM model).  A wave whose systems couple none either (checked by the pass) has gains
                   // K(c, j) = 0 for j % nu != c at every stage, exactly: the whole recursion stays axis by axis.  Those entries are neither written nor read.
    int lane_cref; // offset (doubles from lane_tab) of the reference coefficients: [cost][row (6)][nz + nx]
    int lane_tlds; // > 0: that many doubles of tables -- the rows of every step, then ub and lb -- sit in LDS behind H | h (the pass reads them there
                   // instead of through scalar loads: three round trips per stage less); 0: they do not fit next to four waves' staging areas
    int lane_bp; // columns of a workspace row: the batch rounded up to whole waves, + 64 spare ones (what lanes without an instance write)
    int lane_from_list;
    int lane_cap; // > 0 (with lane_from_list): the first tier was launched for the first lane_cap entries of the list only, entry w by workgroup w; the
                  // second launch (the tier-2 kernel) walks the rest -- usually none (copra_hip.hip: the grid follows the last solves' list lengths)
    int lane_rest; // >= 0: the tier-2 launch walks the entries of lane_list from this one on (lane_cap, or 0 where no first tier was launched at all); -1: none
    int axis_waves; // waves of its launch (axis_grid below)
    int axis_pf; // > 0: a wave touches the systems of the wave that many further on (the one that follows it on its SIMD): lmpc_axis.hpp
    int* seen_out; // non-null: the tier-2 launch leaves the lengths of this solve's lists in pinned host memory -- [0] <- *seen_src0, [2] <- *seen_src1
    const int *seen_src0, *seen_src1; //   (copra_hip.hip: the grids of the next solves follow them; two 4-byte copies in the stream cost 8 us a step)
    int* axis_acc; // [axis_grid's spare instances]: where the counters of an instance on spare lanes meet (zero between solves)
    const int* axis_list_in; // the second chance (lmpc_axis.hpp, LIST): the list the first launch left, ...
    const int* axis_list_count; // ... its length, ...
    int* axis_count2; // ... and (first launch) the counter of the list the second chance appends to: zeroed on the way
    int axis_const; // 1: its tables are the same at every step (pure state rows present at all N + 1 steps with one E and f and indices affine in the step, one pair of bounds per control): the builds that keep them in registers
    int axis_order; // which order its systems' states are in: 0: state i on axis i % nu (x = (p, v): the benchmark's CoM model), 1: state i on axis i / nxa
                    //   (x = (p_x, v_x, p_y, v_y, ..)) -- seen when the systems are set (plan_builder.hpp: axis_order_of); the tables below are that order's
    int axis_cref; // ... the coefficients of the cost references in its affine terms, per axis: the number of cost rows that look at the axis, then
                   //     kAxisMaxRef entries [cost | row | prows | pstride | offP of the cost | coefficients in h (nxa + 1) | in hN (nxa)] (-1: none, or an axis with more such rows --
                   //     controllers with per-instance references or reference trajectories keep the one-instance-per-lane pass then)
    int axis_tab, axis_rpa; // the (instance, axis)-per-lane solver's tables in `params` (-1: the controller is not eligible) and rows per axis and step (lmpc_axis.hpp)
    int lane_handover; // 1: the first tier takes its stage records from lane_ws instead of sweeping (compact variant of the tier)
    int lane_spec; // 1 (with lane_handover): the pass takes the first step of the active-set iteration itself where a bound on u_0 is the pick (lmpc_lane.hpp)
    double* lane_ws; // [N][lane_ws_rows][lane_bp]: what the sweep leaves per stage, lane-major
    double* lane_ws2; // [batch][lane_ws2_doubles]: the hand-over block of every instance (Lam^-1 | norm sums), instance-major
    int* lane_list;
    int* lane_count;
    int* lane_zero;
    int* lane_hist; // [kLaneHistBins] or nullptr: histogram of the violated-row counts of the instances the pass leaves over (first solve)
    // constraint rows
    int meq, mineq, mgen, mtotal; // mgen = meq + mineq, mtotal = mgen + 2n (QuadProgSolver.cpp:51)
    int any_state_rows; // 1 if any row has a state term (then the trajectory is refreshed before every scan)
    int rows_pure; // 1 if, on top of rows_direct, no row has both a state term and a control term (lmpc_fused_ric.hpp, compact variant)
    int rows_direct; // 1 if every state term is one component of one state (bounds on the trajectory): such a slack is
                     // evaluated straight from G and the iterate, without refreshing the whole trajectory first
    int n_full_rows; // rows with a full-size entry (kEFull / kGFull), listed below; -1: more than kMaxFullRows
    int full_row[kMaxFullRows];
    const int* row_step; // [mgen]
    const int* row_ekind; // [mgen]
    const int* row_eoff; // [mgen]
    const int* row_gkind; // [mgen]
    const int* row_goff; // [mgen]
    const double* row_f; // [mgen]
    const double* row_f_inst; // per-instance right-hand sides [batch][mgen] (copra_batch_set_constraint_rhs) or nullptr
    const double* lb_inst; // per-instance control bounds [batch][n] (copra_batch_set_control_bounds) or nullptr
    const double* ub_inst;
    const double* params; // blob
    const double* lb; // [n]
    const double* ub; // [n]
    // solver constants
    double vsmall; // qpgen2's machine-precision guard
    int max_iter;
    // batch I/O (device pointers in the product, host pointers in the CPU emulator)
    const double* A; // [batch][nx*nx]
    const double* B; // [batch][nx*nu]
    const double* d; // [batch][nx]
    const double* x0; // [batch][nx]
    double* control; // [batch][n]
    double* trajectory; // [batch][X]
    int* status; // [batch]
    int* iter; // [batch][2]
    // InitialStateLMPC variant (include/InitialStateLMPC.h): decision vector [x0; U]
    int initial_state; // 0: LMPC, 1: InitialStateLMPC
    const double* is_R; // nx x nx  (resetInitialStateCost, InitialStateLMPC.cpp:35-40)
    const double* is_r; // nx
    const double* x0lb; // [batch][nx] or nullptr = ps->x0 (InitialStateLMPC.cpp:20-28, 42-46)
    const double* x0ub;
    double* x0_opt; // [batch][nx]  (InitialStateLMPC::initialState())
    IsLayout isl;
    int inst_offset; // instance handled by workgroup 0 (normally 0)
    // optional parity dump (dump_instance >= 0): the dense QP of ONE instance written by the condense code
    int dump_instance;
    int dump_only; // 1: stop after the dump (no solve, no result stores)
    double* dumpQ; // n x n (symmetric, both triangles written)
    double* dumpc; // n
    double* dumpA; // mgen x n  (rows in stacking order, <= / = orientation of the reference)
    double* dumpb; // mgen
    // optional phase profile: 8 shader-clock stamps per instance (preview, costs, norms, cholesky, inverse+x0,
    // active set, results, total) -- the device-side analogue of LMPC::solveTime()/solveAndBuildTime()
    // two-tier execution: instances whose active set outgrows lds.rcap in the compact layout are queued ...
    int* ovf_count; // device counter (zero when the first tier starts)
    int* ovf_zero; // != nullptr: the first tier zeroes this one -- the next solve's counter (copra_hip.hip: begin_overflow_queue)
    int* ovf_list; // [batch] instance ids
    int from_list; // ... and the second launch (full layout) takes its instances from that queue
    // shared-model fast path (lmpc_shared.hpp): the whole batch shares (A, B, d), only x0 differs per instance.
    //   model_out != nullptr : "prepare" launch of the fused kernel -- after the factorisation instance
    //                          `dump_instance` stores J = R^-1, G, Phi, xi and the row norms there and stops
    //   model     != nullptr : lmpc_shared_body reads them (+ c0, C1: c = c0 + C1 x0) instead of rebuilding
    // layout (doubles): status | J [n * ldj] | G [N nx nu] | Phi [(N+1) nx nx] | xi [X] | nb [mgen] | c0 [n] | C1 [n x nx]
    //                   | Qinv [n * ldj] (= J J', symmetric: the unconstrained minimiser is -Qinv c without touching J)
    //                   | Rtri [n (n + 1) / 2], rinv [n] : the packed Cholesky factor and 1 / R(i,i) (factor-only tier)
    //                   | xu0 [n], K1 [n x nx] (and K2 [n x R] after C2): the unconstrained minimiser xu0 + K1 x0 + K2 p
    //                   | C2 [n x R] : dc/dp of the costs that have per-instance references (model_ref_off[t] = first
    //                     column of cost t, -1 = controller-wide reference already folded into c0)
    int model_ref_off[kMaxCosts];
    int model_rtot; // columns of C2 (and of K2, which follows it)
    double* model_out;
    const double* model;
    // warm start of the shared-model path (copra_batch_set_warm_start): the active set each instance ended its previous
    // solve with, already shifted by one step -- [batch][kWarmCap] row indices, -1 = none -- and for every general row the
    // same row one step earlier (-1: none)
    int* warm_set;
    const int* row_prev;
    // more than 64 decision variables: workgroup-per-instance kernel, J / R in the per-workgroup HBM workspace `ws`
    int use_large;
    LargeLayout large;
    double* ws; // [resident workgroups][large.ws_total]
    long long* prof;
    long long* prof_fine; // profiling builds only (-DCOPRA_FINE_PROFILE): 32 raw stamps per instance
    LdsLayout lds;
};

// offsets (doubles) into the shared-model buffer, see FusedPlan::model
struct ModelLayout {
    long long status, J, G, Phi, Xi, nb, c0, C1, Qinv, Rtri, rinv, xu0, K1, C2, total; // (C2, then K2 of the same size, end the buffer)
};
#ifdef __HIPCC__
#define COPRA_HOST_DEVICE __host__ __device__
#else
#define COPRA_HOST_DEVICE
#endif
COPRA_HOST_DEVICE inline ModelLayout model_layout(int nx, int nu, int N, int n, int X, int ldj, int mgen)
{
    ModelLayout m;
    long long o = 0;
#define COPRA_TAKE(field, count)                                                                                      \
    m.field = o;                                                                                                      \
    o += ((long long)(count) + 1) & ~1LL
    COPRA_TAKE(status, 2);
    COPRA_TAKE(J, (long long)n * ldj);
    COPRA_TAKE(G, (long long)N * nx * nu);
    COPRA_TAKE(Phi, (long long)(N + 1) * nx * nx);
    COPRA_TAKE(Xi, X);
    COPRA_TAKE(nb, mgen > 0 ? mgen : 1);
    COPRA_TAKE(c0, n);
    COPRA_TAKE(C1, (long long)n * nx);
    COPRA_TAKE(Qinv, (long long)n * ldj);
    COPRA_TAKE(Rtri, (long long)n * (n + 1) / 2); // the Cholesky factor itself, packed (factor-only first tier)
    COPRA_TAKE(rinv, n); // 1 / R(i, i)
    COPRA_TAKE(xu0, n); // the unconstrained minimiser is affine in (x0, p): x = xu0 + K1 x0 + K2 p with
    COPRA_TAKE(K1, (long long)n * nx); // xu0 = -Qinv c0, K1 = -Qinv C1, K2 = -Qinv C2 (K2 follows C2, same shape)
    m.C2 = o;
#undef COPRA_TAKE
    m.total = o;
    return m;
}

// dense batched QP kernel (plug-in point 1): plain SolverInterface::SI_solve arguments, batch-major
struct DensePlan {
    int n, meq, mineq, mgen, mtotal, batch;
    const double *Q, *c, *Aeq, *beq, *Aineq, *bineq, *XL, *XU;
    double* x;
    int* fail;
    int* iter;
    double vsmall;
    int max_iter;
    LdsLayout lds;
    // n > 64: workgroup-per-problem kernel (qp_dense_large.hpp); J and the factor live in `ws`
    LargeLds llds;
    double* ws; // [gridDim.x][2][n * ld] doubles
};

} // namespace copra_hip
