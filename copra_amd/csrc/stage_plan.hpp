// stage_plan.hpp -- the UNCONDENSED (stage-wise) form of one LMPC / InitialStateLMPC controller, for the Riccati
// interior-point kernel (lmpc_riccati.hpp): SURVEY.md 8(f) rank 3, "structure-exploiting long-horizon solver".
//
// The reference condenses the optimal-control problem into a dense QP in U (src/LMPC.cpp:225-280: n = N nu variables,
// Hessian N nu x N nu) and hands it to a dense active-set solver: O(N^3 nu^3) to factorise, O(N^2 nu^2) per active-set
// iteration.  The SAME problem written over the stage variables z_k = (x_k, u_k) with the dynamics
// x_{k+1} = A x_k + B u_k + d as equality constraints is block-banded:
//
//     min  sum_k  1/2 z_k' W_k z_k + q_k' z_k      s.t.  rows of stage k:  a_i' z_k  <= | =  f_i
//
// and every cost / constraint class of the reference with a per-step entry is exactly of this form:
//   TrajectoryCost / TargetCost / ControlCost / MixedCost  (src/costFunctions.cpp:63-82, 100-108, 139-158, 195-215):
//       1/2 || M x_k + N u_k - p ||^2_W  at the steps the class sums over  ->  cost ROWS  a = [M_i, N_i], weight w_i, target p_i
//   LMPC::updateSystem's 1e-6 I (src/LMPC.cpp:228-230)  ->  1e-6 on the u-u diagonal of every W_k
//   Trajectory / Control / Mixed constraints, TrajectoryBoundConstraint (src/constraints.cpp:66-84, 137-148, 197-226,
//       284-315, quirk Q1 included: the plan's rows are taken as they are)  ->  constraint rows  a = [E_i, G_i]
//   ControlBoundConstraint and the [I; -I] rows of QuadProgSolver.cpp:59-69  ->  unit rows on u_k
//   InitialStateLMPC (src/InitialStateLMPC.cpp:77-122): x_0 is a variable with unit rows x0lb <= x_0 <= x0ub; its
//       objective differs from the stage-wise one only in x_0 (lmpc_riccati.hpp works that term out per instance).
// Full-size entries are stage-wise when every row touches ONE step only (block-diagonal M / E / G: what AutoSpan
// produces, and e.g. a terminal constraint); a row that couples several steps makes the controller ineligible, and so do
// more than kRicMaxEq equality rows (they are handled by a proximal multiplier iteration that suits a few
// well-posed rows, not the hundreds of redundant ones of the reference's EqSystem fixture).  Ineligible controllers
// stay on the condensed Goldfarb-Idnani path.
//
// Pure C++ (no HIP): shared by the C-ABI library and the CPU emulator harness.
#pragma once

#include "plan_builder.hpp"

#include <algorithm>
#include <cstring>

namespace copra_hip {

constexpr int kRicMaxNz = 32; // xDim + uDim the Riccati kernel covers
constexpr int kRicMaxEq = 32; // equality rows (proximal multiplier iteration)
constexpr int kRicMaxStageRows = 192; // rows of one stage (LDS staging of their weights)
constexpr int kRicMaxNnz = 512; // non-zero coefficients of the rows of one stage ...
constexpr int kRicMaxNnzE = 1024; // ... and non-zero products a_i a_j over all its rows (LDS tables of the kernel)
// LDS-resident kernel (lmpc_riccati_mfma.hpp): blocks of the v_mfma_f64_4x4x4 products and the widths of its fixed tables
constexpr int kRfNX = 12, kRfNU = 6, kRfNZ = 18; // padded xDim / uDim: three blocks of four states, two blocks of three controls
constexpr int kRfMR = 15; // 64-lane registers of per-row state: at most 960 rows
constexpr int kRfZR = 15; // 64-lane registers of the stage vectors: (N + 1) 18 <= 960
constexpr int kRfGTerms = 4, kRfWTerms = 4, kRfTTerms = 4, kRfMaxTouched = 64;
constexpr int kRfRing = 4, kRfRingStride = 108; // stage records of the LDS-resident kernel that sit in LDS at a time (round 5: the other N - 4 wait in a per-wave workspace)
constexpr int kRfKStride = 107; // stage record: Ka 3 x 12 | Kb 3 x 12 | Kba 3 x 3 | -Mbb^-1 3 x 3 | -M'aa^-1 3 x 3 | kv_a | kv_b | spare | zero

// where the right-hand side of a constraint row comes from (per instance when the caller set per-instance data)
enum { kSrcRowF = 0, kSrcUb = 1, kSrcNegLb = 2, kSrcX0Ub = 3, kSrcNegX0Lb = 4 };

// device view (all pointers into HBM copies, or host vectors in the emulator)
// (the convergence test of the interior-point kernels, ric_converged, follows the struct)

struct StagePlan {
    int nx, nu, N, nz; // nz = nx + nu
    int m; // constraint rows of one instance (all stages)
    int ncls; // stage classes (stages with identical W and row templates share one)
    int max_stage_rows;
    int max_dense; // most dense rows of one stage
    int x0_free; // InitialStateLMPC
    const int* cls_of_stage; // [N + 1]
    const int* stage_row0; // [N + 2] first row of stage k in the per-instance row arrays
    // per class
    const int* cls_W; // [ncls] offset of W (nz x nz, column-major, symmetric) in `blob`
    const int* cls_crow0; // [ncls + 1] cost rows of the class
    const int* cls_row0; // [ncls + 1] constraint-row templates of the class: dense rows first ...
    const int* cls_ndense; // [ncls] ... this many, unit rows after them
    // cost rows (for q_k = - sum w p a)
    const int* cr_aoff; // offset of a (nz doubles) in `blob`
    const int* cr_cost; // index of the cost (FusedPlan::cost / cost_p)
    const int* cr_pidx; // index into that cost's p vector
    const double* cr_w;
    // constraint-row templates
    const int* r_kind; // 0: dense (nz coefficients at blob[r_aoff]), 1: unit (coefficient r_sign on component r_aoff)
    const int* r_aoff;
    const double* r_sign;
    const int* r_eq; // 1: equality
    const int* r_src; // kSrc*
    const int* r_sidx; // index into the source array ...
    const int* r_sstride; // ... + r_sstride * step
    const double* blob;
    // the rows of a stage class as a sparse matrix A (rows x nz), three ways (offsets into iblob / blob per class):
    //   by row     : a_r' v          = sum_{q in [rptr[r], rptr[r+1])} rval[q] v[rcol[q]]
    //   by column  : (A' c)_i        = sum_{q in [gptr[i], gptr[i+1])} gval[q] c[grow[q]]
    //   by entry   : (A' D A)_(i,j)  = sum_{q in [eptr[e], eptr[e+1])} eval[q] D[erow[q]],  e = i + nz j, eval = a_i a_j
    // bound rows have one coefficient, the reference's mixed rows two: the loops over them have 0 - 3 iterations
    const int* iblob;
    const int *cls_rptr, *cls_rcol, *cls_gptr, *cls_grow, *cls_eptr, *cls_erow; // [ncls] offsets into iblob
    const int *cls_rval, *cls_gval, *cls_eval; // [ncls] offsets into blob
    int max_nnz, max_nnze;
    int* next_instance; // device counter of the work queue (reset before every launch), or nullptr
    // per-resident-wave workspace
    double* ws;
    double* rec_ws; // LDS-resident kernel: [persistent waves][N x kRfKStride] -- the stage records of the instance in flight (lmpc_riccati_mfma.hpp)
    long long ws_total; // doubles per wave
    // workspace offsets (doubles)
    long long oZ, oDZ, oQ, oGB, oF, oS, oLam, oDS, oDL, oRP, oFlag, oK, oMi, oKv, oH0, oG0;
    int lds_doubles;
    int max_iter;
    double step_tol, mu_tol; // convergence (ric_converged below): residuals <= 1e-9, mu <= mu_tol, and step <= step_tol (1 + |z|) or its second way out
    double s_floor, lam0; // starting point of the interior-point iteration: slacks max(f - a'z, s_floor), multipliers lam0
    double delta; // proximal weight of the equality rows
    // ---- LDS-resident kernel (lmpc_riccati_mfma.hpp): fixed-width views over the PADDED stage vector z_k = (x: 12 | u: 6),
    //      entry (k, i) at k * kRfNZ + i; usable when fast_ok (else the streaming kernel of lmpc_riccati.hpp runs) ----
    int fast_ok;
    int fast_lds_doubles;
    const int* f_rinfo; // [m] stage | template << 8 (template: index into the r_* arrays)
    const int* f_rcomp; // [templates][2] padded components of the (at most two) coefficients of a row template, -1: none
    const double* f_rval; // [templates][2]
    const int* f_gcnt; // per (class, padded component): rows of the stage that touch it, at most kRfGTerms ...
    const int* f_grow; // [ncls][kRfNZ][kRfGTerms] ... as row offsets inside the stage
    const double* f_gval; // [ncls][kRfNZ][kRfGTerms]
    const int* f_wcol; // [ncls][kRfNZ][kRfWTerms] non-zeros of row i of the padded W (column, -1: none)
    const double* f_wval;
    const int* f_Wp; // [ncls] offset in `blob` of the padded dense W (18 x 18 column-major, unit diagonal on padded controls)
    const int* f_tptr; // [ncls + 1] touched entries of H = W + A' D A
    const int* f_tent; // entry i + kRfNZ j (padded)
    const int* f_trow; // [touched][kRfTTerms] row offsets inside the stage (-1: none)
    const double* f_tval; // [touched][kRfTTerms] a_i a_j
    const int* f_qcnt; // [ncls] cost rows (for q_k) -- padded coefficient vectors:
    const int* f_qoff; // [ncls] first cost row of the class in cr_*
    const double* f_qa; // [cost rows][kRfNZ] padded coefficient vectors a
    const double* f_q; // [(N + 1) kRfNZ] q_k = - sum_rows w p a with the controller-wide references p (per-instance references: streaming kernel)
    int fast_ntmpl; // row templates (their coefficient pairs sit in LDS)
    int f_q_uniform; // q_k is the same at every stage of a class (per-step references): one load per class instead of one per stage
};

struct HostStagePlan {
    bool eligible = false;
    std::string why; // reason when not eligible
    StagePlan sp {};
    std::vector<int> cls_of_stage, stage_row0, cls_W, cls_crow0, cls_row0, cls_ndense;
    std::vector<int> cr_aoff, cr_cost, cr_pidx;
    std::vector<double> cr_w;
    std::vector<int> r_kind, r_aoff, r_eq, r_src, r_sidx, r_sstride;
    std::vector<double> r_sign;
    std::vector<double> blob;
    std::vector<int> iblob, cls_rptr, cls_rcol, cls_gptr, cls_grow, cls_eptr, cls_erow, cls_rval, cls_gval, cls_eval;
    bool all_bounds = false; // bound rows for every control (per-instance bounds may make any of them finite)
    // LDS-resident kernel
    std::string fast_why; // reason when not fast_ok
    std::vector<int> f_rinfo, f_rcomp, f_gcnt, f_grow, f_wcol, f_Wp, f_tptr, f_tent, f_trow, f_qcnt, f_qoff;
    std::vector<double> f_rval, f_gval, f_wval, f_tval, f_qa, f_q;
};

namespace stage_detail {
    struct CostRow {
        std::vector<double> a;
        double w;
        int cost, pidx;
    };
    struct Row {
        int kind; // 0 dense, 1 unit
        std::vector<double> a; // dense
        int comp;
        double sign;
        int eq, src, sidx, sstride;
    };
    struct StageDesc {
        std::vector<CostRow> crows;
        std::vector<Row> rows;
        bool last;
    };
    inline bool same(const StageDesc& x, const StageDesc& y)
    {
        if (x.last != y.last || x.crows.size() != y.crows.size() || x.rows.size() != y.rows.size()) return false;
        for (size_t i = 0; i < x.crows.size(); ++i) {
            const CostRow &a = x.crows[i], &b = y.crows[i];
            if (a.a != b.a || a.w != b.w || a.cost != b.cost || a.pidx != b.pidx) return false;
        }
        for (size_t i = 0; i < x.rows.size(); ++i) {
            const Row &a = x.rows[i], &b = y.rows[i];
            // (sidx may differ between stages of one class: it advances by sstride per step from the class's first stage)
            if (a.kind != b.kind || a.a != b.a || a.comp != b.comp || a.sign != b.sign || a.eq != b.eq || a.src != b.src
                || (a.src != kSrcRowF && a.sstride != b.sstride))
                return false;
        }
        return true;
    }
} // namespace stage_detail

// Build the stage plan of the controller `hp` (after build_plan).  all_bounds: emit both bound rows of every control
// even where the controller-wide bound is infinite (needed once per-instance bounds are set).
inline void build_stage_plan(const HostPlan& hp, HostStagePlan& out, bool all_bounds)
{
    using namespace stage_detail;
    const FusedPlan& P = hp.plan;
    const int nx = P.nx, nu = P.nu, N = P.N, nz = nx + nu, X = P.X, U = P.n;
    out = HostStagePlan();
    out.all_bounds = all_bounds;
    auto no = [&](const char* why) {
        out.eligible = false;
        out.why = why;
    };
    if (nz > kRicMaxNz) return no("xDim + uDim too large for the Riccati kernel");
    if (nu > 8) return no("uDim > 8");
    if (P.meq > kRicMaxEq) return no("too many equality rows for the proximal multiplier iteration");
    if (P.denseQ >= 0) return no("a dense (host-evaluated) cost couples all steps");
    std::vector<StageDesc> st((size_t)N + 1);
    for (int k = 0; k <= N; ++k) st[(size_t)k].last = (k == N);
    const std::vector<double>& prm = hp.params;
    // a full-size coefficient row touches at most one step?  -> (step, per-step slice); nothing set: step -1
    auto one_step = [&](const double* row, int per, int steps, int& step) {
        step = -1;
        for (int s = 0; s < steps; ++s)
            for (int j = 0; j < per; ++j)
                if (row[(size_t)s * per + j] != 0.0) {
                    if (step >= 0 && step != s) return false;
                    step = s;
                }
        return true;
    };
    // ---- costs ----
    for (int t = 0; t < P.ncost; ++t) {
        const CostTerm& c = P.cost[t];
        const bool hasM = c.offM >= 0, hasN = c.offN >= 0;
        if (!c.full) {
            int k0 = 0, k1 = N; // steps the class sums over
            if (c.kind == kCostTarget) k0 = N; // costFunctions.cpp:100-108
            if (c.kind == kCostControl || c.kind == kCostMixed) k1 = N - 1; // :139-158, :195-215
            for (int k = k0; k <= k1; ++k)
                for (int i = 0; i < c.rows; ++i) {
                    CostRow r { std::vector<double>((size_t)nz, 0.0), prm[(size_t)c.offW + i], t, (c.pstride ? k * c.pstride : 0) + i }; // (a reference trajectory: the reference of step k)
                    if (hasM && c.kind != kCostControl)
                        for (int j = 0; j < nx; ++j) r.a[(size_t)j] = prm[(size_t)c.offM + (size_t)j * c.rows + i];
                    if (hasN && (c.kind == kCostControl || c.kind == kCostMixed))
                        for (int j = 0; j < nu; ++j) r.a[(size_t)nx + j] = prm[(size_t)c.offN + (size_t)j * c.rows + i];
                    st[(size_t)k].crows.push_back(r);
                }
        } else { // full-size entry: row-major M (rows x X), N (rows x U)
            for (int i = 0; i < c.rows; ++i) {
                int sx = -1, su = -1;
                if (hasM && !one_step(&prm[(size_t)c.offM + (size_t)i * X], nx, N + 1, sx))
                    return no("a full-size cost row couples several steps");
                if (hasN && !one_step(&prm[(size_t)c.offN + (size_t)i * U], nu, N, su))
                    return no("a full-size cost row couples several steps");
                if (sx >= 0 && su >= 0 && sx != su) return no("a full-size cost row couples several steps");
                const int s = sx >= 0 ? sx : su;
                CostRow r { std::vector<double>((size_t)nz, 0.0), prm[(size_t)c.offW + i], t, i };
                if (sx >= 0)
                    for (int j = 0; j < nx; ++j) r.a[(size_t)j] = prm[(size_t)c.offM + (size_t)i * X + (size_t)sx * nx + j];
                if (su >= 0)
                    for (int j = 0; j < nu; ++j) r.a[(size_t)nx + j] = prm[(size_t)c.offN + (size_t)i * U + (size_t)su * nu + j];
                // (an all-zero row still carries  1/2 w p^2 : a constant, dropped)
                if (s >= 0) st[(size_t)s].crows.push_back(r);
            }
        }
    }
    // ---- general constraint rows, in the plan's stacked order ----
    for (int i = 0; i < P.mgen; ++i) {
        const int ek = hp.row_ekind[(size_t)i], gk = hp.row_gkind[(size_t)i];
        Row r { 0, std::vector<double>((size_t)nz, 0.0), 0, 1.0, i < P.meq ? 1 : 0, kSrcRowF, i, 0 };
        int step = hp.row_step[(size_t)i];
        int sx = -1, su = -1;
        if (ek == kEDense)
            for (int j = 0; j < nx; ++j) r.a[(size_t)j] = prm[(size_t)hp.row_eoff[(size_t)i] + j];
        else if (e_onehot(ek))
            r.a[(size_t)hp.row_eoff[(size_t)i]] = e_sign(ek);
        else if (ek == kEFull) {
            const double* row = &prm[(size_t)hp.row_eoff[(size_t)i]];
            if (!one_step(row, nx, N + 1, sx)) return no("a full-size constraint row couples several steps");
            if (sx >= 0)
                for (int j = 0; j < nx; ++j) r.a[(size_t)j] = row[(size_t)sx * nx + j];
        }
        if (gk == kGStep)
            for (int j = 0; j < nu; ++j) r.a[(size_t)nx + j] = prm[(size_t)hp.row_goff[(size_t)i] + j];
        else if (gk == kGFull) {
            const double* row = &prm[(size_t)hp.row_goff[(size_t)i]];
            if (!one_step(row, nu, N, su)) return no("a full-size constraint row couples several steps");
            if (su >= 0)
                for (int j = 0; j < nu; ++j) r.a[(size_t)nx + j] = row[(size_t)su * nu + j];
        }
        if (ek == kEFull || gk == kGFull) {
            if (sx >= 0 && su >= 0 && sx != su) return no("a full-size constraint row couples several steps");
            step = sx >= 0 ? sx : (su >= 0 ? su : 0);
        }
        if (step == N) // no u_N: a control term there cannot exist (the reference's classes never produce one)
            for (int j = 0; j < nu; ++j)
                if (r.a[(size_t)nx + j] != 0.0) return no("constraint row on u_N");
        // a row with exactly one coefficient of +-1 is a unit row (bounds on a state / control component)
        int nnz = 0, comp = -1;
        for (int j = 0; j < nz; ++j)
            if (r.a[(size_t)j] != 0.0) ++nnz, comp = j;
        if (nnz == 1 && (r.a[(size_t)comp] == 1.0 || r.a[(size_t)comp] == -1.0)) {
            r.kind = 1;
            r.comp = comp;
            r.sign = r.a[(size_t)comp];
            r.a.clear();
        }
        st[(size_t)step].rows.push_back(r);
    }
    // ---- bounds on U: rows [I; -I] with [XU; -XL] (QuadProgSolver.cpp:59-69) ----
    for (int k = 0; k < N; ++k)
        for (int j = 0; j < nu; ++j) {
            const double lo = hp.lb[(size_t)k * nu + j], up = hp.ub[(size_t)k * nu + j];
            if (all_bounds || (up < DBL_MAX && !is_pos_inf(up)))
                st[(size_t)k].rows.push_back(Row { 1, {}, nx + j, 1.0, 0, kSrcUb, j, nu });
            if (all_bounds || (lo > -DBL_MAX && !is_neg_inf(lo)))
                st[(size_t)k].rows.push_back(Row { 1, {}, nx + j, -1.0, 0, kSrcNegLb, j, nu });
        }
    // ---- InitialStateLMPC: bounds on x_0 (InitialStateLMPC.cpp:119-121) ----
    if (P.initial_state)
        for (int j = 0; j < nx; ++j) {
            st[0].rows.push_back(Row { 1, {}, j, 1.0, 0, kSrcX0Ub, j, 0 });
            st[0].rows.push_back(Row { 1, {}, j, -1.0, 0, kSrcNegX0Lb, j, 0 });
        }
    // dense rows first, unit rows after (the kernel treats the two groups in separate loops)
    for (int k = 0; k <= N; ++k)
        std::stable_partition(st[(size_t)k].rows.begin(), st[(size_t)k].rows.end(), [](const Row& r) { return r.kind == 0; });
    // ---- classes: stages with the same cost rows and the same row templates share one.  The right-hand sides of the
    //      general rows are addressed by their position in the plan's stacked order, which advances by a fixed stride
    //      from stage to stage inside a class (fixed when the class gets its second member) ----
    std::vector<int> first_stage; // of each class
    std::vector<char> stride_set;
    out.cls_of_stage.assign((size_t)N + 1, -1);
    for (int k = 0; k <= N; ++k) {
        int c = -1;
        for (size_t q = 0; q < first_stage.size() && c < 0; ++q) {
            const int k0 = first_stage[q];
            if (!same(st[(size_t)k0], st[(size_t)k])) continue;
            bool ok = true;
            for (size_t i = 0; i < st[(size_t)k].rows.size() && ok; ++i) {
                const Row &a = st[(size_t)k0].rows[i], &b = st[(size_t)k].rows[i];
                if (a.src != kSrcRowF) {
                    ok = a.sidx == b.sidx;
                    continue;
                }
                const int dd = b.sidx - a.sidx;
                ok = stride_set[q] ? (dd == a.sstride * (k - k0)) : (dd % (k - k0) == 0);
            }
            if (!ok) continue;
            if (!stride_set[q]) {
                for (size_t i = 0; i < st[(size_t)k].rows.size(); ++i) {
                    Row& a = st[(size_t)k0].rows[i];
                    if (a.src == kSrcRowF) a.sstride = (st[(size_t)k].rows[i].sidx - a.sidx) / (k - k0);
                }
                stride_set[q] = 1;
            }
            c = (int)q;
        }
        if (c < 0) {
            c = (int)first_stage.size();
            first_stage.push_back(k);
            stride_set.push_back(0);
        }
        out.cls_of_stage[(size_t)k] = c;
    }
    const int ncls = (int)first_stage.size();
    auto push_blob = [&](const std::vector<double>& v) {
        const int at = (int)out.blob.size();
        out.blob.insert(out.blob.end(), v.begin(), v.end());
        if (out.blob.size() & 1) out.blob.push_back(0.0);
        return at;
    };
    int max_rows = 0, max_dense = 0, max_nnz = 0, max_nnze = 0;
    out.cls_crow0.push_back(0);
    out.cls_row0.push_back(0);
    for (int c = 0; c < ncls; ++c) {
        const int k0 = first_stage[(size_t)c];
        const StageDesc& S = st[(size_t)k0];
        std::vector<double> W((size_t)nz * nz, 0.0);
        for (const CostRow& r : S.crows)
            for (int j = 0; j < nz; ++j)
                for (int i = 0; i < nz; ++i) W[(size_t)j * nz + i] += r.w * r.a[(size_t)i] * r.a[(size_t)j];
        if (!S.last)
            for (int j = nx; j < nz; ++j) W[(size_t)j * nz + j] += 1e-6; // LMPC.cpp:228-230
        out.cls_W.push_back(push_blob(W));
        for (const CostRow& r : S.crows) {
            out.cr_aoff.push_back(push_blob(r.a));
            out.cr_cost.push_back(r.cost);
            out.cr_pidx.push_back(r.pidx);
            out.cr_w.push_back(r.w);
        }
        out.cls_crow0.push_back((int)out.cr_aoff.size());
        for (const Row& r : S.rows) {
            out.r_kind.push_back(r.kind);
            out.r_aoff.push_back(r.kind == 0 ? push_blob(r.a) : r.comp);
            out.r_sign.push_back(r.sign);
            out.r_eq.push_back(r.eq);
            out.r_src.push_back(r.src);
            // index of the class's FIRST stage minus stride * k0, so that the kernel computes sidx + sstride * k
            out.r_sidx.push_back(r.sidx - r.sstride * (r.src == kSrcRowF ? k0 : 0));
            out.r_sstride.push_back(r.sstride);
        }
        out.cls_row0.push_back((int)out.r_kind.size());
        {
            int nd = 0;
            for (const Row& r : S.rows) nd += r.kind == 0 ? 1 : 0;
            out.cls_ndense.push_back(nd);
        }
        if ((int)S.rows.size() > max_rows) max_rows = (int)S.rows.size();
        if (out.cls_ndense.back() > max_dense) max_dense = out.cls_ndense.back();
        { // the three sparse views of the class's rows
            const int nr = (int)S.rows.size();
            std::vector<std::vector<std::pair<int, double>>> byrow((size_t)nr), bycol((size_t)nz);
            std::vector<std::vector<std::pair<int, double>>> byent((size_t)nz * nz);
            for (int r = 0; r < nr; ++r) {
                const Row& R = S.rows[(size_t)r];
                for (int j = 0; j < nz; ++j) {
                    const double v = R.kind == 0 ? R.a[(size_t)j] : (j == R.comp ? R.sign : 0.0);
                    if (v == 0.0) continue;
                    byrow[(size_t)r].push_back({ j, v });
                    bycol[(size_t)j].push_back({ r, v });
                }
                for (auto& a : byrow[(size_t)r])
                    for (auto& b : byrow[(size_t)r]) byent[(size_t)a.first + (size_t)nz * b.first].push_back({ r, a.second * b.second });
            }
            auto emit = [&](const std::vector<std::vector<std::pair<int, double>>>& lists, std::vector<int>& cptr,
                            std::vector<int>& cidx, std::vector<int>& cval) {
                cptr.push_back((int)out.iblob.size());
                int run = 0;
                for (auto& l : lists) {
                    out.iblob.push_back(run);
                    run += (int)l.size();
                }
                out.iblob.push_back(run);
                cidx.push_back((int)out.iblob.size());
                cval.push_back((int)out.blob.size());
                for (auto& l : lists)
                    for (auto& e : l) {
                        out.iblob.push_back(e.first);
                        out.blob.push_back(e.second);
                    }
                if (out.blob.size() & 1) out.blob.push_back(0.0);
                return run;
            };
            const int nnz = emit(byrow, out.cls_rptr, out.cls_rcol, out.cls_rval);
            emit(bycol, out.cls_gptr, out.cls_grow, out.cls_gval);
            const int nnze = emit(byent, out.cls_eptr, out.cls_erow, out.cls_eval);
            if (nnz > max_nnz) max_nnz = nnz;
            if (nnze > max_nnze) max_nnze = nnze;
        }
    }
    if (max_nnz > kRicMaxNnz || max_nnze > kRicMaxNnzE) return no("the rows of one stage are too dense for the kernel's LDS tables");
    if (max_rows > kRicMaxStageRows) return no("too many constraint rows in one stage");
    out.stage_row0.assign((size_t)N + 2, 0);
    for (int k = 0; k <= N; ++k)
        out.stage_row0[(size_t)k + 1] = out.stage_row0[(size_t)k] + (int)st[(size_t)k].rows.size();
    StagePlan& sp = out.sp;
    sp.nx = nx, sp.nu = nu, sp.N = N, sp.nz = nz;
    sp.m = out.stage_row0[(size_t)N + 1];
    sp.ncls = ncls;
    sp.max_stage_rows = max_rows;
    sp.max_dense = max_dense;
    sp.max_nnz = max_nnz;
    sp.max_nnze = max_nnze;
    sp.x0_free = P.initial_state;
    sp.max_iter = 60;
    // starting point: slacks s = max(f - a'z_0, 0.05) at the roll-out z_0 of u = 0 and multipliers lam = mean(s) / s -- every
    // complementarity product starts at the mean slack (a CENTRED start at the problem's own scale).  The first version
    // (s = max(., 1), lam = 1) made every row with a slack below 1 start with a primal residual, and the first five Newton steps were
    // spent recovering from that: 19 -> 14 steps on config 5.  (lam = 1 / s -- products of 1 whatever the scale -- does the same for
    // config 5 but takes 27 steps instead of 8 on the reference's falling-mass fixtures, whose bounds are 200.)  The factor on the mean
    // slack, |lam0| = 10: config 5 takes 13.3 / 13.5 / 13.4 / 13.6 steps at 1 / 3 / 10 / 30 (full batch, profiles/r03/lam0_scan.log);
    // the trajectory-cost fixtures (weights of 1e4 -- large multipliers at the optimum) 22 / 19 / 15 / 12.
    sp.s_floor = 0.05, sp.lam0 = -10.0;
    // (step_tol: 1e-10 until round 4 -- an iterate whose last step was 1e-9 went on, and the next factorisation, at multipliers / slacks
    //  of 1e15, broke down: the instance went to the Goldfarb-Idnani kernel for nothing)
    // (mu_tol: 1e-8 until round 4 -- a weakly active row, multiplier 1e-3, then keeps a slack of 1e-5: 4e-6 in U on a random controller)
    sp.step_tol = 1e-8, sp.mu_tol = 1e-11;
    if (hp.opt.ric_step_tol != 0.0) sp.step_tol = hp.opt.ric_step_tol; // (copra_options_t: experiments)
    if (hp.opt.ric_mu_tol != 0.0) sp.mu_tol = hp.opt.ric_mu_tol;
    // proximal weight of the equality rows.  1e-9 until round 4: the multiplier of such a row is (a'z - f + a'dz) / delta, a difference of
    // O(|z|) quantities -- rounding at 1e-13 divided by 1e-9 left the multiplier, and with it U, uncertain at the 1e-4 level whenever the
    // row's right-hand side is not tiny (config 5 pins terminal velocities to ZERO, where that noise is 1e-18; a random controller pins a
    // state to 0.3: found by tests/random_controllers.py).  At 1e-6 the proximal iteration still contracts by 1e-6 x (curvature along the
    // row) per Newton step -- config 5 takes the same 13.4 steps -- and the noise is 1e-7 x smaller than the tolerance of the parity tests.
    sp.delta = 1e-6;
    // workspace of one resident wave
    long long o = 0;
    auto take = [&](long long count) {
        const long long at = o;
        o += (count + 7) & ~7LL;
        return at;
    };
    const long long NZ = (long long)(N + 1) * nz, m = sp.m > 0 ? sp.m : 1;
    sp.oZ = take(NZ), sp.oDZ = take(NZ), sp.oQ = take(NZ), sp.oGB = take(NZ);
    sp.oF = take(m), sp.oS = take(m), sp.oLam = take(m), sp.oDS = take(m), sp.oDL = take(m), sp.oRP = take(m), sp.oFlag = take(m);
    sp.oK = take((long long)N * nu * nx), sp.oMi = take((long long)N * nu * nu), sp.oKv = take((long long)N * nu);
    sp.oH0 = take((long long)nx * nx), sp.oG0 = take(nx);
    sp.ws_total = o;
    // LDS of one wave (doubles): AB | P | T | M | pv, h, g, zk, dzk, dxn, d | K | Mi | row weights / coefficients
    {
        const int mr = max_rows > 0 ? max_rows : 1, nn = max_nnz > 0 ? max_nnz : 1, ne = max_nnze > 0 ? max_nnze : 1;
        const int ints = (mr + 1) + nn + (nz + 1) + nn + (nz * nz + 1) + ne; // rptr rcol gptr grow eptr erow
        sp.lds_doubles = align2(nx * nz) + align2(nx * nx) + align2(nx * nz > 2 * nu * nu ? nx * nz : 2 * nu * nu) + align2(nz * nz)
            + 7 * align2(nz) + align2(nu * nx) + align2(nu * nu) + 2 * align2(mr) + align2(nz * nz) + 2 * align2(nn) + align2(ne)
            + align2((ints + 1) / 2) + 2;
    } // == carve_riccati
    out.eligible = true;
    // ---- fixed-width tables of the LDS-resident kernel ----
    sp.fast_ok = 0;
    auto slow = [&](const char* why) { out.fast_why = why; };
    if (nx > kRfNX || nu > kRfNU) return slow("xDim > 12 or uDim > 6");
    if (sp.m > 64 * kRfMR) return slow("more than 960 constraint rows");
    if ((N + 1) * kRfNZ > 64 * kRfZR) return slow("more than 52 steps");
    if ((int)out.r_kind.size() > 128) return slow("more than 128 row templates");
    auto pad = [&](int j) { return j < nx ? j : kRfNX + (j - nx); };
    const int ntmpl = (int)out.r_kind.size();
    out.f_rcomp.assign((size_t)ntmpl * 2, -1);
    out.f_rval.assign((size_t)ntmpl * 2, 0.0);
    out.f_gcnt.assign((size_t)ncls * kRfNZ, 0);
    out.f_grow.assign((size_t)ncls * kRfNZ * kRfGTerms, 0);
    out.f_gval.assign((size_t)ncls * kRfNZ * kRfGTerms, 0.0);
    out.f_wcol.assign((size_t)ncls * kRfNZ * kRfWTerms, -1);
    out.f_wval.assign((size_t)ncls * kRfNZ * kRfWTerms, 0.0);
    out.f_tptr.assign(1, 0);
    for (int c = 0; c < ncls; ++c) {
        const StageDesc& Sd = st[(size_t)first_stage[(size_t)c]];
        const int nr = (int)Sd.rows.size();
        std::vector<std::vector<std::pair<int, double>>> coef((size_t)nr); // padded (component, value) of every row
        for (int r = 0; r < nr; ++r) {
            const Row& R = Sd.rows[(size_t)r];
            for (int j = 0; j < nz; ++j) {
                const double v = R.kind == 0 ? R.a[(size_t)j] : (j == R.comp ? R.sign : 0.0);
                if (v != 0.0) coef[(size_t)r].push_back({ pad(j), v });
            }
            if (coef[(size_t)r].size() > 2) return slow("a constraint row with more than two coefficients");
            const int t = out.cls_row0[(size_t)c] + r;
            for (size_t q = 0; q < coef[(size_t)r].size(); ++q) {
                out.f_rcomp[(size_t)t * 2 + q] = coef[(size_t)r][q].first;
                out.f_rval[(size_t)t * 2 + q] = coef[(size_t)r][q].second;
                int& cnt = out.f_gcnt[(size_t)c * kRfNZ + coef[(size_t)r][q].first];
                if (cnt >= kRfGTerms) return slow("a component with more than four constraint rows in one stage");
                const size_t at = ((size_t)c * kRfNZ + coef[(size_t)r][q].first) * kRfGTerms + cnt;
                out.f_grow[at] = r;
                out.f_gval[at] = coef[(size_t)r][q].second;
                cnt += 1;
            }
        }
        // padded dense W (the kernel's copy of the stage Hessian starts from it) and its non-zeros row by row
        std::vector<double> Wp((size_t)kRfNZ * kRfNZ, 0.0);
        const double* Wc = &out.blob[(size_t)out.cls_W[(size_t)c]];
        for (int j = 0; j < nz; ++j)
            for (int i = 0; i < nz; ++i) Wp[(size_t)pad(j) * kRfNZ + pad(i)] = Wc[(size_t)j * nz + i];
        for (int j = nu; j < kRfNU; ++j) Wp[(size_t)(kRfNX + j) * kRfNZ + kRfNX + j] = 1.0; // (controls that do not exist stay zero)
        for (int i = 0; i < kRfNZ; ++i) {
            int cnt = 0;
            for (int j = 0; j < kRfNZ; ++j) {
                const double v = Wp[(size_t)j * kRfNZ + i];
                if (v == 0.0 || (i >= kRfNX + nu && i == j)) continue; // (the padded unit diagonal multiplies zeros)
                if (cnt >= kRfWTerms) return slow("a stage cost with more than four entries per row");
                out.f_wcol[((size_t)c * kRfNZ + i) * kRfWTerms + cnt] = j;
                out.f_wval[((size_t)c * kRfNZ + i) * kRfWTerms + cnt] = v;
                cnt += 1;
            }
        }
        out.f_Wp.push_back(push_blob(Wp));
        // entries of H = W + sum_r D_r a_r a_r' that the rows touch
        std::vector<std::vector<std::pair<int, double>>> ent((size_t)kRfNZ * kRfNZ);
        for (int r = 0; r < nr; ++r)
            for (auto& a : coef[(size_t)r])
                for (auto& b : coef[(size_t)r]) ent[(size_t)a.first + (size_t)kRfNZ * b.first].push_back({ r, a.second * b.second });
        int touched = 0;
        for (int e = 0; e < kRfNZ * kRfNZ; ++e) {
            if (ent[(size_t)e].empty()) continue;
            if ((int)ent[(size_t)e].size() > kRfTTerms) return slow("an entry of the stage Hessian with more than four row products");
            out.f_tent.push_back(e);
            for (int q = 0; q < kRfTTerms; ++q) {
                const bool on = q < (int)ent[(size_t)e].size();
                out.f_trow.push_back(on ? ent[(size_t)e][(size_t)q].first : -1);
                out.f_tval.push_back(on ? ent[(size_t)e][(size_t)q].second : 0.0);
            }
            touched += 1;
        }
        if (touched > kRfMaxTouched) return slow("the rows of one stage touch more than 64 entries of the stage Hessian");
        out.f_tptr.push_back((int)out.f_tent.size());
        // cost rows with padded coefficient vectors
        out.f_qoff.push_back(out.cls_crow0[(size_t)c]);
        out.f_qcnt.push_back(out.cls_crow0[(size_t)c + 1] - out.cls_crow0[(size_t)c]);
    }
    out.f_qa.assign(out.cr_aoff.size() * (size_t)kRfNZ, 0.0);
    for (size_t t = 0; t < out.cr_aoff.size(); ++t)
        for (int j = 0; j < nz; ++j) out.f_qa[t * kRfNZ + (size_t)pad(j)] = out.blob[(size_t)out.cr_aoff[t] + j];
    out.f_rinfo.assign((size_t)(sp.m > 0 ? sp.m : 1), 0);
    for (int k = 0; k <= N; ++k) {
        const int c = out.cls_of_stage[(size_t)k];
        for (int r = 0; r < (int)st[(size_t)k].rows.size(); ++r)
            out.f_rinfo[(size_t)out.stage_row0[(size_t)k] + r] = k | ((out.cls_row0[(size_t)c] + r) << 8);
    }
    if (N > 255) return slow("more than 255 steps");
    out.f_q.assign((size_t)(N + 1) * kRfNZ, 0.0);
    for (int k = 0; k <= N; ++k) {
        const int c = out.cls_of_stage[(size_t)k];
        for (int t = out.cls_crow0[(size_t)c]; t < out.cls_crow0[(size_t)c + 1]; ++t) {
            const int ct = out.cr_cost[(size_t)t];
            const double pv = prm[(size_t)P.cost[ct].offP + out.cr_pidx[(size_t)t]];
            for (int i = 0; i < kRfNZ; ++i) out.f_q[(size_t)k * kRfNZ + i] -= out.cr_w[(size_t)t] * pv * out.f_qa[(size_t)t * kRfNZ + i];
        }
    }
    sp.f_q_uniform = 1;
    for (int k = 0; k <= N && sp.f_q_uniform; ++k) {
        const int k0 = first_stage[(size_t)out.cls_of_stage[(size_t)k]];
        for (int i = 0; i < kRfNZ; ++i)
            if (out.f_q[(size_t)k * kRfNZ + i] != out.f_q[(size_t)k0 * kRfNZ + i]) sp.f_q_uniform = 0;
    }
    sp.fast_ntmpl = ntmpl > 0 ? ntmpl : 1;
    if (ntmpl > 128) return slow("more than 128 row templates");
    // LDS of one instance (doubles): X | Y | F (64 kRfMR each) | a ring of kRfRing stage records | kv | H | P | Rb | Ra | H0 | G0 | A B | d | Gauss-Jordan
    sp.fast_lds_doubles = 3 * 64 * kRfMR + kRfRing * kRfRingStride + 2 * (kRfNZ * kRfNZ + 2) + kRfNX * kRfNX + 64 + 64 + kRfNX * kRfNX + 16
        + kRfNX * kRfNZ + 16 + kRfNX * (kRfNX + 1) + 12 + 12 + 32 + 40 + 2 * sp.fast_ntmpl; // == carve_rf
    if ((size_t)sp.fast_lds_doubles * sizeof(double) > 80 * 1024) return slow("LDS footprint above 80 KiB (two instances per CU)");
    sp.fast_ok = 1;
}

inline void point_stage_plan_to_host(HostStagePlan& h)
{
    StagePlan& sp = h.sp;
    sp.cls_of_stage = h.cls_of_stage.data();
    sp.stage_row0 = h.stage_row0.data();
    sp.cls_W = h.cls_W.data();
    sp.cls_crow0 = h.cls_crow0.data();
    sp.cls_row0 = h.cls_row0.data();
    sp.cls_ndense = h.cls_ndense.data();
    sp.cr_aoff = h.cr_aoff.data();
    sp.cr_cost = h.cr_cost.data();
    sp.cr_pidx = h.cr_pidx.data();
    sp.cr_w = h.cr_w.data();
    sp.r_kind = h.r_kind.data();
    sp.r_aoff = h.r_aoff.data();
    sp.r_sign = h.r_sign.data();
    sp.r_eq = h.r_eq.data();
    sp.r_src = h.r_src.data();
    sp.r_sidx = h.r_sidx.data();
    sp.r_sstride = h.r_sstride.data();
    sp.blob = h.blob.data();
    sp.iblob = h.iblob.data();
    sp.cls_rptr = h.cls_rptr.data(), sp.cls_rcol = h.cls_rcol.data(), sp.cls_gptr = h.cls_gptr.data(), sp.cls_grow = h.cls_grow.data();
    sp.cls_eptr = h.cls_eptr.data(), sp.cls_erow = h.cls_erow.data();
    sp.cls_rval = h.cls_rval.data(), sp.cls_gval = h.cls_gval.data(), sp.cls_eval = h.cls_eval.data();
    sp.f_rinfo = h.f_rinfo.data(), sp.f_rcomp = h.f_rcomp.data(), sp.f_rval = h.f_rval.data();
    sp.f_gcnt = h.f_gcnt.data(), sp.f_grow = h.f_grow.data(), sp.f_gval = h.f_gval.data();
    sp.f_wcol = h.f_wcol.data(), sp.f_wval = h.f_wval.data(), sp.f_Wp = h.f_Wp.data();
    sp.f_tptr = h.f_tptr.data(), sp.f_tent = h.f_tent.data(), sp.f_trow = h.f_trow.data(), sp.f_tval = h.f_tval.data();
    sp.f_qcnt = h.f_qcnt.data(), sp.f_qoff = h.f_qoff.data(), sp.f_qa = h.f_qa.data(), sp.f_q = h.f_q.data();
}

} // namespace copra_hip
