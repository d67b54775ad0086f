/*
 * copra_hip.h -- C ABI of the MI355X-native batched condensed linear-MPC engine (libcopra_hip.so).
 *
 * This is the drop-in boundary BENEATH copra's C++ classes.  The reference (jrl-umi3218/copra v1.3.3) has no C
 * ABI of its own; each entry point below names the reference interface it replaces (file:line relative to the
 * reference tree).  Plain pointers and sizes only -- no torch / Eigen types.  All matrices are FP64 and
 * column-major per instance (Eigen's default layout, include/PreviewSystem.h:56-62); batched arrays are
 * batch-major: element (b, i, j) of a [batch][rows x cols] array lives at  ptr[b*rows*cols + j*rows + i].
 *
 * Error convention: every function returns a copra_status_t; COPRA_ERR_DOMAIN corresponds to the reference's
 * std::domain_error (include/debugUtils.h:32-36), COPRA_ERR_RUNTIME to std::runtime_error (:38-42).  No C++
 * exception crosses this boundary; the C++ wrappers (copra_amd/cpp) translate codes back into those exceptions.
 * copra_last_error() returns a thread-local message.
 *
 * The library REQUIRES a HIP device (gfx950).  There is no CPU fallback: without a usable device every compute
 * entry point returns COPRA_ERR_HIP.
 */
#ifndef COPRA_HIP_H
#define COPRA_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    COPRA_OK = 0,
    COPRA_ERR_DOMAIN = 1, /* std::domain_error: dimension mismatch (debugUtils.h:32-36) */
    COPRA_ERR_RUNTIME = 2, /* std::runtime_error (debugUtils.h:38-42) */
    COPRA_ERR_HIP = 3, /* HIP runtime failure / no device */
    COPRA_ERR_UNSUPPORTED = 4, /* valid copra problem that this build of the engine does not cover */
    COPRA_ERR_ARG = 5 /* NULL / negative argument */
} copra_status_t;

/* Cost kinds: include/costFunctions.h:103 (TrajectoryCost), :134 (TargetCost), :165 (ControlCost), :196 (MixedCost).
 * COPRA_COST_DENSE: a user-defined CostFunction subclass (plug-in point 2, include/costFunctions.h:22-97) whose update()
 * ran on the host: the base-class members Q_ (fullUDim x fullUDim), c_ (fullUDim), E_ (xDim x fullUDim), f_ (fullUDim)
 * as LMPC::makeQPForm (src/LMPC.cpp:252-255: Q += Q_, c += c_) and InitialStateLMPC::makeQPForm
 * (src/InitialStateLMPC.cpp:80-84: Q += Q_, E += E_, tail of c += f_) consume them. */
typedef enum {
    COPRA_COST_TRAJECTORY = 0,
    COPRA_COST_TARGET = 1,
    COPRA_COST_CONTROL = 2,
    COPRA_COST_MIXED = 3,
    COPRA_COST_DENSE = 4
} copra_cost_kind_t;

/* Constraint kinds: include/constraints.h:114, :153, :193, :234, :284 */
typedef enum {
    COPRA_CSTR_TRAJECTORY = 0,
    COPRA_CSTR_CONTROL = 1,
    COPRA_CSTR_MIXED = 2,
    COPRA_CSTR_TRAJECTORY_BOUND = 3,
    COPRA_CSTR_CONTROL_BOUND = 4,
    /* a user-defined EqIneqConstraint subclass (plug-in point 2, include/constraints.h:42-107) whose update() ran on the
     * host: A_ (rows x fullUDim), b_ (rows), Y_ (rows x xDim), z_ (rows) with b = z - Y x0.  LMPC stacks [A | b]
     * (src/LMPC.cpp:257-271), InitialStateLMPC stacks [Y, A | z] (src/InitialStateLMPC.cpp:88-102). */
    COPRA_CSTR_DENSE = 5
} copra_cstr_kind_t;

/* Per-instance solver status == SolverInterface::SI_fail() of QuadProgDenseSolver (include/QuadProgSolver.h:21-27),
 * plus one engine-specific code. */
typedef enum {
    COPRA_QP_OK = 0, /* "No problems" */
    COPRA_QP_INFEASIBLE = 1, /* "The minimization problem has no solution" */
    COPRA_QP_NOT_PD = 2, /* "Problems with the decomposition of Q" */
    COPRA_QP_ITER_LIMIT = 3 /* engine safety cap on active-set iterations (never hit by a well-posed QP) */
} copra_qp_status_t;

/* A cost function exactly as handed to LMPC::addCost (src/LMPC.cpp:118-122), i.e. the constructor arguments of
 * include/costFunctions.h:112-118 / :142-148 / :171-177 / :203-210 plus CostFunction::weights (:54-67).
 * Host pointers; copied at copra_batch_create.  The parameters are shared by every instance of the batch. */
typedef struct {
    int kind; /* copra_cost_kind_t */
    int rows; /* rows of M / N / p / weights */
    int m_cols; /* cols of M: xDim (per-step entry) or fullXDim (full-size entry); 0 if unused */
    int n_cols; /* cols of N: uDim or fullUDim; 0 if unused */
    const double* M; /* rows x m_cols */
    const double* N; /* rows x n_cols */
    const double* p; /* rows */
    const double* weights; /* rows */
    /* COPRA_COST_DENSE only (column-major; rows / M / N / p / weights unused): LMPC reads Q and c, InitialStateLMPC reads
     * Q, E and f; a pointer the variant does not read may be NULL */
    const double* Q; /* fullUDim x fullUDim, symmetric */
    const double* c; /* fullUDim */
    const double* E; /* xDim x fullUDim */
    const double* f; /* fullUDim */
} copra_cost_desc_t;

/* A constraint exactly as handed to LMPC::addConstraint (src/LMPC.cpp:124-128): constructor arguments of
 * include/constraints.h:125-132 / :165-172 / :209-217 / :242-255 / :296-303. */
typedef struct {
    int kind; /* copra_cstr_kind_t */
    int rows; /* rows of E / G / f, or length of lower / upper */
    int e_cols; /* cols of E (xDim | fullXDim), 0 if unused */
    int g_cols; /* cols of G (uDim | fullUDim), 0 if unused */
    int is_inequality; /* isInequalityConstraint (constraints.h:126); ignored by the bound kinds */
    const double* E;
    const double* G;
    const double* f;
    const double* lower;
    const double* upper;
    /* COPRA_CSTR_DENSE only (column-major, `rows` rows, is_inequality as above): LMPC reads A and b, InitialStateLMPC
     * reads Y, A and z */
    const double* A; /* rows x fullUDim */
    const double* b; /* rows */
    const double* Y; /* rows x xDim */
    const double* z; /* rows */
} copra_cstr_desc_t;

/* Dimensions of the preview systems of one batch: PreviewSystem::system(state, control, bias, xInit, numberOfSteps)
 * (src/PreviewSystem.cpp:16-55). */
typedef struct {
    int nx; /* xDim */
    int nu; /* uDim */
    int N; /* nrUStep */
    int batch; /* number of independent preview systems (instances) */
} copra_dims_t;

/* InitialStateLMPC::resetInitialStateCost(R, r) (src/InitialStateLMPC.cpp:35-40): R [nx x nx] positive definite, r [nx];
 * shared by the batch. */
typedef struct {
    const double* R;
    const double* r;
} copra_initial_state_desc_t;

typedef struct copra_batch copra_batch_t; /* opaque handle == one batched LMPC controller */

/* ---- engine options: every switch the engine consults, fixed when the controller is created (no environment
 *      variable is read by the library on any path; earlier rounds steered it through ~50 COPRA_* variables).  There is no reference
 *      counterpart: copra's LMPC has no tuning surface beyond selectQPSolver.  copra_options_init fills in the process-wide defaults
 *      (all zeros / "the engine decides" unless copra_set_default_options changed them); a caller sets `struct_size =
 *      sizeof(copra_options_t)` (done by copra_options_init) so that a library built against a longer struct keeps defaults for the
 *      fields the caller does not know.  Integer switches: 0 = off / engine's choice.  Results are the same under every setting --
 *      the options choose WHICH kernels run (tests pin each tier by switching the others off), never what they compute. ---- */
typedef struct {
    int struct_size;
    /* plan builder: classification of full-size entries as per-step entries */
    int no_stage_refs; /* block-diagonal full-size costs with repeating blocks stay dense Psi'WPsi contractions */
    int no_step_rows; /* full-size constraint rows inside one step keep the full-row machinery */
    int no_selection_rows; /* one-hot TrajectoryConstraint rows are not treated like TrajectoryBound rows */
    /* tier selection of the one-wave kernels */
    int no_ric; /* never the Riccati-factor tier (lmpc_fused_ric.hpp) */
    int no_tri; /* never a factor-only tier */
    int ric_general; /* Riccati-factor tier: the general variant even where the compact one applies */
    int no_dense_layout; /* run-time shapes: the safe compact layout instead of the densest one */
    int no_q1regs; /* factor-only tier with Q1 in LDS */
    int no_ladder; /* the layout ladder behaves as exhausted */
    int no_packed; /* small problems: one wavefront per instance instead of 16 / 32 lanes */
    int ric_k; /* Riccati-factor tier: start on the LDS-Q1 layout for this many instances per CU (0: Q1 in registers) */
    /* the one-instance-per-lane pass in front of the first tier (lmpc_lane.hpp) */
    int no_lane_pass;
    int no_lane_handover; /* the pass only filters; the tier sweeps itself */
    int no_lane_spec; /* the pass does not take the first step of the active-set iteration itself (a bound on u_0 as the first pick) */
    int no_lane_axes; /* the pass treats every system as dense: decoupled axes (FusedPlan::lane_axes -- state i on axis i % nu, control c on axis c, nothing
                         between them in A, B and the costs: the CoM model) are not looked for, no product is left out */
    int lane_min_batch; /* smallest batch that runs the pass (0: 20480 in front of the Riccati-factor tier, 4096 elsewhere; -1: any) */
    /* shared-model mode */
    int no_ric_shared; /* lmpc_shared.hpp instead of the Riccati-factor tier's shared-model mode */
    /* long horizons */
    int no_riccati; /* COPRA_SOLVER_DEFAULT keeps Goldfarb-Idnani where the Riccati interior-point kernel would be picked */
    int no_ric_fast; /* the streaming interior-point kernel instead of the LDS-resident one */
    double ric_step_tol, ric_mu_tol; /* interior-point tolerances (0: defaults of stage_plan.hpp) */
    /* misc */
    int debug; /* print launch geometry and adaptation decisions to stderr */
    /* (fields are only ever APPENDED from here on: a caller built against a shorter struct keeps its meaning, struct_size says how much it knows) */
    int no_axis_solver; /* never the one-(instance, axis)-per-lane solver (lmpc_axis.hpp; round 6): controllers whose axes are decoupled keep the
                           one-instance-per-lane pass + first tier */
} copra_options_t;
void copra_options_init(copra_options_t* opts);
/* the process-wide defaults copra_options_init hands out and the entry points without an options argument use (copra_batch_create,
 * copra_batch_create_initial_state, copra_plan_check, copra_qp_solve_dense_batch); NULL restores the built-in ones */
copra_status_t copra_set_default_options(const copra_options_t* opts);

/* ---- controller life cycle (replaces LMPC::LMPC / initializeController / addCost / addConstraint,
 *      src/LMPC.cpp:56-77, 118-128; dimension checks of costFunctions.cpp:44-61,88-98,122-137,173-193 and
 *      constraints.cpp:45-64,106-135,171-195,263-282,333-357 -> COPRA_ERR_DOMAIN) ---- */
copra_status_t copra_batch_create(copra_batch_t** out, const copra_dims_t* dims, int n_costs,
    const copra_cost_desc_t* costs, int n_cstrs, const copra_cstr_desc_t* cstrs);
/* the same with explicit engine options (`is` NULL: LMPC, else InitialStateLMPC as copra_batch_create_initial_state; `opts` NULL:
 * the process-wide defaults) */
copra_status_t copra_batch_create_with_options(copra_batch_t** out, const copra_dims_t* dims, int n_costs,
    const copra_cost_desc_t* costs, int n_cstrs, const copra_cstr_desc_t* cstrs, const copra_initial_state_desc_t* is,
    const copra_options_t* opts);
void copra_batch_destroy(copra_batch_t* h);

/* ---- how the controller is mapped onto the device: lanes that work on ONE instance -- 16 or 32 (several small
 *      problems share a 64-lane wavefront), 64 (one wavefront per instance) or the workgroup size of the
 *      workgroup-per-instance kernel (64 < decision variables <= 512). ---- */
int copra_batch_lanes_per_instance(const copra_batch_t* h);

/* ---- run-time specialisation: compile the kernels for THIS controller's (xDim, uDim, nrStep, cost rows) with
 *      `hipcc --genco` from the headers next to libcopra_hip.so (about 20-40 s, once per shape: the code object is kept in
 *      cache_dir, or $COPRA_JIT_CACHE, or ~/.cache/copra_amd when NULL; the compiler is $HIPCC or /opt/rocm/bin/hipcc -- the
 *      only environment the library reads, and only here) and use them from the next solve on.  The library
 *      ships such instantiations for the BASELINE shapes only; every other shape otherwise runs on the run-time-shape
 *      kernel, which is ~2.5x slower on the same problem.  A no-op (COPRA_OK) for shapes that already have dedicated
 *      kernels (BASELINE shapes, packed small problems, InitialStateLMPC, more than 64 variables).  Results are the
 *      same either way; COPRA_ERR_RUNTIME if hipcc is not available. ---- */
copra_status_t copra_batch_specialise(copra_batch_t* h, const char* cache_dir);
/* the same with a gate between compiler and loader: `check(path, user)` is called with the code object's path (cached or freshly
 * compiled) BEFORE hipModuleLoad; a non-zero answer deletes the object, leaves the handle exactly as it was (the library's kernels)
 * and returns COPRA_ERR_RUNTIME.  copra_amd/batch.py passes the matrix-instruction hazard lint (copra_amd/hazard_lint.py) here. */
typedef int (*copra_code_object_check_t)(const char* code_object_path, void* user);
copra_status_t copra_batch_specialise_checked(copra_batch_t* h, const char* cache_dir, copra_code_object_check_t check, void* user);

/* ---- the checks of copra_batch_create / copra_batch_create_initial_state WITHOUT touching the device: what
 *      LMPC::addCost / addConstraint do when they call initializeCost / initializeConstraint (src/LMPC.cpp:118-128).
 *      `is` may be NULL.  Returns COPRA_OK, COPRA_ERR_DOMAIN, COPRA_ERR_RUNTIME or COPRA_ERR_UNSUPPORTED. ---- */
copra_status_t copra_plan_check(const copra_dims_t* dims, int n_costs, const copra_cost_desc_t* costs, int n_cstrs,
    const copra_cstr_desc_t* cstrs, const copra_initial_state_desc_t* is);

/* ---- InitialStateLMPC variant (include/InitialStateLMPC.h:18-42, src/InitialStateLMPC.cpp): the decision vector is
 *      [x0; U]; costs contribute E and f, the Hessian is [[R + E Q^-1 E', E], [E', Q]] (InitialStateLMPC.cpp:77-122).
 *      Covered on the device for xDim + fullUDim <= 512 (xDim <= 16), per-step and full-size entries; other shapes
 *      COPRA_ERR_UNSUPPORTED.
 *      Initial-state bounds (resetInitialStateBounds, :42-46) are per instance, [batch][nx]; when they are never set
 *      both default to the x0 handed to copra_batch_set_system (InitialStateLMPC.cpp:20-28).
 *      copra_batch_get_initial_state == InitialStateLMPC::initialState() (:30-33), [batch][nx]. ---- */
copra_status_t copra_batch_create_initial_state(copra_batch_t** out, const copra_dims_t* dims, int n_costs,
    const copra_cost_desc_t* costs, int n_cstrs, const copra_cstr_desc_t* cstrs, const copra_initial_state_desc_t* is);
copra_status_t copra_batch_set_initial_state_bounds(copra_batch_t* h, const double* x0lb, const double* x0ub,
    int on_device);
copra_status_t copra_batch_get_initial_state(copra_batch_t* h, double* x0_opt);

/* ---- PreviewSystem::system / xInit for every instance (src/PreviewSystem.cpp:16-55, include/PreviewSystem.h:52).
 *      A [batch][nx x nx], B [batch][nx x nu], d [batch][nx], x0 [batch][nx].
 *      on_device != 0: the pointers are device pointers and are USED IN PLACE (they must stay valid until the next
 *      call); on_device == 0: host pointers, copied H2D into engine-owned HBM buffers. ---- */
copra_status_t copra_batch_set_system(copra_batch_t* h, const double* A, const double* B, const double* d,
    const double* x0, int on_device);
copra_status_t copra_batch_set_x0(copra_batch_t* h, const double* x0, int on_device);
/* ---- the same from ROW-major per-instance matrices that are already on the device (what C arrays and numpy hold; Eigen --
 *      src/PreviewSystem.cpp:16-55 takes Eigen matrices -- and every other entry point here are column-major): the library
 *      transposes A and B into buffers of its own with a kernel on `hip_stream` (no host-side layout conversion, nothing
 *      blocks); d and x0 are used in place.  For host pipelines that stage pinned buffers asynchronously. ---- */
copra_status_t copra_batch_set_system_rowmajor_async(copra_batch_t* h, const double* A, const double* B, const double* d,
    const double* x0, void* hip_stream);

/* ---- per-instance references: p of cost `cost_index` (the order of the `costs` array given at creation) for EVERY
 *      instance, [batch][rows] with the rows of that cost as created (per-step entry: r, full-size entry: r (N+1) or
 *      r N) -- each instance of the batch tracks its own goal / reference trajectory; in the reference this is one
 *      TrajectoryCost(M, p_b) / TargetCost / ControlCost / MixedCost object per LMPC (include/costFunctions.h:103-219).
 *      p == NULL restores the controller-wide p.  on_device != 0: used in place.  Works on the shared-model fast path
 *      too (the gradient is affine in p: c = c0 + C1 x0 + C2 p, probed once -- one column of C2 per entry of p, reference
 *      trajectories included; from 10 240 instances on, on the shapes of the Riccati-factor tier, the batch-wide stage records stay
 *      and every instance adds the delta of its feed-forward terms: 193 M solves/s with per-instance goals, 327 M with per-instance
 *      reference trajectories at the headline shape, profiles/r04/shared_goals.txt, shared_tracking.txt). ---- */
copra_status_t copra_batch_set_cost_reference(copra_batch_t* h, int cost_index, const double* p, int on_device);
/* ... and ONE new reference for every instance: p[rows] (rows as above -- a reference trajectory: r (N+1) or r N), host or device.
 *      The reference's API has no setter for p (include/costFunctions.h:103-219: a constructor argument): a tracking controller
 *      replaces the cost object and the next solve evaluates the new one (src/LMPC.cpp:233-247).  Here the new reference is written
 *      once per instance into the library's buffer (a broadcast on the device) and the per-instance path above is taken: no new
 *      plan, no new handle. */
copra_status_t copra_batch_set_cost_reference_all(copra_batch_t* h, int cost_index, const double* p, int on_device);
/*      Cost of that convenience: the controller is in per-instance-reference mode afterwards (until copra_batch_set_cost_reference(h,
 *      k, NULL, 0), which restores the reference given at CREATION).  In that mode a long-horizon controller runs the streaming
 *      interior-point kernel instead of the LDS-resident one (config 5: about 0.6 x the rate), a shared-model controller leaves the
 *      Riccati-factor tier's shared mode, and every call broadcasts batch x rows doubles.  Where that matters -- InitialStateLMPC /
 *      long horizons, shared-model ticks -- create a new controller with the new reference instead; the C++ and Python mirrors do so
 *      for controllers with more than 64 decision variables. */

/* ---- per-instance constraint data.  copra_batch_set_constraint_rhs: f of the Trajectory / Control / Mixed constraint
 *      `cstr_index` (position in the `cstrs` array given at creation) for every instance, [batch][rows] with the rows
 *      of that constraint as created (per-step entry: r values used at every step; full-size entry: all of them) -- in
 *      the reference one TrajectoryConstraint(E, f_b) ... object per LMPC (include/constraints.h:114-226).
 *      copra_batch_set_control_bounds: lower / upper of the ControlBoundConstraint for every instance, [batch][fullUDim]
 *      (include/constraints.h:284-308).  TrajectoryBoundConstraint stays controller-wide (its infinite components
 *      are dropped at creation, src/constraints.cpp:263-282).  Both copy their input. ---- */
copra_status_t copra_batch_set_constraint_rhs(copra_batch_t* h, int cstr_index, const double* f, int on_device);
copra_status_t copra_batch_set_control_bounds(copra_batch_t* h, const double* lower, const double* upper, int on_device);

/* ---- shared-model receding-horizon fast path: ONE preview system (A [nx x nx], B [nx x nu], d [nx], column-major) for
 *      the whole batch; only x0 differs per instance (copra_batch_set_x0, [batch][nx]).  This is the reference's own
 *      receding-horizon use: PreviewSystem::xInit between solves with isUpdated left true (include/PreviewSystem.h:
 *      52-54, src/LMPC.cpp:233), where c = E'x0 + f and b = z - Y x0 carry the whole x0-dependence
 *      (src/costFunctions.cpp:80, src/constraints.cpp:81).  The preview matrices, the Hessian, its Cholesky factor and
 *      J = R^-1 are then computed ONCE (at the first copra_batch_solve after this call) and every solve only forms the
 *      free response and the gradient, copies J and runs the active-set loop.  LMPC with at most 64 decision variables;
 *      InitialStateLMPC / larger problems: COPRA_ERR_UNSUPPORTED.  Results are those of copra_batch_set_system with the
 *      same system replicated. ---- */
copra_status_t copra_batch_set_shared_system(copra_batch_t* h, const double* A, const double* B, const double* d,
    int on_device);

/* ---- LMPC::selectQPSolver(SolverFlag) (src/LMPC.cpp:62-65, include/solverUtils.h:34-50) for the batched controller.
 *      COPRA_SOLVER_DEFAULT (SolverFlag::DEFAULT): the engine picks -- the condensed Goldfarb-Idnani kernels up to 64
 *        decision variables, and for longer horizons the stage-wise Riccati interior-point kernel when every cost /
 *        constraint of the controller is stage-wise (per-step entries, block-diagonal full-size entries, at most 32
 *        equality rows), else the condensed workgroup-per-instance Goldfarb-Idnani kernel; instances the interior-point
 *        kernel does not converge on (infeasible ones) are finished by Goldfarb-Idnani, which sets their status.
 *      COPRA_SOLVER_QUADPROG_DENSE (SolverFlag::QuadProgDense): always the Goldfarb-Idnani kernels -- the arithmetic of
 *        the reference's QuadProgDense path, same active-set iteration counts as the CPU path.
 *      COPRA_SOLVER_RICCATI_IPM: force the interior-point kernel; COPRA_ERR_UNSUPPORTED when the controller is not
 *        stage-wise.
 *      Both solvers return the unique optimum of the same strictly convex QP: controls / trajectories agree to solver
 *      accuracy; status codes agree; iter[0] counts active-set iterations or Newton steps respectively.
 *      copra_batch_solver_info: which one the next copra_batch_solve runs (a copra_solver_t). ---- */
typedef enum { COPRA_SOLVER_DEFAULT = 0, COPRA_SOLVER_QUADPROG_DENSE = 1, COPRA_SOLVER_RICCATI_IPM = 2 } copra_solver_t;
copra_status_t copra_batch_select_solver(copra_batch_t* h, int solver);
int copra_batch_solver_info(const copra_batch_t* h);

/* ---- warm start of the active set across receding-horizon ticks on the shared-model path: with
 *      enable != 0 every instance remembers the active set it ended with, moved one step towards the present (the rows of
 *      step 0 leave), and its next solve takes those rows as the first candidates of the dual active-set method: while the
 *      list lasts, the next constraint to activate is the list's next row that is violated at the current iterate (one slack
 *      evaluation instead of a scan over all rows; the method may pick ANY violated constraint, its invariants hold
 *      throughout).  Same optimum and status as a cold start.  enable == 0 forgets the stored sets. ---- */
copra_status_t copra_batch_set_warm_start(copra_batch_t* h, int enable);

/* ---- optional: let the caller own the result buffers (device pointers, e.g. torch tensors that are later handed to
 *      an RCCL gather); must be called before copra_batch_solve and stay valid.  Sizes as in the results block. ---- */
copra_status_t copra_batch_set_outputs(copra_batch_t* h, double* control, double* trajectory, int* status, int* iter);

/* ---- LMPC::solve for every instance (src/LMPC.cpp:79-101): condensed-QP build + Goldfarb-Idnani solve +
 *      updateResults, one launch on `hip_stream` (a hipStream_t, may be NULL = default stream).  Asynchronous. ---- */
copra_status_t copra_batch_solve(copra_batch_t* h, void* hip_stream);
copra_status_t copra_batch_synchronize(copra_batch_t* h);

/* ---- results: LMPC::control() / trajectory() (include/LMPC.h:108-110), SI_fail / SI_iter
 *      (src/QuadProgSolver.cpp:14-22).  control [batch][nu*N], trajectory [batch][nx*(N+1)], status [batch],
 *      iter [batch][2] (main iterations, constraint drops).  Device-resident accessors return engine-owned HBM
 *      pointers valid for the lifetime of the handle.  A failed instance keeps NaN in control/trajectory. ---- */
const double* copra_batch_control_device(const copra_batch_t* h);
const double* copra_batch_trajectory_device(const copra_batch_t* h);
const int* copra_batch_status_device(const copra_batch_t* h);
const int* copra_batch_iter_device(const copra_batch_t* h);
copra_status_t copra_batch_get_results(copra_batch_t* h, double* control, double* trajectory, int* status, int* iter);

/* ---- parity hooks: the dense QP of one instance as LMPC exposes it (include/LMPC.h:112-127: Q c Aineq bineq Aeq
 *      beq lb ub), rebuilt ON THE DEVICE by the same condense code the solver runs.  Any pointer may be NULL.
 *      Q [n x n], c [n], Aeq [neq x n], beq [neq], Aineq [nineq x n], bineq [nineq], lb [n], ub [n]. ---- */
copra_status_t copra_batch_qp_sizes(const copra_batch_t* h, int* nvar, int* neq, int* nineq);
copra_status_t copra_batch_dump_qp(copra_batch_t* h, int instance, double* Q, double* c, double* Aeq, double* beq,
    double* Aineq, double* bineq, double* lb, double* ub);

/* ---- how the next copra_batch_solve maps instances to the GPU (no reference counterpart; for tests and tuning):
 *      LDS bytes per instance of the first launch, number of active constraints that launch has room for (instances
 *      that need more finish in a second launch), 1 if it keeps only the Cholesky factor (no inverse factor),
 *      1 if a second launch exists.  The first solves of a controller may step to a roomier layout. ---- */
copra_status_t copra_batch_layout_info(const copra_batch_t* h, int* lds_bytes, int* active_capacity, int* factor_only,
    int* two_tier);

/* ---- timing contract of LMPC::solveTime()/solveAndBuildTime() (src/LMPC.cpp:82-99, 108-116): device time of the
 *      last copra_batch_solve in seconds (hipEvent pair on the launch stream); whole batch, EVERY launch of the solve (the
 *      second launch, which only sees the instances whose active set outgrew the first one's layout, included).  For the
 *      one-wave kernels the events ride in the dispatch packets of the launches themselves; copra_options_t::recorded_events
 *      brackets the solve with recorded events instead. ---- */
copra_status_t copra_batch_last_solve_seconds(copra_batch_t* h, double* seconds);
/* ---- the first launch of that solve alone (no reference counterpart: the figure a kernel profile -- rocprofv3 -- of the
 *      dominant kernel is compared with); equals copra_batch_last_solve_seconds where the solve is not timed per launch. ---- */
copra_status_t copra_batch_last_first_tier_seconds(copra_batch_t* h, double* seconds);
/* ---- the one-instance-per-lane pass of the last solve (no reference counterpart; for tests, tuning and the bench line).  For a
 *      controller whose costs are all per-step entries (src/costFunctions.cpp:63-215) the unconstrained minimiser qpgen2 starts from
 *      is the LQ roll-out of one Riccati sweep; a pass in front of the first tier does that sweep and roll-out for the whole batch
 *      with one instance per LANE, finishes every instance whose minimiser violates no constraint (what QuadProgDense::solve returns
 *      after its first scan, src/QuadProgSolver.cpp:45-72) and hands the factor of the others to the first tier.
 *      ran: 1 if the last solve ran it, 2 if it ran the one-(instance, axis)-per-lane solver instead (lmpc_axis.hpp, round 6: controllers whose
 *      axes are decoupled -- the WHOLE solve per lane; what it cannot finish goes to the first tier the same way); finished: the instances that
 *      ended in it (waits for the solve; NULL: not asked for). ---- */
copra_status_t copra_batch_lane_pass_info(copra_batch_t* h, int* ran, int* finished);
/*      Whether the pass runs is decided per controller from the batch size (from 20480 instances on in front of the Riccati-factor
 *      tier, 4096 elsewhere) and, where it only filters, from the share of instances that ended in it in the first two solves --
 *      sampled again every 256 solves.  Which instances the pass or the tier finish does not change statuses or iteration counts,
 *      but the two evaluate the unconstrained minimiser in different orders of summation: results are reproducible to rounding
 *      (1e-11 relative), not bit for bit, across batch sizes and across the solves around such a decision. */

/* ---- device-side split of that time: per-instance shader-clock cycles of the 7 phases of the fused kernel
 *      (preview, costs, norms, cholesky, inverse+x0, active set, result stores) + total; 8 values per instance.
 *      enable != 0 turns the stamps on for subsequent solves; cycles_out (host, [batch][8]) may be NULL. ---- */
copra_status_t copra_batch_phase_profile(copra_batch_t* h, int enable, long long* cycles_out);

/* ---- plug-in point 1, batched: QuadProgDenseSolver::SI_problem + SI_solve (src/QuadProgSolver.cpp:45-72) for
 *      `batch` independent dense QPs  min 1/2 x'Qx + c'x  s.t. Aeq x = beq, Aineq x <= bineq, XL <= x <= XU.
 *      Q [batch][n x n] (upper triangle read), c [batch][n], Aeq [batch][neq x n], beq [batch][neq],
 *      Aineq [batch][nineq x n], bineq [batch][nineq], XL/XU [batch][n]; outputs x [batch][n], fail [batch]
 *      (SI_fail), iter [batch][2].  on_device selects host or device pointers for ALL arrays.
 *      n <= 64: one QP per wavefront, Q/J/R in LDS; 64 < n <= 512: one QP per workgroup, J/R in an HBM workspace
 *      (the call then allocates that workspace and synchronises the stream); n > 512: COPRA_ERR_UNSUPPORTED. ---- */
copra_status_t copra_qp_solve_dense_batch(int batch, int n, int neq, int nineq, const double* Q, const double* c,
    const double* Aeq, const double* beq, const double* Aineq, const double* bineq, const double* XL,
    const double* XU, double* x, int* fail, int* iter, int on_device, void* hip_stream);

/* ---- PreviewSystem::updateSystem (src/PreviewSystem.cpp:57-74) for ONE system, on the device: Phi [fullXDim x xDim],
 *      Psi [fullXDim x fullUDim], xi [fullXDim], column-major host buffers.  The solve path never materialises them (they
 *      only exist block-wise in LDS); this entry point serves host-evaluated user subclasses of Constraint / CostFunction,
 *      whose update(const PreviewSystem&) reads ps.Phi / ps.Psi / ps.xi. ---- */
copra_status_t copra_preview_update(int nx, int nu, int N, const double* A, const double* B, const double* d, double* Phi,
    double* Psi, double* xi);

/* ---- misc ---- */
/* ---- run-time specialisation of the dense-QP kernel for problems with `n` variables (n <= 64): as
 *      copra_batch_specialise; later copra_qp_solve_dense_batch calls with that n use the compiled kernels. ---- */
copra_status_t copra_qp_dense_specialise(int n, const char* cache_dir);

const char* copra_last_error(void);
copra_status_t copra_device_info(int* n_devices, int* cu_count, char* arch_name, int arch_name_len);
int copra_abi_version(void);
/* sha1 over the sources this library was built from (copra_amd/csrc/Makefile: COPRA_SRC_HASH; the same value keys the cache of run-time-
 * compiled kernels): what a committed profile records so that a benchmark can tell whether the counters it quotes were taken on THIS build */
const char* copra_source_hash(void);

#ifdef __cplusplus
}
#endif
#endif /* COPRA_HIP_H */
