/*
 * copra_oracle.h -- CPU ORACLE (test infrastructure only, never shipped, never on the product path).
 *
 * A plain-C restatement of the reference's (jrl-umi3218/copra v1.3.3) condensed linear-MPC hot path:
 *   PreviewSystem -> cost functions -> constraints -> LMPC::makeQPForm -> QuadProgDenseSolver
 *   -> Goldfarb-Idnani dual active-set QP (eigen-quadprog / qpgen2 semantics) -> LMPC::updateResults,
 * plus the InitialStateLMPC variant.  Every function cites the reference file:line it follows.
 *
 * PARITY STATUS: the reference cannot be compiled or imported in the build container (no Eigen3, no
 * eigen-quadprog, no gfortran), and the QP arithmetic lives in the un-vendored, un-pinned third-party
 * dependency eigen-quadprog (wrapping Turlach/Weingessel's qpgen2).  The reference's own tests hold no
 * numeric golden vectors for this path.  The oracle is therefore pinned ONLY against
 *   (1) the Scilab-qld known-answer QP of tests/systems.h:11-38,
 *   (2) the analytic answer of tests/systems.h:187-229 (EqSystem, u_k = m*g),
 *   (3) every property check of tests/TestLMPC.cpp / TestLMPC_InitialState.cpp replayed at N=300,
 *   (4) an independent numpy/scipy KKT cross-check (tests/golden/, self-generated).
 * Bit-level pivot-order parity with eigen-quadprog is UNPINNED ("parity unpinned" at the QP boundary);
 * parity rests on the QP being strictly convex (unique optimum).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef COPRA_ORACLE_H
#define COPRA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* cost kinds -- include/costFunctions.h:103,134,165,196 */
enum { OR_COST_TRAJECTORY = 0, OR_COST_TARGET = 1, OR_COST_CONTROL = 2, OR_COST_MIXED = 3 };
/* constraint kinds -- include/constraints.h:114,153,193,234,284 */
enum {
    OR_CSTR_TRAJECTORY = 0,
    OR_CSTR_CONTROL = 1,
    OR_CSTR_MIXED = 2,
    OR_CSTR_TRAJECTORY_BOUND = 3,
    OR_CSTR_CONTROL_BOUND = 4
};

/* return codes */
enum {
    OR_OK = 0,
    OR_FAIL_NO_SOLUTION = 1, /* QuadProgSolver.h:24 */
    OR_FAIL_DECOMPOSITION = 2, /* QuadProgSolver.h:25 */
    OR_ERR_DOMAIN = -1, /* std::domain_error of debugUtils.h:32-36 */
    OR_ERR_RUNTIME = -2 /* std::runtime_error of debugUtils.h:38-42 */
};

/* One cost function as the user hands it to LMPC::addCost (all matrices column-major like Eigen). */
typedef struct {
    int kind;
    int rows; /* rows of M / N / p / weights */
    int m_cols; /* cols of M: xDim (per-step entry) or fullXDim (full-size entry); 0 when unused */
    int n_cols; /* cols of N: uDim or fullUDim; 0 when unused */
    const double* M;
    const double* N;
    const double* p;
    const double* w; /* weights_, length rows (costFunctions.h:117: default ones) */
} or_cost_t;

/* One constraint as the user hands it to LMPC::addConstraint. */
typedef struct {
    int kind;
    int rows; /* rows of E / G / f, or length of lower / upper for the bound kinds */
    int e_cols; /* cols of E (xDim | fullXDim), 0 when unused */
    int g_cols; /* cols of G (uDim | fullUDim), 0 when unused */
    int is_ineq; /* constraints.h:126: isInequalityConstraint */
    const double* E;
    const double* G;
    const double* f;
    const double* lower;
    const double* upper;
} or_cstr_t;

/* The dense QP LMPC hands to SolverInterface::SI_solve (LMPC.h:113-127 getters). */
typedef struct {
    int nvar, neq, nineq;
    int nx, nu, N, fullX, fullU;
    double *Q, *c, *Aeq, *beq, *Aineq, *bineq, *lb, *ub; /* column-major, owned */
    double *Phi, *Psi, *xi; /* preview matrices, owned */
} or_qp_t;

void or_qp_free(or_qp_t* qp);

/* PreviewSystem::updateSystem -- src/PreviewSystem.cpp:57-74 */
void or_preview_update(int nx, int nu, int N, const double* A, const double* B, const double* d,
    double* Phi, double* Psi, double* xi);

/* LMPC::updateSystem + makeQPForm -- src/LMPC.cpp:225-280 (fresh controller, first solve()) */
int or_lmpc_build(int nx, int nu, int N, const double* A, const double* B, const double* d, const double* x0,
    int ncost, const or_cost_t* costs, int ncstr, const or_cstr_t* cstrs, or_qp_t* out);

/* InitialStateLMPC::makeQPForm -- src/InitialStateLMPC.cpp:77-122 */
int or_islmpc_build(int nx, int nu, int N, const double* A, const double* B, const double* d, const double* x0,
    int ncost, const or_cost_t* costs, int ncstr, const or_cstr_t* cstrs,
    const double* R, const double* r, const double* x0lb, const double* x0ub, or_qp_t* out);

/* QuadProgDenseSolver::SI_problem + SI_solve -- src/QuadProgSolver.cpp:45-72, then eigen-quadprog's
 * qpgen2 (Goldfarb-Idnani).  Returns SI_fail() (0, 1, 2).  iter[0] = SI_iter(), iter[1] = #drops. */
int or_quadprog_dense(int n, int neq, int nineq, const double* Q, const double* c, const double* Aeq,
    const double* beq, const double* Aineq, const double* bineq, const double* XL, const double* XU,
    double* x, int* iter);

/* LMPC::solve -- src/LMPC.cpp:79-101: build, solve, updateResults (control = U, trajectory = Phi x0 + Psi U + xi).
 * Returns SI_fail() or a negative OR_ERR_*. On failure control/trajectory are left untouched (LMPC.cpp:95-97). */
int or_lmpc_solve(int nx, int nu, int N, const double* A, const double* B, const double* d, const double* x0,
    int ncost, const or_cost_t* costs, int ncstr, const or_cstr_t* cstrs,
    double* control, double* trajectory, int* iter);

/* InitialStateLMPC::solve: control = tail(fullU), trajectory = Phi x0* + Psi U + xi, x0_opt = head(nx). */
int or_islmpc_solve(int nx, int nu, int N, const double* A, const double* B, const double* d, const double* x0,
    int ncost, const or_cost_t* costs, int ncstr, const or_cstr_t* cstrs,
    const double* R, const double* r, const double* x0lb, const double* x0ub,
    double* control, double* trajectory, double* x0_opt, int* iter);

/* Batched driver (one independent controller per instance, static partition over nthreads pthreads): the
 * CPU baseline of bench.py.  A/B/d/x0 are batch-major ([b][...], column-major inside an instance).
 * status[b] = SI_fail(); iter[2*b..] .  Returns 0 or the first negative error. */
int or_lmpc_solve_batch(int batch, int nthreads, int nx, int nu, int N, const double* A, const double* B,
    const double* d, const double* x0, int ncost, const or_cost_t* costs, int ncstr, const or_cstr_t* cstrs,
    double* control, double* trajectory, int* status, int* iter);

#ifdef __cplusplus
}
#endif
#endif
