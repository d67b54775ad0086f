/*
 * copra_oracle.c -- CPU ORACLE, test infrastructure only (see copra_oracle.h for the parity status:
 * "parity unpinned" at the eigen-quadprog boundary; pinned by known answers + property checks).
 *
 * Dense, column-major, FP64, same loop structure as the reference.  Compile WITHOUT fast-math and with
 * -ffp-contract=off so that the arithmetic is plain IEEE (oracle/Makefile).
 */
#include "copra_oracle.h"

#include <float.h>
#include <malloc.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define AT(M, ld, i, j) ((M)[(size_t)(j) * (size_t)(ld) + (size_t)(i)])

static double* dzeros(size_t n)
{
    double* p = (double*)calloc(n ? n : 1, sizeof(double));
    return p;
}

/* ------------------------------------------------------------------------------------------------
 * PreviewSystem::updateSystem -- src/PreviewSystem.cpp:57-74 (Phi_0 = I, rest zero: :47-54)
 * Phi is fullX x nx, Psi is fullX x fullU, xi is fullX; all column-major.
 * ---------------------------------------------------------------------------------------------- */
void or_preview_update(int nx, int nu, int N, const double* A, const double* B, const double* d,
    double* Phi, double* Psi, double* xi)
{
    const int X = nx * (N + 1), U = nu * N;
    memset(Phi, 0, sizeof(double) * (size_t)X * nx);
    memset(Psi, 0, sizeof(double) * (size_t)X * U);
    memset(xi, 0, sizeof(double) * (size_t)X);
    for (int i = 0; i < nx; ++i) AT(Phi, X, i, i) = 1.0; /* PreviewSystem.cpp:51 */

    /* :59-61 */
    for (int j = 0; j < nx; ++j)
        for (int i = 0; i < nx; ++i) AT(Phi, X, nx + i, j) = AT(A, nx, i, j);
    for (int j = 0; j < nu; ++j)
        for (int i = 0; i < nx; ++i) AT(Psi, X, nx + i, j) = AT(B, nx, i, j);
    for (int i = 0; i < nx; ++i) xi[nx + i] = d[i];

    for (int s = 2; s <= N; ++s) { /* :63 (i < nrXStep) */
        /* Phi_s = A * Phi_{s-1}  (:64) */
        for (int j = 0; j < nx; ++j)
            for (int i = 0; i < nx; ++i) {
                double acc = 0.0;
                for (int k = 0; k < nx; ++k) acc += AT(A, nx, i, k) * AT(Phi, X, (s - 1) * nx + k, j);
                AT(Phi, X, s * nx + i, j) = acc;
            }
        /* Psi_{s,0} = A * Psi_{s-1,0}  (:65) */
        for (int j = 0; j < nu; ++j)
            for (int i = 0; i < nx; ++i) {
                double acc = 0.0;
                for (int k = 0; k < nx; ++k) acc += AT(A, nx, i, k) * AT(Psi, X, (s - 1) * nx + k, j);
                AT(Psi, X, s * nx + i, j) = acc;
            }
        /* Psi_{s,j} = Psi_{s-1,j-1}  (:66-68) */
        for (int jb = 1; jb < s; ++jb)
            for (int j = 0; j < nu; ++j)
                for (int i = 0; i < nx; ++i)
                    AT(Psi, X, s * nx + i, jb * nu + j) = AT(Psi, X, (s - 1) * nx + i, (jb - 1) * nu + j);
        /* xi_s = A xi_{s-1} + d  (:70) */
        for (int i = 0; i < nx; ++i) {
            double acc = 0.0;
            for (int k = 0; k < nx; ++k) acc += AT(A, nx, i, k) * xi[(s - 1) * nx + k];
            xi[s * nx + i] = acc + d[i];
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Cost functions -- src/costFunctions.cpp.  Each produces Q (U x U), c (U), E (nx x U), f (U).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int nx, nu, N, X, U;
    const double *Phi, *Psi, *xi, *x0;
} psys_t;

/* acc helpers: given tmp (r x U, ld r), weights w (r), Mphi (r x nx), resid (r):
 *   Q += tmp^T W tmp ; E += Mphi^T W tmp ; f += resid^T W tmp        (costFunctions.cpp:75-78)       */
static void cost_accumulate(int r, int U, int nx, const double* tmp, const double* w, const double* Mphi,
    const double* resid, double* Q, double* E, double* f, int assign)
{
    for (int b = 0; b < U; ++b) {
        for (int a = 0; a < U; ++a) {
            double acc = 0.0;
            for (int k = 0; k < r; ++k) acc += (AT(tmp, r, k, a) * w[k]) * AT(tmp, r, k, b);
            if (assign)
                AT(Q, U, a, b) = acc;
            else
                AT(Q, U, a, b) += acc;
        }
        for (int a = 0; a < nx; ++a) {
            double acc = 0.0;
            if (Mphi)
                for (int k = 0; k < r; ++k) acc += (AT(Mphi, r, k, a) * w[k]) * AT(tmp, r, k, b);
            if (assign)
                AT(E, nx, a, b) = acc;
            else
                AT(E, nx, a, b) += acc;
        }
        {
            double acc = 0.0;
            for (int k = 0; k < r; ++k) acc += (resid[k] * w[k]) * AT(tmp, r, k, b);
            if (assign)
                f[b] = acc;
            else
                f[b] += acc;
        }
    }
}

/* c = E^T x0 + f   (costFunctions.cpp:71,80,107,202) ; add != 0 => c += (MixedCost, :213) */
static void cost_c_from_Ef(int nx, int U, const double* E, const double* f, const double* x0, double* c, int add)
{
    for (int b = 0; b < U; ++b) {
        double acc = 0.0;
        for (int a = 0; a < nx; ++a) acc += AT(E, nx, a, b) * x0[a];
        if (add)
            c[b] += acc + f[b];
        else
            c[b] = acc + f[b];
    }
}

/* tmp (r x U) = M (r x kdim) * Psi rows [row0, row0+kdim)  */
static void mat_times_rows(int r, int kdim, const double* M, int ldm, const double* P, int ldp, int row0, int cols,
    double* out)
{
    for (int j = 0; j < cols; ++j)
        for (int i = 0; i < r; ++i) {
            double acc = 0.0;
            for (int k = 0; k < kdim; ++k) acc += AT(M, ldm, i, k) * AT(P, ldp, row0 + k, j);
            AT(out, r, i, j) = acc;
        }
}

static int cost_update(const or_cost_t* cf, const psys_t* ps, double* Q, double* c, double* E, double* f)
{
    const int nx = ps->nx, nu = ps->nu, N = ps->N, X = ps->X, U = ps->U;
    const int r = cf->rows;
    /* CostFunction::initializeCost resizes (costFunctions.cpp:24-30); per-step entries zero them (:52-55). */
    memset(Q, 0, sizeof(double) * (size_t)U * U);
    memset(c, 0, sizeof(double) * (size_t)U);
    memset(E, 0, sizeof(double) * (size_t)nx * U);
    memset(f, 0, sizeof(double) * (size_t)U);
    if (r <= 0) return OR_ERR_DOMAIN;

    switch (cf->kind) {
    case OR_COST_TRAJECTORY: { /* costFunctions.cpp:44-82 */
        int full;
        if (cf->m_cols == nx)
            full = 0;
        else if (cf->m_cols == X)
            full = 1;
        else
            return OR_ERR_DOMAIN; /* :58-60 */
        double* tmp = dzeros((size_t)r * U);
        double* Mphi = dzeros((size_t)r * nx);
        double* resid = dzeros((size_t)r);
        if (full) { /* :65-71 */
            mat_times_rows(r, X, cf->M, r, ps->Psi, X, 0, U, tmp);
            mat_times_rows(r, X, cf->M, r, ps->Phi, X, 0, nx, Mphi);
            for (int i = 0; i < r; ++i) {
                double acc = 0.0;
                for (int k = 0; k < X; ++k) acc += AT(cf->M, r, i, k) * ps->xi[k];
                resid[i] = acc - cf->p[i];
            }
            cost_accumulate(r, U, nx, tmp, cf->w, Mphi, resid, Q, E, f, 1);
        } else { /* :73-79 */
            for (int s = 0; s <= N; ++s) {
                mat_times_rows(r, nx, cf->M, r, ps->Psi, X, s * nx, U, tmp);
                mat_times_rows(r, nx, cf->M, r, ps->Phi, X, s * nx, nx, Mphi);
                for (int i = 0; i < r; ++i) {
                    double acc = 0.0;
                    for (int k = 0; k < nx; ++k) acc += AT(cf->M, r, i, k) * ps->xi[s * nx + k];
                    resid[i] = acc - cf->p[i];
                }
                cost_accumulate(r, U, nx, tmp, cf->w, Mphi, resid, Q, E, f, 0);
            }
        }
        cost_c_from_Ef(nx, U, E, f, ps->x0, c, 0);
        free(tmp);
        free(Mphi);
        free(resid);
        return OR_OK;
    }
    case OR_COST_TARGET: { /* costFunctions.cpp:88-108 */
        if (cf->m_cols != nx) return OR_ERR_DOMAIN; /* :95-97 */
        double* tmp = dzeros((size_t)r * U);
        double* Mphi = dzeros((size_t)r * nx);
        double* resid = dzeros((size_t)r);
        mat_times_rows(r, nx, cf->M, r, ps->Psi, X, N * nx, U, tmp); /* bottomRows(xDim) */
        mat_times_rows(r, nx, cf->M, r, ps->Phi, X, N * nx, nx, Mphi);
        for (int i = 0; i < r; ++i) {
            double acc = 0.0;
            for (int k = 0; k < nx; ++k) acc += AT(cf->M, r, i, k) * ps->xi[N * nx + k];
            resid[i] = acc - cf->p[i];
        }
        cost_accumulate(r, U, nx, tmp, cf->w, Mphi, resid, Q, E, f, 1);
        cost_c_from_Ef(nx, U, E, f, ps->x0, c, 0);
        free(tmp);
        free(Mphi);
        free(resid);
        return OR_OK;
    }
    case OR_COST_CONTROL: { /* costFunctions.cpp:122-158 */
        int full;
        if (cf->n_cols == nu)
            full = 0;
        else if (cf->n_cols == U)
            full = 1;
        else
            return OR_ERR_DOMAIN; /* :134-136 */
        const int nc = cf->n_cols;
        double* mat = dzeros((size_t)nc * nc);
        double* vec = dzeros((size_t)nc);
        for (int b = 0; b < nc; ++b) {
            for (int a = 0; a < nc; ++a) {
                double acc = 0.0;
                for (int k = 0; k < r; ++k) acc += (AT(cf->N, r, k, a) * cf->w[k]) * AT(cf->N, r, k, b);
                AT(mat, nc, a, b) = acc;
            }
            double acc = 0.0;
            for (int k = 0; k < r; ++k) acc += ((-cf->p[k]) * cf->w[k]) * AT(cf->N, r, k, b);
            vec[b] = acc;
        }
        if (full) { /* :141-146 */
            memcpy(Q, mat, sizeof(double) * (size_t)U * U);
            memcpy(f, vec, sizeof(double) * (size_t)U);
            memcpy(c, vec, sizeof(double) * (size_t)U);
        } else { /* :148-156 */
            for (int s = 0; s < N; ++s) {
                for (int b = 0; b < nu; ++b) {
                    for (int a = 0; a < nu; ++a) AT(Q, U, s * nu + a, s * nu + b) = AT(mat, nu, a, b);
                    f[s * nu + b] = vec[b];
                    c[s * nu + b] = vec[b];
                }
            }
        }
        free(mat);
        free(vec);
        return OR_OK;
    }
    case OR_COST_MIXED: { /* costFunctions.cpp:173-215 */
        int full;
        if (cf->m_cols == nx && cf->n_cols == nu)
            full = 0;
        else if (cf->m_cols == X && cf->n_cols == U)
            full = 1;
        else
            return OR_ERR_DOMAIN; /* :190-192 */
        double* tmp = dzeros((size_t)r * U);
        double* Mphi = dzeros((size_t)r * nx);
        double* resid = dzeros((size_t)r);
        if (full) { /* :197-203 */
            mat_times_rows(r, X, cf->M, r, ps->Psi, X, 0, U, tmp);
            for (int j = 0; j < U; ++j)
                for (int i = 0; i < r; ++i) AT(tmp, r, i, j) += AT(cf->N, r, i, j);
            mat_times_rows(r, X, cf->M, r, ps->Phi, X, 0, nx, Mphi);
            for (int i = 0; i < r; ++i) {
                double acc = 0.0;
                for (int k = 0; k < X; ++k) acc += AT(cf->M, r, i, k) * ps->xi[k];
                resid[i] = acc - cf->p[i];
            }
            cost_accumulate(r, U, nx, tmp, cf->w, Mphi, resid, Q, E, f, 1);
            cost_c_from_Ef(nx, U, E, f, ps->x0, c, 0);
        } else { /* :205-213 */
            for (int s = 0; s < N; ++s) {
                mat_times_rows(r, nx, cf->M, r, ps->Psi, X, s * nx, U, tmp);
                for (int j = 0; j < nu; ++j)
                    for (int i = 0; i < r; ++i) AT(tmp, r, i, s * nu + j) += AT(cf->N, r, i, j);
                mat_times_rows(r, nx, cf->M, r, ps->Phi, X, s * nx, nx, Mphi);
                for (int i = 0; i < r; ++i) {
                    double acc = 0.0;
                    for (int k = 0; k < nx; ++k) acc += AT(cf->M, r, i, k) * ps->xi[s * nx + k];
                    resid[i] = acc - cf->p[i];
                }
                cost_accumulate(r, U, nx, tmp, cf->w, Mphi, resid, Q, E, f, 0);
            }
            cost_c_from_Ef(nx, U, E, f, ps->x0, c, 1); /* c_ += ... on a zeroed c_ (first solve) */
        }
        free(tmp);
        free(Mphi);
        free(resid);
        return OR_OK;
    }
    default:
        return OR_ERR_DOMAIN;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Constraints -- src/constraints.cpp.  nrConstr (initializeConstraint) and update.
 * ---------------------------------------------------------------------------------------------- */
static int is_neg_inf(double v) { return isinf(v) && v < 0; }
static int is_pos_inf(double v) { return isinf(v) && v > 0; }

/* returns nrConstr_ or a negative error */
static int cstr_nr(const or_cstr_t* cs, int nx, int nu, int N)
{
    const int X = nx * (N + 1), U = nu * N;
    switch (cs->kind) {
    case OR_CSTR_TRAJECTORY: /* constraints.cpp:45-64 */
        if (cs->e_cols == nx) return cs->rows * (N + 1);
        if (cs->e_cols == X) return cs->rows;
        return OR_ERR_DOMAIN;
    case OR_CSTR_CONTROL: /* :106-135 */
        if (cs->g_cols == nu) return cs->rows * N;
        if (cs->g_cols == U) return cs->rows;
        return OR_ERR_DOMAIN;
    case OR_CSTR_MIXED: /* :171-195 */
        if (cs->e_cols == nx && cs->g_cols == nu) return cs->rows * N;
        if (cs->e_cols == X && cs->g_cols == U) return cs->rows;
        return OR_ERR_DOMAIN;
    case OR_CSTR_TRAJECTORY_BOUND: { /* constraints.h:248-255 + constraints.cpp:263-282 */
        int nl = 0, nup = 0;
        for (int i = 0; i < cs->rows; ++i) {
            if (!is_neg_inf(cs->lower[i])) ++nl;
            if (!is_pos_inf(cs->upper[i])) ++nup;
        }
        if (cs->rows == nx) return (nl + nup) * (N + 1);
        if (cs->rows == X) return nl + nup;
        return OR_ERR_DOMAIN;
    }
    case OR_CSTR_CONTROL_BOUND: /* :333-357 */
        if (cs->rows == nu) return U;
        if (cs->rows == U) return U;
        return OR_ERR_DOMAIN;
    default:
        return OR_ERR_DOMAIN;
    }
}

/* EqIneq constraints: A (m x U), b (m), Y (m x nx), z (m) */
static int cstr_update(const or_cstr_t* cs, const psys_t* ps, int m, double* A, double* b, double* Y, double* z)
{
    const int nx = ps->nx, nu = ps->nu, N = ps->N, X = ps->X, U = ps->U;
    const int r = cs->rows;
    memset(A, 0, sizeof(double) * (size_t)m * U);
    memset(Y, 0, sizeof(double) * (size_t)m * nx);
    memset(b, 0, sizeof(double) * (size_t)m);
    memset(z, 0, sizeof(double) * (size_t)m);
    switch (cs->kind) {
    case OR_CSTR_TRAJECTORY: { /* constraints.cpp:66-84 */
        if (cs->e_cols == X) { /* :68-73 */
            for (int j = 0; j < U; ++j)
                for (int i = 0; i < r; ++i) {
                    double acc = 0.0;
                    for (int k = 0; k < X; ++k) acc += AT(cs->E, r, i, k) * AT(ps->Psi, X, k, j);
                    AT(A, m, i, j) = acc;
                }
            for (int j = 0; j < nx; ++j)
                for (int i = 0; i < r; ++i) {
                    double acc = 0.0;
                    for (int k = 0; k < X; ++k) acc += AT(cs->E, r, i, k) * AT(ps->Phi, X, k, j);
                    AT(Y, m, i, j) = acc;
                }
            for (int i = 0; i < r; ++i) {
                double acc = 0.0;
                for (int k = 0; k < X; ++k) acc += AT(cs->E, r, i, k) * ps->xi[k];
                z[i] = cs->f[i] - acc;
                double yx = 0.0;
                for (int k = 0; k < nx; ++k) yx += AT(Y, m, i, k) * ps->x0[k];
                b[i] = z[i] - yx;
            }
        } else { /* :75-82 */
            for (int s = 0; s <= N; ++s) {
                for (int j = 0; j < U; ++j)
                    for (int i = 0; i < r; ++i) {
                        double acc = 0.0;
                        for (int k = 0; k < nx; ++k) acc += AT(cs->E, r, i, k) * AT(ps->Psi, X, s * nx + k, j);
                        AT(A, m, s * r + i, j) = acc;
                    }
                for (int j = 0; j < nx; ++j)
                    for (int i = 0; i < r; ++i) {
                        double acc = 0.0;
                        for (int k = 0; k < nx; ++k) acc += AT(cs->E, r, i, k) * AT(ps->Phi, X, s * nx + k, j);
                        AT(Y, m, s * r + i, j) = acc;
                    }
                for (int i = 0; i < r; ++i) {
                    double acc = 0.0;
                    for (int k = 0; k < nx; ++k) acc += AT(cs->E, r, i, k) * ps->xi[s * nx + k];
                    z[s * r + i] = cs->f[i] - acc;
                    double yx = 0.0;
                    for (int k = 0; k < nx; ++k) yx += AT(Y, m, s * r + i, k) * ps->x0[k];
                    b[s * r + i] = z[s * r + i] - yx;
                }
            }
        }
        return OR_OK;
    }
    case OR_CSTR_CONTROL: { /* constraints.cpp:106-148 */
        if (cs->g_cols == U) { /* :121-125: A = G, b = f */
            memcpy(A, cs->G, sizeof(double) * (size_t)m * U);
            memcpy(b, cs->f, sizeof(double) * (size_t)m);
        } else { /* :139-143 */
            for (int s = 0; s < N; ++s)
                for (int i = 0; i < r; ++i) {
                    for (int j = 0; j < nu; ++j) AT(A, m, s * r + i, s * nu + j) = AT(cs->G, r, i, j);
                    b[s * r + i] = cs->f[i];
                }
        }
        memcpy(z, b, sizeof(double) * (size_t)m); /* :131-132, :145-146: Y = 0, z = b */
        return OR_OK;
    }
    case OR_CSTR_MIXED: { /* constraints.cpp:197-226 */
        if (cs->e_cols == X) { /* :199-204 */
            for (int j = 0; j < U; ++j)
                for (int i = 0; i < r; ++i) {
                    double acc = 0.0;
                    for (int k = 0; k < X; ++k) acc += AT(cs->E, r, i, k) * AT(ps->Psi, X, k, j);
                    AT(A, m, i, j) = acc + AT(cs->G, r, i, j);
                }
            for (int j = 0; j < nx; ++j)
                for (int i = 0; i < r; ++i) {
                    double acc = 0.0;
                    for (int k = 0; k < X; ++k) acc += AT(cs->E, r, i, k) * AT(ps->Phi, X, k, j);
                    AT(Y, m, i, j) = acc;
                }
            for (int i = 0; i < r; ++i) {
                double acc = 0.0;
                for (int k = 0; k < X; ++k) acc += AT(cs->E, r, i, k) * ps->xi[k];
                z[i] = cs->f[i] - acc;
                double yx = 0.0;
                for (int k = 0; k < nx; ++k) yx += AT(Y, m, i, k) * ps->x0[k];
                b[i] = z[i] - yx;
            }
        } else {
            /* :209-213 : row-block 0 */
            for (int i = 0; i < r; ++i) {
                for (int j = 0; j < nu; ++j) AT(A, m, i, j) = AT(cs->G, r, i, j);
                for (int j = 0; j < nx; ++j) AT(Y, m, i, j) = AT(cs->E, r, i, j);
                z[i] = cs->f[i];
                double yx = 0.0;
                for (int k = 0; k < nx; ++k) yx += AT(Y, m, i, k) * ps->x0[k];
                b[i] = z[i] - yx;
            }
            for (int s = 1; s < N; ++s) { /* :214-224 */
                for (int j = 0; j < nu; ++j)
                    for (int i = 0; i < r; ++i) {
                        double acc = 0.0;
                        for (int k = 0; k < nx; ++k) acc += AT(cs->E, r, i, k) * AT(ps->Psi, X, s * nx + k, j);
                        AT(A, m, s * r + i, j) = acc;
                    }
                for (int j = 0; j < nx; ++j)
                    for (int i = 0; i < r; ++i) {
                        double acc = 0.0;
                        for (int k = 0; k < nx; ++k) acc += AT(cs->E, r, i, k) * AT(ps->Phi, X, s * nx + k, j);
                        AT(Y, m, s * r + i, j) = acc;
                    }
                for (int jb = 1; jb <= s; ++jb)
                    for (int j = 0; j < nu; ++j)
                        for (int i = 0; i < r; ++i)
                            AT(A, m, s * r + i, jb * nu + j) = AT(A, m, (s - 1) * r + i, (jb - 1) * nu + j);
                for (int i = 0; i < r; ++i) {
                    double acc = 0.0;
                    for (int k = 0; k < nx; ++k) acc += AT(cs->E, r, i, k) * ps->xi[s * nx + k];
                    z[s * r + i] = cs->f[i] - acc;
                    double yx = 0.0;
                    for (int k = 0; k < nx; ++k) yx += AT(Y, m, s * r + i, k) * ps->x0[k];
                    b[s * r + i] = z[s * r + i] - yx;
                }
            }
        }
        return OR_OK;
    }
    case OR_CSTR_TRAJECTORY_BOUND: { /* constraints.cpp:284-315 (reference quirk Q1: lower rows are NOT negated) */
        const int full = (cs->rows == X);
        int line_out = 0;
        for (int pass = 0; pass < 2; ++pass) {
            const double* bound = pass == 0 ? cs->lower : cs->upper;
            for (int s = 0; s <= N; ++s) {
                for (int line = 0; line < cs->rows; ++line) {
                    if (pass == 0 ? is_neg_inf(bound[line]) : is_pos_inf(bound[line])) continue;
                    const int row = line + nx * s;
                    for (int j = 0; j < U; ++j) AT(A, m, line_out, j) = AT(ps->Psi, X, row, j);
                    for (int j = 0; j < nx; ++j) AT(Y, m, line_out, j) = AT(ps->Phi, X, row, j);
                    z[line_out] = bound[line] - ps->xi[row];
                    double yx = 0.0;
                    for (int k = 0; k < nx; ++k) yx += AT(Y, m, line_out, k) * ps->x0[k];
                    b[line_out] = z[line_out] - yx;
                    ++line_out;
                }
                if (full) break; /* :298-300, :312-314 */
            }
        }
        return line_out == m ? OR_OK : OR_ERR_RUNTIME;
    }
    default:
        return OR_ERR_DOMAIN;
    }
}

void or_qp_free(or_qp_t* qp)
{
    if (!qp) return;
    free(qp->Q);
    free(qp->c);
    free(qp->Aeq);
    free(qp->beq);
    free(qp->Aineq);
    free(qp->bineq);
    free(qp->lb);
    free(qp->ub);
    free(qp->Phi);
    free(qp->Psi);
    free(qp->xi);
    memset(qp, 0, sizeof(*qp));
}

/* Shared body of LMPC::updateSystem + LMPC::makeQPForm / InitialStateLMPC::makeQPForm.
 * off = 0 (LMPC) or nx (InitialStateLMPC: decision vector [x0; U]). */
static int build_common(int nx, int nu, int N, const double* A, const double* B, const double* d, const double* x0,
    int ncost, const or_cost_t* costs, int ncstr, const or_cstr_t* cstrs, int initial_state, const double* R,
    const double* rvec, const double* x0lb, const double* x0ub, or_qp_t* out)
{
    memset(out, 0, sizeof(*out));
    if (nx <= 0 || nu <= 0 || N <= 0) return OR_ERR_DOMAIN; /* PreviewSystem.cpp:19-33 */
    const int X = nx * (N + 1), U = nu * N;
    const int off = initial_state ? nx : 0;
    const int nvar = U + off;
    out->nx = nx;
    out->nu = nu;
    out->N = N;
    out->fullX = X;
    out->fullU = U;
    out->nvar = nvar;

    out->Phi = dzeros((size_t)X * nx);
    out->Psi = dzeros((size_t)X * U);
    out->xi = dzeros((size_t)X);
    or_preview_update(nx, nu, N, A, B, d, out->Phi, out->Psi, out->xi); /* LMPC.cpp:233-235 */
    psys_t ps = { nx, nu, N, X, U, out->Phi, out->Psi, out->xi, x0 };

    /* counts (LMPC.cpp:20-35, :173-197): eq / ineq lists in insertion order */
    int neq = 0, nineq = 0;
    int* nr = (int*)calloc((size_t)(ncstr ? ncstr : 1), sizeof(int));
    for (int k = 0; k < ncstr; ++k) {
        int m = cstr_nr(&cstrs[k], nx, nu, N);
        if (m < 0) {
            free(nr);
            return m;
        }
        nr[k] = m;
        if (cstrs[k].kind == OR_CSTR_CONTROL_BOUND) continue;
        if (cstrs[k].kind == OR_CSTR_TRAJECTORY_BOUND || cstrs[k].is_ineq)
            nineq += m;
        else
            neq += m;
    }
    out->neq = neq;
    out->nineq = nineq;

    /* LMPC.cpp:228-230 : Q = 1e-6 I, c = 0 (full decision size for the InitialState variant too) */
    out->Q = dzeros((size_t)nvar * nvar);
    out->c = dzeros((size_t)nvar);
    for (int i = 0; i < nvar; ++i) {
        AT(out->Q, nvar, i, i) = 1.0;
        AT(out->Q, nvar, i, i) *= 1e-6;
    }
    out->Aeq = dzeros((size_t)neq * nvar);
    out->beq = dzeros((size_t)neq);
    out->Aineq = dzeros((size_t)nineq * nvar);
    out->bineq = dzeros((size_t)nineq);
    out->lb = dzeros((size_t)nvar);
    out->ub = dzeros((size_t)nvar);
    for (int i = 0; i < nvar; ++i) { /* LMPC.cpp:207-208 / InitialStateLMPC.cpp:59-60 */
        out->lb[i] = -DBL_MAX;
        out->ub[i] = DBL_MAX;
    }

    int rc = OR_OK;
    /* ---- constraints (LMPC.cpp:240-242 update, :257-279 stacking) ---- */
    int eq_line = 0, ineq_line = 0, bound_line = off; /* InitialStateLMPC.cpp:105 starts at xDim */
    for (int k = 0; k < ncstr && rc == OR_OK; ++k) {
        const or_cstr_t* cs = &cstrs[k];
        const int m = nr[k];
        if (cs->kind == OR_CSTR_CONTROL_BOUND) { /* constraints.cpp:359-367, LMPC.cpp:274-279 */
            if (bound_line + m > nvar) {
                rc = OR_ERR_RUNTIME;
                break;
            }
            for (int i = 0; i < m; ++i) {
                const int src = (cs->rows == U) ? i : (i % nu);
                out->lb[bound_line + i] = cs->lower[src];
                out->ub[bound_line + i] = cs->upper[src];
            }
            bound_line += m;
            continue;
        }
        double* Ak = dzeros((size_t)m * U);
        double* bk = dzeros((size_t)m);
        double* Yk = dzeros((size_t)m * nx);
        double* zk = dzeros((size_t)m);
        rc = cstr_update(cs, &ps, m, Ak, bk, Yk, zk);
        if (rc == OR_OK) {
            const int ineq = (cs->kind == OR_CSTR_TRAJECTORY_BOUND || cs->is_ineq);
            double* Adst = ineq ? out->Aineq : out->Aeq;
            double* bdst = ineq ? out->bineq : out->beq;
            const int ld = ineq ? nineq : neq;
            const int line = ineq ? ineq_line : eq_line;
            for (int i = 0; i < m; ++i) {
                if (initial_state) { /* InitialStateLMPC.cpp:88-102 : [Y A], rhs z */
                    for (int j = 0; j < nx; ++j) AT(Adst, ld, line + i, j) = AT(Yk, m, i, j);
                    bdst[line + i] = zk[i];
                } else { /* LMPC.cpp:259-271 : A, rhs b */
                    bdst[line + i] = bk[i];
                }
                for (int j = 0; j < U; ++j) AT(Adst, ld, line + i, off + j) = AT(Ak, m, i, j);
            }
            if (ineq)
                ineq_line += m;
            else
                eq_line += m;
        }
        free(Ak);
        free(bk);
        free(Yk);
        free(zk);
    }

    /* ---- costs (LMPC.cpp:245-247 update, :252-255 / InitialStateLMPC.cpp:80-84 sum) ---- */
    double* Qk = dzeros((size_t)U * U);
    double* ck = dzeros((size_t)U);
    double* Ek = dzeros((size_t)nx * U);
    double* fk = dzeros((size_t)U);
    for (int k = 0; k < ncost && rc == OR_OK; ++k) {
        rc = cost_update(&costs[k], &ps, Qk, ck, Ek, fk);
        if (rc != OR_OK) break;
        for (int j = 0; j < U; ++j) {
            for (int i = 0; i < U; ++i) AT(out->Q, nvar, off + i, off + j) += AT(Qk, U, i, j);
            if (initial_state) {
                for (int i = 0; i < nx; ++i) AT(out->Q, nvar, i, off + j) += AT(Ek, nx, i, j);
                out->c[off + j] += fk[j];
            } else {
                out->c[j] += ck[j];
            }
        }
    }
    free(Qk);
    free(ck);
    free(Ek);
    free(fk);

    if (rc == OR_OK && initial_state) {
        /* InitialStateLMPC.cpp:113-121 */
        double* Qbr = dzeros((size_t)U * U); /* bottomRightCorner */
        for (int j = 0; j < U; ++j)
            for (int i = 0; i < U; ++i) AT(Qbr, U, i, j) = AT(out->Q, nvar, nx + i, nx + j);
        /* Q.inverse(): Eigen uses partial-pivoting LU for dynamic sizes */
        double* inv = dzeros((size_t)U * U);
        int* piv = (int*)calloc((size_t)U, sizeof(int));
        int singular = 0;
        for (int i = 0; i < U; ++i) piv[i] = i;
        for (int k = 0; k < U; ++k) { /* LU in place, row pivoting */
            int pr = k;
            double best = fabs(AT(Qbr, U, k, k));
            for (int i = k + 1; i < U; ++i)
                if (fabs(AT(Qbr, U, i, k)) > best) {
                    best = fabs(AT(Qbr, U, i, k));
                    pr = i;
                }
            if (best == 0.0) {
                singular = 1;
                break;
            }
            if (pr != k) {
                for (int j = 0; j < U; ++j) {
                    double t = AT(Qbr, U, k, j);
                    AT(Qbr, U, k, j) = AT(Qbr, U, pr, j);
                    AT(Qbr, U, pr, j) = t;
                }
                int t = piv[k];
                piv[k] = piv[pr];
                piv[pr] = t;
            }
            for (int i = k + 1; i < U; ++i) {
                AT(Qbr, U, i, k) /= AT(Qbr, U, k, k);
                const double l = AT(Qbr, U, i, k);
                for (int j = k + 1; j < U; ++j) AT(Qbr, U, i, j) -= l * AT(Qbr, U, k, j);
            }
        }
        if (!singular) {
            for (int col = 0; col < U; ++col) { /* solve LU x = P e_col */
                double* xcol = &AT(inv, U, 0, col);
                for (int i = 0; i < U; ++i) xcol[i] = (piv[i] == col) ? 1.0 : 0.0;
                for (int i = 0; i < U; ++i)
                    for (int k = 0; k < i; ++k) xcol[i] -= AT(Qbr, U, i, k) * xcol[k];
                for (int i = U - 1; i >= 0; --i) {
                    for (int k = i + 1; k < U; ++k) xcol[i] -= AT(Qbr, U, i, k) * xcol[k];
                    xcol[i] /= AT(Qbr, U, i, i);
                }
            }
            /* bottomLeft = E^T ; topLeft = R + E Q^-1 E^T */
            double* EQi = dzeros((size_t)nx * U);
            for (int j = 0; j < U; ++j)
                for (int i = 0; i < nx; ++i) {
                    double acc = 0.0;
                    for (int k = 0; k < U; ++k) acc += AT(out->Q, nvar, i, nx + k) * AT(inv, U, k, j);
                    AT(EQi, nx, i, j) = acc;
                }
            for (int j = 0; j < nx; ++j)
                for (int i = 0; i < nx; ++i) {
                    double acc = 0.0;
                    for (int k = 0; k < U; ++k) acc += AT(EQi, nx, i, k) * AT(out->Q, nvar, j, nx + k);
                    AT(out->Q, nvar, i, j) = AT(R, nx, i, j) + acc;
                }
            for (int j = 0; j < nx; ++j)
                for (int i = 0; i < U; ++i) AT(out->Q, nvar, nx + i, j) = AT(out->Q, nvar, j, nx + i);
            free(EQi);
        } else {
            rc = OR_ERR_RUNTIME;
        }
        for (int i = 0; i < nx; ++i) {
            out->c[i] = rvec[i];
            out->lb[i] = x0lb[i];
            out->ub[i] = x0ub[i];
        }
        free(Qbr);
        free(inv);
        free(piv);
    }
    free(nr);
    if (rc != OR_OK) or_qp_free(out);
    return rc;
}

int or_lmpc_build(int nx, int nu, int N, const double* A, const double* B, const double* d, const double* x0,
    int ncost, const or_cost_t* costs, int ncstr, const or_cstr_t* cstrs, or_qp_t* out)
{
    return build_common(nx, nu, N, A, B, d, x0, ncost, costs, ncstr, cstrs, 0, NULL, NULL, NULL, NULL, out);
}

int or_islmpc_build(int nx, int nu, int N, const double* A, const double* B, const double* d, const double* x0,
    int ncost, const or_cost_t* costs, int ncstr, const or_cstr_t* cstrs, const double* R, const double* r,
    const double* x0lb, const double* x0ub, or_qp_t* out)
{
    return build_common(nx, nu, N, A, B, d, x0, ncost, costs, ncstr, cstrs, 1, R, r, x0lb, x0ub, out);
}

/* ------------------------------------------------------------------------------------------------
 * Goldfarb-Idnani dual active-set solver with the semantics of quadprog's qpgen2, which eigen-quadprog's
 * Eigen::QuadProgDense wraps (NOT in /root/reference: third-party, unpinned; restated from the published
 * algorithm -- D. Goldfarb, A. Idnani, Math. Prog. 27 (1983) 1-33 -- and qpgen2's documented behaviour).
 *   min  -dvec^T x + 1/2 x^T D x   s.t.  amat(:,i)^T x  =  bvec(i), i <  meq
 *                                          amat(:,i)^T x >= bvec(i), i >= meq
 * dmat: n x n (upper triangle used; destroyed -> holds J = R^-1).  amat: n x q column-major (eq columns may be
 * sign-flipped in place).  Returns ierr in {0,1,2}.
 * ---------------------------------------------------------------------------------------------- */
static int gi_qpgen2(int n, int q, int meq, double* dmat, double* dvec, double* amat, double* bvec, double* sol,
    int* iter)
{
    const int r = n < q ? n : q;
    int ierr = 0;
    /* smallest number such that 1 + 0.1*vsmall > 1 (Powell's ZQPCVX device) */
    volatile double vsmall = 1.0e-60, tmpa, tmpb;
    do {
        vsmall = vsmall + vsmall;
        tmpa = 1.0 + 0.1 * vsmall;
        tmpb = 1.0 + 0.2 * vsmall;
    } while (tmpa <= 1.0 || tmpb <= 1.0);

    double* dv = dzeros((size_t)n); /* d = J^T n+ */
    double* zv = dzeros((size_t)n); /* z */
    double* rv = dzeros((size_t)r + 1); /* r */
    double* uv = dzeros((size_t)r + 2); /* Lagrange multipliers of the active set */
    double* rm = dzeros((size_t)r * (r + 1) / 2 + 1); /* packed upper-triangular R */
    double* sv = dzeros((size_t)q + 1); /* constraint slacks */
    double* nbv = dzeros((size_t)q + 1); /* column norms */
    int* iact = (int*)calloc((size_t)q + 1, sizeof(int));
    int nact = 0;
    iter[0] = 0;
    iter[1] = 0;

    /* --- Cholesky D = R^T R on the upper triangle (LINPACK dpofa) --- */
    for (int j = 0; j < n; ++j) {
        double s = 0.0;
        for (int k = 0; k < j; ++k) {
            double t = AT(dmat, n, k, j);
            for (int i = 0; i < k; ++i) t -= AT(dmat, n, i, k) * AT(dmat, n, i, j);
            t = t / AT(dmat, n, k, k);
            AT(dmat, n, k, j) = t;
            s += t * t;
        }
        s = AT(dmat, n, j, j) - s;
        if (s <= 0.0) {
            ierr = 2;
            goto done;
        }
        AT(dmat, n, j, j) = sqrt(s);
    }
    /* --- unconstrained minimiser: solve R^T R x = dvec (dposl) --- */
    for (int k = 0; k < n; ++k) {
        double t = 0.0;
        for (int i = 0; i < k; ++i) t += AT(dmat, n, i, k) * dvec[i];
        dvec[k] = (dvec[k] - t) / AT(dmat, n, k, k);
    }
    for (int k = n - 1; k >= 0; --k) {
        dvec[k] = dvec[k] / AT(dmat, n, k, k);
        const double t = -dvec[k];
        for (int i = 0; i < k; ++i) dvec[i] += t * AT(dmat, n, i, k);
    }
    /* --- J = R^-1 in the upper triangle (dpori) --- */
    for (int k = 0; k < n; ++k) {
        AT(dmat, n, k, k) = 1.0 / AT(dmat, n, k, k);
        const double t = -AT(dmat, n, k, k);
        for (int i = 0; i < k; ++i) AT(dmat, n, i, k) *= t;
        for (int j = k + 1; j < n; ++j) {
            const double tt = AT(dmat, n, k, j);
            AT(dmat, n, k, j) = 0.0;
            for (int i = 0; i <= k; ++i) AT(dmat, n, i, j) += tt * AT(dmat, n, i, k);
        }
    }
    for (int j = 0; j < n; ++j) {
        sol[j] = dvec[j];
        for (int i = j + 1; i < n; ++i) AT(dmat, n, i, j) = 0.0;
    }
    /* column norms */
    for (int i = 0; i < q; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += AT(amat, n, j, i) * AT(amat, n, j, i);
        nbv[i] = sqrt(s);
    }

    for (;;) {
        /* ---- step 1: pick the most violated constraint ---- */
        iter[0] += 1;
        for (int i = 0; i < q; ++i) {
            double s = -bvec[i];
            for (int j = 0; j < n; ++j) s += AT(amat, n, j, i) * sol[j];
            if (fabs(s) < vsmall) s = 0.0;
            if (i >= meq) {
                sv[i] = s;
            } else {
                sv[i] = -fabs(s);
                if (s > 0.0) {
                    for (int j = 0; j < n; ++j) AT(amat, n, j, i) = -AT(amat, n, j, i);
                    bvec[i] = -bvec[i];
                }
            }
        }
        for (int i = 0; i < nact; ++i) sv[iact[i]] = 0.0;
        int nvl = -1;
        double temp = 0.0;
        for (int i = 0; i < q; ++i) {
            if (sv[i] < temp * nbv[i]) {
                nvl = i;
                temp = sv[i] / nbv[i];
            }
        }
        if (nvl < 0) goto done; /* optimal */

        int it1 = 0;
        for (;;) {
            /* ---- step 2a: d = J^T n+ ---- */
            for (int i = 0; i < n; ++i) {
                double s = 0.0;
                for (int j = 0; j < n; ++j) s += AT(dmat, n, j, i) * AT(amat, n, j, nvl);
                dv[i] = s;
            }
            /* z = J2 d2 */
            for (int i = 0; i < n; ++i) zv[i] = 0.0;
            for (int j = nact; j < n; ++j)
                for (int i = 0; i < n; ++i) zv[i] += AT(dmat, n, i, j) * dv[j];
            /* r = R^-1 d1 ; note positive entries among inequalities */
            int t1inf = 1;
            for (int i = nact - 1; i >= 0; --i) {
                double s = dv[i];
                /* R(i,j) for j>i lives at rm[j*(j+1)/2 + i] */
                for (int j = i + 1; j < nact; ++j) s -= rm[(size_t)j * (j + 1) / 2 + i] * rv[j];
                s = s / rm[(size_t)i * (i + 1) / 2 + i];
                rv[i] = s;
                if (iact[i] < meq) continue;
                if (s <= 0.0) continue;
                t1inf = 0;
                it1 = i;
            }
            double t1 = 0.0;
            if (!t1inf) {
                t1 = uv[it1] / rv[it1];
                for (int i = 0; i < nact; ++i) {
                    if (iact[i] < meq) continue;
                    if (rv[i] <= 0.0) continue;
                    const double tq = uv[i] / rv[i];
                    if (tq < t1) {
                        t1 = tq;
                        it1 = i;
                    }
                }
            }
            double zz = 0.0;
            for (int i = 0; i < n; ++i) zz += zv[i] * zv[i];
            int drop = 0;
            if (fabs(zz) <= vsmall) {
                /* no primal step possible */
                if (t1inf) {
                    ierr = 1; /* infeasible */
                    goto done;
                }
                for (int i = 0; i < nact; ++i) uv[i] -= t1 * rv[i];
                uv[nact] += t1;
                drop = 1;
            } else {
                double zn = 0.0;
                for (int i = 0; i < n; ++i) zn += zv[i] * AT(amat, n, i, nvl);
                double tt = -sv[nvl] / zn;
                int t2min = 1;
                if (!t1inf && t1 < tt) {
                    tt = t1;
                    t2min = 0;
                }
                for (int i = 0; i < n; ++i) sol[i] += tt * zv[i];
                for (int i = 0; i < nact; ++i) uv[i] -= tt * rv[i];
                uv[nact] += tt;
                if (t2min) {
                    /* full step: add constraint nvl, update J and R */
                    iact[nact] = nvl;
                    nact += 1;
                    size_t l = (size_t)(nact - 1) * nact / 2; /* start of column nact-1 of R */
                    for (int i = 0; i < nact - 1; ++i) rm[l + i] = dv[i];
                    l += (size_t)(nact - 1);
                    if (nact == n) {
                        rm[l] = dv[n - 1];
                    } else {
                        for (int i = n - 1; i >= nact; --i) {
                            if (dv[i] == 0.0) continue;
                            double gc = fmax(fabs(dv[i - 1]), fabs(dv[i]));
                            double gs = fmin(fabs(dv[i - 1]), fabs(dv[i]));
                            double tg = copysign(gc * sqrt(1.0 + (gs / gc) * (gs / gc)), dv[i - 1]);
                            gc = dv[i - 1] / tg;
                            gs = dv[i] / tg;
                            if (gc == 1.0) continue;
                            if (gc == 0.0) {
                                dv[i - 1] = gs * tg;
                                for (int j = 0; j < n; ++j) {
                                    const double t = AT(dmat, n, j, i - 1);
                                    AT(dmat, n, j, i - 1) = AT(dmat, n, j, i);
                                    AT(dmat, n, j, i) = t;
                                }
                            } else {
                                dv[i - 1] = tg;
                                const double nu_ = gs / (1.0 + gc);
                                for (int j = 0; j < n; ++j) {
                                    const double t = gc * AT(dmat, n, j, i - 1) + gs * AT(dmat, n, j, i);
                                    AT(dmat, n, j, i) = nu_ * (AT(dmat, n, j, i - 1) + t) - AT(dmat, n, j, i);
                                    AT(dmat, n, j, i - 1) = t;
                                }
                            }
                        }
                        rm[l] = dv[nact - 1];
                    }
                    break; /* back to step 1 */
                } else {
                    /* partial step: drop it1, recompute the slack of nvl */
                    double s = -bvec[nvl];
                    for (int j = 0; j < n; ++j) s += sol[j] * AT(amat, n, j, nvl);
                    if (nvl >= meq) {
                        sv[nvl] = s;
                    } else {
                        sv[nvl] = -fabs(s);
                        if (s > 0.0) {
                            for (int j = 0; j < n; ++j) AT(amat, n, j, nvl) = -AT(amat, n, j, nvl);
                            bvec[nvl] = -bvec[nvl];
                        }
                    }
                    drop = 1;
                }
            }
            if (drop) {
                /* ---- drop the it1-th active constraint (0-based position it1) ---- */
                int p = it1; /* position to remove */
                while (p < nact - 1) {
                    /* Givens on rows p, p+1 of R restricted to columns p+1..nact-1, and columns p,p+1 of J */
                    /* R(p,p+1) at rm[(p+1)(p+2)/2 + p], R(p+1,p+1) at rm[(p+1)(p+2)/2 + p+1] */
                    size_t l = (size_t)(p + 1) * (p + 2) / 2; /* start of column p+1 */
                    size_t l1 = l + (size_t)p + 1; /* R(p+1,p+1) */
                    if (rm[l1] != 0.0) {
                        double gc = fmax(fabs(rm[l1 - 1]), fabs(rm[l1]));
                        double gs = fmin(fabs(rm[l1 - 1]), fabs(rm[l1]));
                        double tg = copysign(gc * sqrt(1.0 + (gs / gc) * (gs / gc)), rm[l1 - 1]);
                        gc = rm[l1 - 1] / tg;
                        gs = rm[l1] / tg;
                        if (gc != 1.0) {
                            if (gc == 0.0) {
                                size_t ll = l1;
                                for (int i = p + 1; i < nact; ++i) {
                                    const double t = rm[ll - 1];
                                    rm[ll - 1] = rm[ll];
                                    rm[ll] = t;
                                    ll += (size_t)i + 1;
                                }
                                for (int i = 0; i < n; ++i) {
                                    const double t = AT(dmat, n, i, p);
                                    AT(dmat, n, i, p) = AT(dmat, n, i, p + 1);
                                    AT(dmat, n, i, p + 1) = t;
                                }
                            } else {
                                const double nu_ = gs / (1.0 + gc);
                                size_t ll = l1;
                                for (int i = p + 1; i < nact; ++i) {
                                    const double t = gc * rm[ll - 1] + gs * rm[ll];
                                    rm[ll] = nu_ * (rm[ll - 1] + t) - rm[ll];
                                    rm[ll - 1] = t;
                                    ll += (size_t)i + 1;
                                }
                                for (int i = 0; i < n; ++i) {
                                    const double t = gc * AT(dmat, n, i, p) + gs * AT(dmat, n, i, p + 1);
                                    AT(dmat, n, i, p + 1) = nu_ * (AT(dmat, n, i, p) + t) - AT(dmat, n, i, p + 1);
                                    AT(dmat, n, i, p) = t;
                                }
                            }
                        }
                    }
                    /* shift column p+1 of R (its first p+1 entries) into column p */
                    {
                        size_t src = (size_t)(p + 1) * (p + 2) / 2;
                        size_t dst = (size_t)p * (p + 1) / 2;
                        for (int i = 0; i <= p; ++i) rm[dst + i] = rm[src + i];
                    }
                    uv[p] = uv[p + 1];
                    iact[p] = iact[p + 1];
                    ++p;
                }
                uv[nact - 1] = uv[nact];
                uv[nact] = 0.0;
                iact[nact - 1] = 0;
                nact -= 1;
                iter[1] += 1;
                /* continue inner loop at step 2a with the same nvl */
            }
        }
    }
done:
    free(dv);
    free(zv);
    free(rv);
    free(uv);
    free(rm);
    free(sv);
    free(nbv);
    free(iact);
    return ierr;
}

/* ------------------------------------------------------------------------------------------------
 * QuadProgDenseSolver::SI_problem / SI_solve -- src/QuadProgSolver.cpp:45-72
 * then Eigen::QuadProgDense::solve: Q_=Q, C_=-c, A_ = [Aeq^T, -ineqMat^T], B_ = [beq, -ineqVec].
 * ---------------------------------------------------------------------------------------------- */
int or_quadprog_dense(int n, int neq, int nineq, const double* Q, const double* c, const double* Aeq,
    const double* beq, const double* Aineq, const double* bineq, const double* XL, const double* XU, double* x,
    int* iter)
{
    const int nin = nineq + 2 * n; /* QuadProgSolver.cpp:51 */
    const int q = neq + nin;
    double* dmat = dzeros((size_t)n * n);
    double* dvec = dzeros((size_t)n);
    double* amat = dzeros((size_t)n * q);
    double* bvec = dzeros((size_t)q);
    double* sol = dzeros((size_t)n);
    memcpy(dmat, Q, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; ++i) dvec[i] = -c[i];
    for (int i = 0; i < neq; ++i) {
        for (int j = 0; j < n; ++j) AT(amat, n, j, i) = AT(Aeq, neq, i, j);
        bvec[i] = beq[i];
    }
    /* ineqMat = [Aineq; I; -I], ineqVec = [bineq; XU; -XL]  (QuadProgSolver.cpp:61-69), negated */
    for (int i = 0; i < nineq; ++i) {
        for (int j = 0; j < n; ++j) AT(amat, n, j, neq + i) = -AT(Aineq, nineq, i, j);
        bvec[neq + i] = -bineq[i];
    }
    for (int i = 0; i < n; ++i) {
        AT(amat, n, i, neq + nineq + i) = -1.0;
        bvec[neq + nineq + i] = -XU[i];
        AT(amat, n, i, neq + nineq + n + i) = 1.0;
        bvec[neq + nineq + n + i] = -(-XL[i]);
    }
    int it[2] = { 0, 0 };
    const int fail = gi_qpgen2(n, q, neq, dmat, dvec, amat, bvec, sol, it);
    if (iter) {
        iter[0] = it[0];
        iter[1] = it[1];
    }
    if (fail == 0 || fail == 1) memcpy(x, sol, sizeof(double) * (size_t)n);
    free(dmat);
    free(dvec);
    free(amat);
    free(bvec);
    free(sol);
    return fail;
}

/* LMPC::solve -- src/LMPC.cpp:79-101 ; updateResults :282-286 */
int or_lmpc_solve(int nx, int nu, int N, const double* A, const double* B, const double* d, const double* x0,
    int ncost, const or_cost_t* costs, int ncstr, const or_cstr_t* cstrs, double* control, double* trajectory,
    int* iter)
{
    or_qp_t qp;
    int rc = or_lmpc_build(nx, nu, N, A, B, d, x0, ncost, costs, ncstr, cstrs, &qp);
    if (rc != OR_OK) return rc;
    const int U = qp.fullU, X = qp.fullX;
    double* u = dzeros((size_t)U);
    const int fail = or_quadprog_dense(U, qp.neq, qp.nineq, qp.Q, qp.c, qp.Aeq, qp.beq, qp.Aineq, qp.bineq, qp.lb,
        qp.ub, u, iter);
    if (fail == 0) {
        memcpy(control, u, sizeof(double) * (size_t)U);
        for (int i = 0; i < X; ++i) { /* trajectory = Phi x0 + Psi U + xi */
            double a = 0.0, b = 0.0;
            for (int k = 0; k < nx; ++k) a += AT(qp.Phi, X, i, k) * x0[k];
            for (int k = 0; k < U; ++k) b += AT(qp.Psi, X, i, k) * u[k];
            trajectory[i] = (a + b) + qp.xi[i];
        }
    }
    free(u);
    or_qp_free(&qp);
    return fail;
}

/* InitialStateLMPC::solve ; updateResults -- src/InitialStateLMPC.cpp:124-128 */
int or_islmpc_solve(int nx, int nu, int N, const double* A, const double* B, const double* d, const double* x0,
    int ncost, const or_cost_t* costs, int ncstr, const or_cstr_t* cstrs, const double* R, const double* r,
    const double* x0lb, const double* x0ub, double* control, double* trajectory, double* x0_opt, int* iter)
{
    or_qp_t qp;
    int rc = or_islmpc_build(nx, nu, N, A, B, d, x0, ncost, costs, ncstr, cstrs, R, r, x0lb, x0ub, &qp);
    if (rc != OR_OK) return rc;
    const int U = qp.fullU, X = qp.fullX, nv = qp.nvar;
    double* v = dzeros((size_t)nv);
    const int fail = or_quadprog_dense(nv, qp.neq, qp.nineq, qp.Q, qp.c, qp.Aeq, qp.beq, qp.Aineq, qp.bineq, qp.lb,
        qp.ub, v, iter);
    if (fail == 0) {
        memcpy(x0_opt, v, sizeof(double) * (size_t)nx);
        memcpy(control, v + nx, sizeof(double) * (size_t)U);
        for (int i = 0; i < X; ++i) {
            double a = 0.0, b = 0.0;
            for (int k = 0; k < nx; ++k) a += AT(qp.Phi, X, i, k) * v[k];
            for (int k = 0; k < U; ++k) b += AT(qp.Psi, X, i, k) * v[nx + k];
            trajectory[i] = (a + b) + qp.xi[i];
        }
    }
    free(v);
    or_qp_free(&qp);
    return fail;
}

/* ------------------------------------------------------------------------------------------------
 * Batched CPU driver: one controller per instance, static partition over pthreads (the reference has no
 * shared state between LMPC objects, so this is the legitimate "all host cores" baseline).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int b0, b1, nx, nu, N, ncost, ncstr;
    const double *A, *B, *d, *x0;
    const or_cost_t* costs;
    const or_cstr_t* cstrs;
    double *control, *trajectory;
    int *status, *iter;
    int err;
} batch_job_t;

static void* batch_worker(void* arg)
{
    batch_job_t* j = (batch_job_t*)arg;
    const int nx = j->nx, nu = j->nu, N = j->N;
    const int X = nx * (N + 1), U = nu * N;
    for (int b = j->b0; b < j->b1; ++b) {
        int it[2] = { 0, 0 };
        int rc = or_lmpc_solve(nx, nu, N, j->A + (size_t)b * nx * nx, j->B + (size_t)b * nx * nu,
            j->d + (size_t)b * nx, j->x0 + (size_t)b * nx, j->ncost, j->costs, j->ncstr, j->cstrs,
            j->control + (size_t)b * U, j->trajectory + (size_t)b * X, it);
        j->status[b] = rc;
        j->iter[2 * b] = it[0];
        j->iter[2 * b + 1] = it[1];
        if (rc < 0 && j->err == 0) j->err = rc;
    }
    return NULL;
}

int or_lmpc_solve_batch(int batch, int nthreads, int nx, int nu, int N, const double* A, const double* B,
    const double* d, const double* x0, int ncost, const or_cost_t* costs, int ncstr, const or_cstr_t* cstrs,
    double* control, double* trajectory, int* status, int* iter)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > batch) nthreads = batch > 0 ? batch : 1;
    /* The restatement allocates per solve like the reference does (Eigen temporaries).  Keep glibc from returning
     * that memory to the kernel after every solve: with many threads the mmap/munmap/page-fault traffic serialises
     * on the process-wide mm lock and the "all cores" baseline would measure the allocator, not the algorithm. */
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 16 << 20);
    batch_job_t* jobs = (batch_job_t*)calloc((size_t)nthreads, sizeof(batch_job_t));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    int err = 0;
    for (int t = 0; t < nthreads; ++t) {
        batch_job_t* j = &jobs[t];
        j->b0 = (int)((long long)batch * t / nthreads);
        j->b1 = (int)((long long)batch * (t + 1) / nthreads);
        j->nx = nx;
        j->nu = nu;
        j->N = N;
        j->ncost = ncost;
        j->ncstr = ncstr;
        j->A = A;
        j->B = B;
        j->d = d;
        j->x0 = x0;
        j->costs = costs;
        j->cstrs = cstrs;
        j->control = control;
        j->trajectory = trajectory;
        j->status = status;
        j->iter = iter;
        if (nthreads == 1)
            batch_worker(j);
        else
            pthread_create(&th[t], NULL, batch_worker, j);
    }
    for (int t = 0; t < nthreads; ++t) {
        if (nthreads > 1) pthread_join(th[t], NULL);
        if (jobs[t].err && !err) err = jobs[t].err;
    }
    free(jobs);
    free(th);
    return err;
}
