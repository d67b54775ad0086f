// copra_hip_setters.hip -- everything of the C ABI (include/copra_hip.h) that hands data in or out of a controller: PreviewSystem::system /
// xInit, per-instance references / right-hand sides / bounds, solver selection, results, timing and profile read-outs, copra_preview_update.
// The solve path itself is copra_hip.hip.
#include <vector>
#include "engine.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>

// PreviewSystem::updateSystem (src/PreviewSystem.cpp:57-74) as matrices, for host-evaluated user subclasses of Constraint /
// CostFunction (copra_preview_update).  One workgroup runs the recursion Phi_i = A Phi_{i-1}, G_i = A G_{i-1} (G_0 = B),
// xi_i = A xi_{i-1} + d; a second launch spreads the first block column over Psi_{i,j} = G_{i-1-j}.
__global__ __launch_bounds__(256) void copra_preview_recursion_kernel(int nx, int nu, int N, const double* A, const double* B,
    const double* d, double* Phi, double* G, double* xi)
{
    const int X = nx * (N + 1), tid = (int)threadIdx.x, T = (int)blockDim.x;
    for (int e = tid; e < nx * nx; e += T) Phi[(e % nx) + (size_t)X * (e / nx)] = (e % nx == e / nx) ? 1.0 : 0.0; // Phi_0 = I (:51)
    for (int e = tid; e < nx * nu; e += T) G[e] = B[e]; // Psi_{1,0} = B (:60)
    for (int e = tid; e < nx; e += T) xi[e] = 0.0;
    __syncthreads();
    for (int i = 1; i <= N; ++i) {
        for (int e = tid; e < nx * (nx + nu + 1); e += T) {
            const int c = e / nx, r = e - c * nx;
            double acc = 0.0;
            if (c < nx) { // Phi_i = A Phi_{i-1} (:59, :64)
                for (int t = 0; t < nx; ++t) acc += A[r + nx * t] * Phi[((i - 1) * nx + t) + (size_t)X * c];
                Phi[(i * nx + r) + (size_t)X * c] = acc;
            } else if (c < nx + nu) { // G_i = A G_{i-1} (:65); G_N is not part of Psi
                if (i < N) {
                    const int cc = c - nx;
                    for (int t = 0; t < nx; ++t) acc += A[r + nx * t] * G[(size_t)(i - 1) * nx * nu + t + nx * cc];
                    G[(size_t)i * nx * nu + r + nx * cc] = acc;
                }
            } else { // xi_i = A xi_{i-1} + d (:61, :70)
                acc = d[r];
                for (int t = 0; t < nx; ++t) acc += A[r + nx * t] * xi[(i - 1) * nx + t];
                xi[i * nx + r] = acc;
            }
        }
        __syncthreads();
    }
}

__global__ void copra_preview_fill_kernel(int nx, int nu, int N, const double* G, double* Psi)
{
    const size_t X = (size_t)nx * (N + 1), U = (size_t)nu * N;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= X * U) return;
    const size_t col = e / X, row = e - col * X;
    const int i = (int)(row / nx), r = (int)(row - (size_t)i * nx), j = (int)(col / nu), c = (int)(col - (size_t)j * nu);
    Psi[e] = (j < i) ? G[(size_t)(i - 1 - j) * nx * nu + r + nx * c] : 0.0; // Psi_{i,j} = A^(i-1-j) B (:66-69), row block 0 is zero
}

// out[b][i] = out[0][i], b >= 1: one reference for every instance (copra_batch_set_cost_reference_all)
__global__ void copra_broadcast_reference_kernel(const double* p, double* out, int rows, long long total)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < total) out[e] = p[e % rows];
}

// out[b][row0 + s * r + i] = f[b][i] for the steps s of one constraint (copra_batch_set_constraint_rhs)
__global__ void copra_scatter_rhs_kernel(const double* f, double* out, int batch, int r, int steps, int row0, int mgen)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)r * steps;
    if (e >= per * batch) return;
    const long long b = e / per, rem = e - b * per;
    out[b * mgen + row0 + rem] = f[b * r + rem % r];
}

extern "C" {


copra_status_t copra_preview_update(int nx, int nu, int N, const double* A, const double* B, const double* d, double* Phi,
    double* Psi, double* xi)
{
    if (nx <= 0 || nu <= 0 || N <= 0) return fail(COPRA_ERR_DOMAIN, "copra_preview_update: dimensions and number of steps must be positive");
    if (!A || !B || !d || !Phi || !Psi || !xi) return fail(COPRA_ERR_ARG, "copra_preview_update: null argument");
    const size_t X = (size_t)nx * (N + 1), U = (size_t)nu * N;
    double *dA = nullptr, *dB = nullptr, *dd = nullptr, *dPhi = nullptr, *dPsi = nullptr, *dxi = nullptr, *dG = nullptr;
    hipError_t e = hipMalloc((void**)&dA, (size_t)nx * nx * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&dB, (size_t)nx * nu * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&dd, (size_t)nx * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&dPhi, X * nx * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&dPsi, X * U * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&dxi, X * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&dG, (size_t)N * nx * nu * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(dA, A, (size_t)nx * nx * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dB, B, (size_t)nx * nu * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dd, d, (size_t)nx * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(copra_preview_recursion_kernel, dim3(1), dim3(256), 0, nullptr, nx, nu, N, dA, dB, dd, dPhi, dG, dxi);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        const size_t total = X * U;
        hipLaunchKernelGGL(copra_preview_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nullptr, nx, nu, N, dG, dPsi);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(Phi, dPhi, X * nx * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(Psi, dPsi, X * U * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(xi, dxi, X * sizeof(double), hipMemcpyDeviceToHost);
    for (double* q : { dA, dB, dd, dPhi, dPsi, dxi, dG }) (void)hipFree(q);
    if (e != hipSuccess) return fail(COPRA_ERR_HIP, std::string("copra_preview_update: ") + hipGetErrorString(e));
    return COPRA_OK;
}

copra_status_t copra_batch_set_warm_start(copra_batch_t* h, int enable)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_set_warm_start: null handle");
    if (h->hp.plan.initial_state || h->hp.large)
        return fail(COPRA_ERR_UNSUPPORTED, "the warm start belongs to the shared-model path (LMPC, at most 64 decision variables)");
    if (!enable) {
        (void)hipFree(h->d_warm);
        h->d_warm = nullptr;
        return COPRA_OK;
    }
    const size_t count = (size_t)(h->hp.plan.batch > 0 ? h->hp.plan.batch : 1) * kWarmCap;
    if (!h->d_warm) HIP_TRY(hipMalloc((void**)&h->d_warm, count * sizeof(int)));
    HIP_TRY(hipMemset(h->d_warm, 0xff, count * sizeof(int))); // every entry -1: the first solve starts cold
    return COPRA_OK;
}

copra_status_t copra_batch_select_solver(copra_batch_t* h, int solver)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_select_solver: null handle");
    if (solver != COPRA_SOLVER_DEFAULT && solver != COPRA_SOLVER_QUADPROG_DENSE && solver != COPRA_SOLVER_RICCATI_IPM)
        return fail(COPRA_ERR_ARG, "copra_batch_select_solver: unknown solver flag");
    if (h->hp.ric_only && solver == COPRA_SOLVER_QUADPROG_DENSE)
        return fail(COPRA_ERR_UNSUPPORTED, "the condensed Goldfarb-Idnani kernels cover at most 512 decision variables (InitialStateLMPC: xDim <= 16)");
    if (solver == COPRA_SOLVER_RICCATI_IPM) {
        const copra_status_t rc = prepare_riccati(h);
        if (rc != COPRA_OK) return rc;
        if (!h->hs.eligible)
            return fail(COPRA_ERR_UNSUPPORTED, "the Riccati interior-point solver needs a stage-wise controller: " + h->hs.why);
        if (!h->hp.large) // the queue of non-converged instances is finished by the workgroup-per-instance kernel
            return fail(COPRA_ERR_UNSUPPORTED, "the Riccati interior-point solver covers controllers with more than 64 decision variables");
    }
    h->solver = solver;
    return COPRA_OK;
}

int copra_batch_solver_info(const copra_batch_t* h)
{
    if (!h) return -1;
    return use_riccati(const_cast<copra_batch_t*>(h)) ? COPRA_SOLVER_RICCATI_IPM : COPRA_SOLVER_QUADPROG_DENSE;
}

copra_status_t copra_device_info(int* n_devices, int* cu_count, char* arch_name, int arch_name_len)
{
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (n_devices) *n_devices = n;
    if (n > 0) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        if (cu_count) *cu_count = prop.multiProcessorCount;
        if (arch_name && arch_name_len > 0) {
            strncpy(arch_name, prop.gcnArchName, (size_t)arch_name_len - 1);
            arch_name[arch_name_len - 1] = 0;
        }
    }
    return COPRA_OK;
}

copra_status_t copra_batch_set_system(copra_batch_t* h, const double* A, const double* B, const double* d,
    const double* x0, int on_device)
{
    if (!h || !A || !B || !d || !x0) return fail(COPRA_ERR_ARG, "copra_batch_set_system: null argument");
    const FusedPlan& P = h->hp.plan;
    const size_t b = (size_t)P.batch;
    h->shared = false; // per-instance systems again (leaves the shared-model fast path)
    h->shared_as_batch = false;
    see_axis_order(h, A, B, on_device != 0);
    if (on_device) {
        h->A = A;
        h->B = B;
        h->d = d;
        h->x0 = x0;
        return COPRA_OK;
    }
    const size_t nA = b * P.nx * P.nx, nB = b * P.nx * P.nu, nd = b * P.nx;
    if (!h->own_A) {
        HIP_TRY(hipMalloc((void**)&h->own_A, (nA ? nA : 1) * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&h->own_B, (nB ? nB : 1) * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&h->own_d, (nd ? nd : 1) * sizeof(double)));
    }
    if (!h->own_x0) HIP_TRY(hipMalloc((void**)&h->own_x0, (nd ? nd : 1) * sizeof(double)));
    HIP_TRY(hipMemcpy(h->own_A, A, nA * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->own_B, B, nB * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->own_d, d, nd * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->own_x0, x0, nd * sizeof(double), hipMemcpyHostToDevice));
    h->A = h->own_A;
    h->B = h->own_B;
    h->d = h->own_d;
    h->x0 = h->own_x0;
    return COPRA_OK;
}

} // extern "C"


// [batch][rows x cols] row-major -> per-instance column-major (what Eigen holds and every kernel here reads)
__global__ void copra_rowmajor_to_colmajor_kernel(const double* __restrict__ src, double* __restrict__ dst, int rows, int cols,
    long long total)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int per = rows * cols;
    const long long inst = e / per;
    const int w = (int)(e - inst * per), j = w / rows, i = w - j * rows; // destination: column j, row i
    dst[e] = src[inst * per + (long long)i * cols + j];
}

extern "C" {


copra_status_t copra_batch_set_system_rowmajor_async(copra_batch_t* h, const double* A, const double* B, const double* d,
    const double* x0, void* hip_stream)
{
    if (!h || !A || !B || !d || !x0) return fail(COPRA_ERR_ARG, "copra_batch_set_system_rowmajor_async: null argument");
    const FusedPlan& P = h->hp.plan;
    const size_t b = (size_t)P.batch;
    const size_t nA = b * P.nx * P.nx, nB = b * P.nx * P.nu, nd = b * P.nx;
    h->shared = false;
    h->shared_as_batch = false;
    if (!h->axis_order_seen && b > 0) { // (the order of the states, from the first system -- once per controller; row-major here: transposed on the host)
        const size_t a1 = (size_t)P.nx * P.nx, b1 = (size_t)P.nx * P.nu;
        std::vector<double> ar(a1), br(b1), ac(a1), bc(b1);
        if (hipMemcpy(ar.data(), A, a1 * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(br.data(), B, b1 * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess) {
            for (int i = 0; i < P.nx; ++i) {
                for (int j = 0; j < P.nx; ++j) ac[(size_t)i + (size_t)P.nx * j] = ar[(size_t)i * P.nx + j];
                for (int c = 0; c < P.nu; ++c) bc[(size_t)i + (size_t)P.nx * c] = br[(size_t)i * P.nu + c];
            }
            see_axis_order(h, ac.data(), bc.data(), false);
        } else {
            (void)hipGetLastError();
        }
    }
    if (!h->own_A) {
        HIP_TRY(hipMalloc((void**)&h->own_A, (nA ? nA : 1) * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&h->own_B, (nB ? nB : 1) * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&h->own_d, (nd ? nd : 1) * sizeof(double)));
    }
    hipStream_t s = (hipStream_t)hip_stream;
    if (nA) hipLaunchKernelGGL(copra_rowmajor_to_colmajor_kernel, dim3((unsigned)((nA + 255) / 256)), dim3(256), 0, s, A, h->own_A, P.nx, P.nx, (long long)nA);
    if (nB) hipLaunchKernelGGL(copra_rowmajor_to_colmajor_kernel, dim3((unsigned)((nB + 255) / 256)), dim3(256), 0, s, B, h->own_B, P.nx, P.nu, (long long)nB);
    HIP_TRY(hipGetLastError());
    h->A = h->own_A;
    h->B = h->own_B;
    h->d = d; // (vectors have no layout: used in place)
    h->x0 = x0;
    return COPRA_OK;
}

int copra_batch_lanes_per_instance(const copra_batch_t* h)
{
    if (!h) return 0;
    if (h->hp.ric_only) return kWave;
    if (h->hp.large) return h->hp.plan.large.threads;
    return h->packed ? h->packed : kWave;
}

copra_status_t copra_batch_set_cost_reference(copra_batch_t* h, int cost_index, const double* p, int on_device)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_set_cost_reference: null handle");
    const FusedPlan& P = h->hp.plan;
    if (cost_index < 0 || cost_index >= (int)h->hp.cost_slot.size()) return fail(COPRA_ERR_ARG, "copra_batch_set_cost_reference: no such cost");
    cost_index = h->hp.cost_slot[(size_t)cost_index]; // (dense costs are not among the kernel-evaluated terms)
    if (cost_index < 0) return fail(COPRA_ERR_UNSUPPORTED, "copra_batch_set_cost_reference: a dense (host-evaluated) cost has no reference p");
    if ((p != nullptr) != (h->cost_p[cost_index] != nullptr)) h->model_dirty = true; // shared model: c0 / C2 change
    if (!p) { // back to the controller-wide reference given at creation
        h->cost_p[cost_index] = nullptr;
        return COPRA_OK;
    }
    if (on_device) {
        h->cost_p[cost_index] = p;
        return COPRA_OK;
    }
    const size_t count = (size_t)(P.batch > 0 ? P.batch : 1) * P.cost[cost_index].prows; // (a reference trajectory: rows x steps per instance)
    if (!h->d_cost_p[cost_index]) HIP_TRY(hipMalloc((void**)&h->d_cost_p[cost_index], count * sizeof(double)));
    HIP_TRY(hipMemcpy(h->d_cost_p[cost_index], p, count * sizeof(double), hipMemcpyHostToDevice));
    h->cost_p[cost_index] = h->d_cost_p[cost_index];
    return COPRA_OK;
}

copra_status_t copra_batch_set_cost_reference_all(copra_batch_t* h, int cost_index, const double* p, int on_device)
{
    if (!h || !p) return fail(COPRA_ERR_ARG, "copra_batch_set_cost_reference_all: null argument");
    const FusedPlan& P = h->hp.plan;
    if (cost_index < 0 || cost_index >= (int)h->hp.cost_slot.size()) return fail(COPRA_ERR_ARG, "copra_batch_set_cost_reference_all: no such cost");
    const int t = h->hp.cost_slot[(size_t)cost_index];
    if (t < 0) return fail(COPRA_ERR_UNSUPPORTED, "copra_batch_set_cost_reference_all: a dense (host-evaluated) cost has no reference p");
    // Every kernel already reads a per-instance reference where one is set: the new reference is written once per instance into the
    // library's own buffer (a broadcast on the device: 66 MB at the headline's batch for a reference trajectory, ~ 10 us) and that
    // path is taken -- nothing that was derived from the creation-time p (tables of the plan builder, the shared model's c0) can go stale.
    const size_t b = (size_t)(P.batch > 0 ? P.batch : 1), rows = (size_t)P.cost[t].prows;
    if (!h->d_cost_p[t]) HIP_TRY(hipMalloc((void**)&h->d_cost_p[t], b * rows * sizeof(double)));
    double* const out = h->d_cost_p[t];
    if (p == out) return fail(COPRA_ERR_ARG, "copra_batch_set_cost_reference_all: p aliases the library's buffer");
    HIP_TRY(hipStreamSynchronize(h->last_stream)); // (a solve that still reads the buffer)
    const double* src = p;
    if (!on_device) {
        HIP_TRY(hipMemcpy(out, p, rows * sizeof(double), hipMemcpyHostToDevice)); // instance 0's slot, then read from there
        src = out;
    }
    const long long first = on_device ? 0 : (long long)rows, total = (long long)(b * rows) - first;
    if (total > 0)
        hipLaunchKernelGGL(copra_broadcast_reference_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->last_stream, src, out + first,
            (int)rows, total);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->last_stream);
    if (e != hipSuccess) return fail(COPRA_ERR_HIP, std::string("copra_batch_set_cost_reference_all: ") + hipGetErrorString(e));
    if (!h->cost_p[t]) h->model_dirty = true; // shared model: c0 / C2 change
    h->cost_p[t] = out;
    return COPRA_OK;
}

copra_status_t copra_batch_set_constraint_rhs(copra_batch_t* h, int cstr_index, const double* f, int on_device)
{
    if (!h || !f) return fail(COPRA_ERR_ARG, "copra_batch_set_constraint_rhs: null argument");
    const FusedPlan& P = h->hp.plan;
    if (cstr_index < 0 || cstr_index >= (int)h->hp.cstr_row0.size() || h->hp.cstr_row0[(size_t)cstr_index] < 0)
        return fail(COPRA_ERR_UNSUPPORTED,
            "copra_batch_set_constraint_rhs: not a Trajectory / Control / Mixed constraint of this controller "
            "(bound constraints: copra_batch_set_control_bounds)");
    const int r = h->hp.cstr_per_step[(size_t)cstr_index], steps = h->hp.cstr_steps[(size_t)cstr_index];
    const int row0 = h->hp.cstr_row0[(size_t)cstr_index];
    const size_t b = (size_t)(P.batch > 0 ? P.batch : 1);
    if (!h->d_row_f_inst) { // first use: every instance starts from the controller-wide right-hand sides
        HIP_TRY(hipMalloc((void**)&h->d_row_f_inst, b * (size_t)P.mgen * sizeof(double)));
        std::vector<double> rep(b * (size_t)P.mgen);
        for (size_t i = 0; i < b; ++i) std::copy(h->hp.row_f.begin(), h->hp.row_f.begin() + P.mgen, rep.begin() + i * P.mgen);
        HIP_TRY(hipMemcpy(h->d_row_f_inst, rep.data(), rep.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    const double* src = f;
    double* tmp = nullptr;
    if (!on_device) {
        HIP_TRY(hipMalloc((void**)&tmp, b * (size_t)r * sizeof(double)));
        hipError_t e = hipMemcpy(tmp, f, b * (size_t)r * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(tmp);
            return fail(COPRA_ERR_HIP, std::string("copra_batch_set_constraint_rhs: ") + hipGetErrorString(e));
        }
        src = tmp;
    }
    const long long total = (long long)P.batch * r * steps;
    if (total > 0) {
        hipLaunchKernelGGL(copra_scatter_rhs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->last_stream, src,
            h->d_row_f_inst, P.batch, r, steps, row0, P.mgen);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->last_stream);
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) return fail(COPRA_ERR_HIP, std::string("copra_batch_set_constraint_rhs: ") + hipGetErrorString(e));
    return COPRA_OK; // (the shared-model factorisation does not depend on right-hand sides)
}

copra_status_t copra_batch_set_control_bounds(copra_batch_t* h, const double* lower, const double* upper, int on_device)
{
    if (!h || !lower || !upper) return fail(COPRA_ERR_ARG, "copra_batch_set_control_bounds: null argument");
    const FusedPlan& P = h->hp.plan;
    const size_t count = (size_t)(P.batch > 0 ? P.batch : 1) * P.n;
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (!h->d_lb_inst) HIP_TRY(hipMalloc((void**)&h->d_lb_inst, count * sizeof(double)));
    if (!h->d_ub_inst) HIP_TRY(hipMalloc((void**)&h->d_ub_inst, count * sizeof(double)));
    HIP_TRY(hipMemcpy(h->d_lb_inst, lower, count * sizeof(double), kind));
    HIP_TRY(hipMemcpy(h->d_ub_inst, upper, count * sizeof(double), kind));
    return COPRA_OK;
}

copra_status_t copra_batch_set_x0(copra_batch_t* h, const double* x0, int on_device)
{
    if (!h || !x0) return fail(COPRA_ERR_ARG, "copra_batch_set_x0: null argument");
    const FusedPlan& P = h->hp.plan;
    if (on_device) {
        h->x0 = x0;
        return COPRA_OK;
    }
    const size_t nd = (size_t)P.batch * P.nx;
    if (!h->own_x0) HIP_TRY(hipMalloc((void**)&h->own_x0, (nd ? nd : 1) * sizeof(double)));
    HIP_TRY(hipMemcpy(h->own_x0, x0, nd * sizeof(double), hipMemcpyHostToDevice));
    h->x0 = h->own_x0;
    return COPRA_OK;
}

copra_status_t copra_batch_set_outputs(copra_batch_t* h, double* control, double* trajectory, int* status, int* iter)
{
    if (!h || !control || !trajectory || !status || !iter)
        return fail(COPRA_ERR_ARG, "copra_batch_set_outputs: null argument");
    // (round-5 advisor: the adaptive controllers read the counters of the LAST solve through pointers to its output buffers; buffers a caller
    //  ROTATES may be gone by then -- what was learnt from the old ones stays, nothing is read from them any more.  Setting the same buffers
    //  again changes nothing.  A caller's buffers must stay valid until the solve that writes them has finished.)
    if (status != h->ext_status || iter != h->ext_iter) {
        h->ad.last_iter = nullptr;
        h->ad.last_status = nullptr;
    }
    h->ext_control = control;
    h->ext_traj = trajectory;
    h->ext_status = status;
    h->ext_iter = iter;
    return COPRA_OK;
}

copra_status_t copra_batch_layout_info(const copra_batch_t* h, int* lds_bytes, int* active_capacity, int* factor_only,
    int* two_tier)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_layout_info: null handle");
    if (lds_bytes) *lds_bytes = (int)h->hp.lds_bytes;
    if (active_capacity) *active_capacity = h->hp.large ? h->hp.plan.n : h->hp.plan.lds.rcap;
    if (factor_only) *factor_only = h->hp.large ? 0 : h->hp.plan.lds.tri;
    if (two_tier) *two_tier = h->hp.two_tier ? 1 : 0;
    return COPRA_OK;
}

copra_status_t copra_batch_synchronize(copra_batch_t* h)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_synchronize: null handle");
    HIP_TRY(hipStreamSynchronize(h->last_stream));
    return COPRA_OK;
}

const double* copra_batch_control_device(const copra_batch_t* h)
{
    return h ? (h->ext_control ? h->ext_control : h->d_control) : nullptr;
}

const double* copra_batch_trajectory_device(const copra_batch_t* h)
{
    return h ? (h->ext_traj ? h->ext_traj : h->d_traj) : nullptr;
}

const int* copra_batch_status_device(const copra_batch_t* h)
{
    return h ? (h->ext_status ? h->ext_status : h->d_status) : nullptr;
}

const int* copra_batch_iter_device(const copra_batch_t* h)
{
    return h ? (h->ext_iter ? h->ext_iter : h->d_iter) : nullptr;
}

copra_status_t copra_batch_get_results(copra_batch_t* h, double* control, double* trajectory, int* status, int* iter)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_get_results: null handle");
    const FusedPlan& P = h->hp.plan;
    const size_t b = (size_t)P.batch;
    if (h->h_results && !h->ext_control && !h->ext_traj && !h->ext_status && !h->ext_iter) {
        // small batch (the single-problem use of copra::LMPC::solve() above all): ONE asynchronous copy of the whole slab into
        // pinned memory behind the solve, one synchronisation -- instead of a stream synchronisation and four blocking copies
        HIP_TRY(hipMemcpyAsync(h->h_results, h->d_results, h->results_bytes, hipMemcpyDeviceToHost, h->last_stream));
        HIP_TRY(hipStreamSynchronize(h->last_stream));
        if (control) std::memcpy(control, h->h_results, b * P.n * sizeof(double));
        if (trajectory) std::memcpy(trajectory, h->h_results + h->off_traj, b * P.X * sizeof(double));
        if (status) std::memcpy(status, h->h_results + h->off_status, b * sizeof(int));
        if (iter) std::memcpy(iter, h->h_results + h->off_iter, b * 2 * sizeof(int));
        return COPRA_OK;
    }
    HIP_TRY(hipStreamSynchronize(h->last_stream));
    if (control) HIP_TRY(hipMemcpy(control, copra_batch_control_device(h), b * P.n * sizeof(double), hipMemcpyDeviceToHost));
    if (trajectory) HIP_TRY(hipMemcpy(trajectory, copra_batch_trajectory_device(h), b * P.X * sizeof(double), hipMemcpyDeviceToHost));
    if (status) HIP_TRY(hipMemcpy(status, copra_batch_status_device(h), b * sizeof(int), hipMemcpyDeviceToHost));
    if (iter) HIP_TRY(hipMemcpy(iter, copra_batch_iter_device(h), b * 2 * sizeof(int), hipMemcpyDeviceToHost));
    return COPRA_OK;
}

copra_status_t copra_batch_set_initial_state_bounds(copra_batch_t* h, const double* x0lb, const double* x0ub,
    int on_device)
{
    if (!h || !x0lb || !x0ub) return fail(COPRA_ERR_ARG, "copra_batch_set_initial_state_bounds: null argument");
    if (!h->hp.plan.initial_state) return fail(COPRA_ERR_RUNTIME, "not an InitialStateLMPC controller");
    const size_t nd = (size_t)h->hp.plan.batch * h->hp.plan.nx;
    if (on_device) {
        h->x0lb = x0lb;
        h->x0ub = x0ub;
        return COPRA_OK;
    }
    if (!h->own_x0lb) {
        HIP_TRY(hipMalloc((void**)&h->own_x0lb, (nd ? nd : 1) * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&h->own_x0ub, (nd ? nd : 1) * sizeof(double)));
    }
    HIP_TRY(hipMemcpy(h->own_x0lb, x0lb, nd * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->own_x0ub, x0ub, nd * sizeof(double), hipMemcpyHostToDevice));
    h->x0lb = h->own_x0lb;
    h->x0ub = h->own_x0ub;
    return COPRA_OK;
}

copra_status_t copra_batch_get_initial_state(copra_batch_t* h, double* x0_opt)
{
    if (!h || !x0_opt) return fail(COPRA_ERR_ARG, "copra_batch_get_initial_state: null argument");
    if (!h->hp.plan.initial_state) return fail(COPRA_ERR_RUNTIME, "not an InitialStateLMPC controller");
    HIP_TRY(hipStreamSynchronize(h->last_stream));
    HIP_TRY(hipMemcpy(x0_opt, h->d_x0opt, (size_t)h->hp.plan.batch * h->hp.plan.nx * sizeof(double), hipMemcpyDeviceToHost));
    return COPRA_OK;
}

copra_status_t copra_batch_qp_sizes(const copra_batch_t* h, int* nvar, int* neq, int* nineq)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_qp_sizes: null handle");
    if (nvar) *nvar = h->hp.plan.initial_state ? h->hp.plan.nx + h->hp.plan.n : h->hp.plan.n;
    if (neq) *neq = h->hp.plan.meq;
    if (nineq) *nineq = h->hp.plan.mineq;
    return COPRA_OK;
}

copra_status_t copra_batch_phase_profile(copra_batch_t* h, int enable, long long* cycles_out)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_phase_profile: null handle");
    const size_t b = (size_t)(h->hp.plan.batch > 0 ? h->hp.plan.batch : 1);
    if (enable && !h->d_prof) {
        HIP_TRY(hipMalloc((void**)&h->d_prof, b * 8 * sizeof(long long)));
        HIP_TRY(hipMemset(h->d_prof, 0, b * 8 * sizeof(long long)));
    }
    if (cycles_out) {
        if (!h->d_prof) return fail(COPRA_ERR_RUNTIME, "copra_batch_phase_profile: profiling was not enabled");
        HIP_TRY(hipStreamSynchronize(h->last_stream));
        HIP_TRY(hipMemcpy(cycles_out, h->d_prof, b * 8 * sizeof(long long), hipMemcpyDeviceToHost));
    }
    if (!enable && h->d_prof) {
        (void)hipFree(h->d_prof);
        h->d_prof = nullptr;
    }
    return COPRA_OK;
}

#ifdef COPRA_FINE_PROFILE
// profiling builds only (libcopra_hip_prof.so): 32 raw shader-clock stamps per instance, -1 = unused
copra_status_t copra_batch_fine_profile(copra_batch_t* h, long long* out)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_fine_profile: null handle");
    const size_t b = (size_t)(h->hp.plan.batch > 0 ? h->hp.plan.batch : 1);
    if (!h->d_prof_fine) {
        HIP_TRY(hipMalloc((void**)&h->d_prof_fine, b * 32 * sizeof(long long)));
        HIP_TRY(hipMemset(h->d_prof_fine, 0xff, b * 32 * sizeof(long long)));
    }
    if (out) {
        HIP_TRY(hipStreamSynchronize(h->last_stream));
        HIP_TRY(hipMemcpy(out, h->d_prof_fine, b * 32 * sizeof(long long), hipMemcpyDeviceToHost));
    }
    return COPRA_OK;
}
#endif

copra_status_t copra_batch_last_solve_seconds(copra_batch_t* h, double* seconds)
{
    if (!h || !seconds) return fail(COPRA_ERR_ARG, "copra_batch_last_solve_seconds: null argument");
    if (!h->timed) return fail(COPRA_ERR_RUNTIME, "copra_batch_last_solve_seconds: no solve has been launched");
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *seconds = (double)ms * 1e-3;
    return COPRA_OK;
}

copra_status_t copra_batch_last_first_tier_seconds(copra_batch_t* h, double* seconds)
{
    if (!h || !seconds) return fail(COPRA_ERR_ARG, "copra_batch_last_first_tier_seconds: null argument");
    if (!h->timed) return fail(COPRA_ERR_RUNTIME, "copra_batch_last_first_tier_seconds: no solve has been launched");
    if (!h->tier_timed) return copra_batch_last_solve_seconds(h, seconds);
    HIP_TRY(hipEventSynchronize(h->evm));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->evm));
    *seconds = (double)ms * 1e-3;
    return COPRA_OK;
}

copra_status_t copra_batch_lane_pass_info(copra_batch_t* h, int* ran, int* finished)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_lane_pass_info: null handle");
    if (ran) *ran = h->ad.axis_ran ? 2 : h->ad.lane_ran ? 1 : 0; // (2: the one-(instance, axis)-per-lane solver, lmpc_axis.hpp)
    if (finished) {
        *finished = 0;
        if (h->ad.lane_ran || h->ad.axis_ran) {
            int left = 0;
            HIP_TRY(hipStreamSynchronize(h->last_stream));
            HIP_TRY(hipMemcpy(&left, (h->ad.axis_ran && !h->ad.axis_quiet) ? h->d_axis_count2 : h->d_lane_count + h->lane_cur, sizeof(int), hipMemcpyDeviceToHost)); // (the solver's second chance included)
            *finished = h->hp.plan.batch - left;
        }
    }
    return COPRA_OK;
}

} // extern "C"

