// copra_hip_qp.hip -- plug-in point 1 of the reference on the device: copra_qp_solve_dense_batch == QuadProgDenseSolver::SI_problem + SI_solve
// (src/QuadProgSolver.cpp:45-72) for a batch of dense QPs, and its kernels (qp_dense.hpp: one QP per wavefront; qp_dense_large.hpp: one per workgroup).
#include "engine.hpp"
#include "qp_dense.hpp"
#include "qp_dense_large.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ __launch_bounds__(64) void copra_qp_dense_kernel(const DensePlan P) { qp_dense_body(P, (int)blockIdx.x); }

// n > 64: one problem per workgroup (thread = row of J), persistent grid over the batch
__global__ __launch_bounds__(kLargeMaxN) void copra_qp_dense_large_kernel(const DensePlan P) { qp_dense_large_body(P); }

// more than four waves per workgroup: 128-VGPR build so that two workgroups share a CU (see copra_lmpc_large_kernel_w4)
__global__ __launch_bounds__(kLargeMaxN, 4) void copra_qp_dense_large_kernel_w4(const DensePlan P) { qp_dense_large_body(P); }

extern "C" {


copra_status_t copra_qp_solve_dense_batch(int batch, int n, int neq, int nineq, const double* Q, const double* c,
    const double* Aeq, const double* beq, const double* Aineq, const double* bineq, const double* XL,
    const double* XU, double* x, int* failv, int* iter, int on_device, void* hip_stream)
{
    if (batch < 0 || n <= 0 || neq < 0 || nineq < 0) // SI_problem(nrVar, nrEq, nrInEq)
        return fail(COPRA_ERR_DOMAIN, "copra_qp_solve_dense_batch: bad problem sizes");
    if (!Q || !c || !XL || !XU || !x || !failv || (neq > 0 && (!Aeq || !beq)) || (nineq > 0 && (!Aineq || !bineq)))
        return fail(COPRA_ERR_ARG, "copra_qp_solve_dense_batch: null argument");
    if (n > kLargeMaxN) return fail(COPRA_ERR_UNSUPPORTED, "dense QP with more than 512 variables is not covered yet");
    if (batch == 0) return COPRA_OK;
    const bool large = n > kWave;
    hipStream_t s = (hipStream_t)hip_stream;
    DensePlan P {};
    P.n = n;
    P.meq = neq;
    P.mineq = nineq;
    P.mgen = neq + nineq;
    P.mtotal = P.mgen + 2 * n; // QuadProgSolver.cpp:51
    P.batch = batch;
    P.vsmall = qpgen2_vsmall();
    P.max_iter = 50 * (n + P.mtotal) + 100;
    size_t lds_bytes;
    if (large) {
        layout_large_solver(P.llds, 0, n, P.mgen, P.meq, P.mtotal);
        lds_bytes = (size_t)P.llds.total * sizeof(double);
    } else {
        (void)layout_lds(P.lds, 0, 0, 0, n, 0, 1, P.mgen, P.meq, P.mtotal, false);
        lds_bytes = (size_t)P.lds.total * sizeof(double);
    }
    if (lds_bytes > 160u * 1024u) return fail(COPRA_ERR_UNSUPPORTED, "dense QP does not fit LDS");
    if (lds_bytes > 48 * 1024) {
        HIP_TRY(hipFuncSetAttribute(large ? reinterpret_cast<const void*>(copra_qp_dense_large_kernel)
                                          : reinterpret_cast<const void*>(copra_qp_dense_kernel),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        if (large)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(copra_qp_dense_large_kernel_w4),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }
    const size_t b = (size_t)batch;
    std::vector<void*> owned;
    auto release = [&]() {
        for (void* p : owned) (void)hipFree(p);
    };
    hipError_t e = hipSuccess;
    auto to_dev = [&](const double* src, size_t count) -> const double* {
        if (on_device) return src;
        double* dptr = nullptr;
        hipError_t r = hipMalloc((void**)&dptr, (count ? count : 1) * sizeof(double));
        if (r == hipSuccess && count) r = hipMemcpyAsync(dptr, src, count * sizeof(double), hipMemcpyHostToDevice, s);
        if (r != hipSuccess && e == hipSuccess) e = r;
        owned.push_back(dptr);
        return dptr;
    };
    P.Q = to_dev(Q, b * n * n);
    P.c = to_dev(c, b * n);
    P.Aeq = to_dev(Aeq, b * neq * n);
    P.beq = to_dev(beq, b * neq);
    P.Aineq = to_dev(Aineq, b * nineq * n);
    P.bineq = to_dev(bineq, b * nineq);
    P.XL = to_dev(XL, b * n);
    P.XU = to_dev(XU, b * n);
    double* dx = x;
    int *dfail = failv, *diter = iter;
    if (!on_device) {
        hipError_t r = hipMalloc((void**)&dx, b * n * sizeof(double));
        if (r == hipSuccess) r = hipMalloc((void**)&dfail, b * sizeof(int));
        if (r == hipSuccess) r = hipMalloc((void**)&diter, b * 2 * sizeof(int));
        if (r != hipSuccess && e == hipSuccess) e = r;
        owned.push_back(dx);
        owned.push_back(dfail);
        owned.push_back(diter);
    } else if (!diter) {
        hipError_t r = hipMalloc((void**)&diter, b * 2 * sizeof(int));
        if (r != hipSuccess && e == hipSuccess) e = r;
        owned.push_back(diter);
    }
    if (e != hipSuccess) {
        release();
        return fail(COPRA_ERR_HIP, std::string("copra_qp_solve_dense_batch: ") + hipGetErrorString(e));
    }
    P.x = dx;
    P.fail = dfail;
    P.iter = diter;
    if (large) {
        const int threads = (n + kWave - 1) & ~(kWave - 1);
        const bool w4 = prefer_w4(default_options(), reinterpret_cast<const void*>(copra_qp_dense_large_kernel),
            reinterpret_cast<const void*>(copra_qp_dense_large_kernel_w4), threads, lds_bytes);
        auto dense_kernel = w4 ? copra_qp_dense_large_kernel_w4 : copra_qp_dense_large_kernel;
        const int grid = large_grid(default_options(), reinterpret_cast<const void*>(dense_kernel), batch, threads, lds_bytes);
        double* ws = nullptr;
        e = hipMalloc((void**)&ws, (size_t)grid * 2 * n * large_ld(n) * sizeof(double));
        if (e != hipSuccess) {
            release();
            return fail(COPRA_ERR_HIP, std::string("copra_qp_solve_dense_batch: ") + hipGetErrorString(e));
        }
        owned.push_back(ws);
        P.ws = ws;
        hipLaunchKernelGGL(dense_kernel, dim3((unsigned)grid), dim3((unsigned)threads), lds_bytes, s, P);
    } else {
        const int pw = default_options().no_packed ? 0 : packed_width(n, 0, false, lds_bytes);
        const hipFunction_t jit = dense_jit_lookup(n, pw ? pw : 64); // (copra_qp_dense_specialise: copra_hip_jit.hip)
        const unsigned per = pw ? 64u / (unsigned)pw : 1u;
        if (jit && (size_t)per * lds_bytes <= 48 * 1024) {
            DensePlan Pj = P;
            void* args[] = { &Pj };
            e = hipModuleLaunchKernel(jit, ((unsigned)batch + per - 1) / per, 1, 1, 64, 1, 1, per * (unsigned)lds_bytes, s, args, nullptr);
        } else if (pw == 16)
            e = packed_dense_launch_w16(P, lds_bytes, s);
        else if (pw == 32)
            e = packed_dense_launch_w32(P, lds_bytes, s);
        else
            hipLaunchKernelGGL(copra_qp_dense_kernel, dim3((unsigned)batch), dim3(64), lds_bytes, s, P);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess && !on_device) {
        e = hipMemcpyAsync(x, dx, b * n * sizeof(double), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(failv, dfail, b * sizeof(int), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && iter) e = hipMemcpyAsync(iter, diter, b * 2 * sizeof(int), hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess && !owned.empty()) e = hipStreamSynchronize(s);
    release();
    if (e != hipSuccess) return fail(COPRA_ERR_HIP, std::string("copra_qp_solve_dense_batch: ") + hipGetErrorString(e));
    return COPRA_OK;
}

} // extern "C"

