// Experiment: register-resident row Cholesky + triangular inverse for n = 60, one matrix per wave (lane = row).
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o regchol regchol.hip ;  run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

constexpr int NV = 60;

__device__ __forceinline__ double bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    e = fma(-x * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    return y;
}

// A: [batch][NV*NV] symmetric, row-major.  L out (lower, row-major), X = L^-1 out.
__global__ __launch_bounds__(64) void regchol_kernel(const double* A, double* Lout, double* Xout, long long* cyc)
{
    const int lane = threadIdx.x;
    const int row = lane < NV ? lane : NV - 1;
    const double* Ab = A + (size_t)blockIdx.x * NV * NV;
    double a[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) a[j] = Ab[row * NV + j];
    const long long t0 = __builtin_readcyclecounter();
    // ---- Cholesky (Banachiewicz): after step j, a[j] = L[lane][j] for lane >= j
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        double s = a[j];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= a[k] * bcast(a[k], j); // L[i][k] * L[j][k]
        const double piv = bcast(s, j);
        const double rinv = fast_rsqrt(piv);
        a[j] = (lane == j) ? piv * rinv : s * rinv;
    }
    const long long t1 = __builtin_readcyclecounter();
    // ---- X = L^-1 (lower): row i of X in lane i.  X[i][c] = (delta_ic - sum_{k=c}^{i-1} L[i][k] X[k][c]) / L[i][i]
    // computed row by row is sequential in i; instead column-oriented forward substitution with lane = row:
    //   for c: x = e_c; for k = c..n-1: X[k][c] = x_k / L[k][k] (lane k), broadcast, x_i -= L[i][k] X[k][c] (i > k)
    // which is n^2/2 broadcasts as well.  Here: x[c] holds the running right-hand side of column c in lane = row.
    double x[NV];
    const double rdiag = 1.0 / a[row < NV ? row : 0]; // placeholder, fixed below
    (void)rdiag;
    double dinv = 1.0;
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (lane == j) dinv = 1.0 / a[j];
#pragma unroll
    for (int c = 0; c < NV; ++c) x[c] = (lane == c) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        // row k of X is final once all columns c <= k have processed step k:  X[k][c] = x_k[c] / L[k][k]
#pragma unroll
        for (int c = 0; c <= k; ++c) {
            const double xk = bcast(x[c] * dinv, k); // X[k][c]
            if (lane == k) x[c] = xk;
            else if (lane > k) x[c] -= a[k] * xk; // L[i][k] X[k][c]
        }
    }
    const long long t2 = __builtin_readcyclecounter();
    if (lane < NV) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            Lout[(size_t)blockIdx.x * NV * NV + lane * NV + j] = (j <= lane) ? a[j] : 0.0;
            Xout[(size_t)blockIdx.x * NV * NV + lane * NV + j] = (j <= lane) ? x[j] : 0.0;
        }
    }
    if (lane == 0) {
        cyc[2 * blockIdx.x] = t1 - t0;
        cyc[2 * blockIdx.x + 1] = t2 - t1;
    }
}

int main()
{
    const int batch = 4096;
    std::vector<double> A((size_t)batch * NV * NV), L(A.size()), X(A.size());
    for (int b = 0; b < batch; ++b) {
        std::vector<double> M(NV * NV);
        unsigned s = 12345u + b;
        for (auto& v : M) {
            s = s * 1664525u + 1013904223u;
            v = ((s >> 8) & 0xffff) / 65536.0 - 0.5;
        }
        for (int i = 0; i < NV; ++i)
            for (int j = 0; j < NV; ++j) {
                double acc = (i == j) ? 1.0 : 0.0;
                for (int k = 0; k < NV; ++k) acc += M[i * NV + k] * M[j * NV + k] / NV;
                A[(size_t)b * NV * NV + i * NV + j] = acc;
            }
    }
    double *dA, *dL, *dX;
    long long* dC;
    hipMalloc(&dA, A.size() * 8);
    hipMalloc(&dL, A.size() * 8);
    hipMalloc(&dX, A.size() * 8);
    hipMalloc(&dC, batch * 16);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(regchol_kernel, dim3(batch), dim3(64), 0, 0, dA, dL, dX, dC);
    hipDeviceSynchronize();
    hipMemcpy(L.data(), dL, A.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(X.data(), dX, A.size() * 8, hipMemcpyDeviceToHost);
    std::vector<long long> C(batch * 2);
    hipMemcpy(C.data(), dC, batch * 16, hipMemcpyDeviceToHost);
    double errL = 0, errX = 0;
    for (int b = 0; b < 8; ++b) {
        const double* Ab = &A[(size_t)b * NV * NV];
        const double* Lb = &L[(size_t)b * NV * NV];
        const double* Xb = &X[(size_t)b * NV * NV];
        for (int i = 0; i < NV; ++i)
            for (int j = 0; j <= i; ++j) {
                double acc = 0, acc2 = 0;
                for (int k = 0; k < NV; ++k) {
                    acc += Lb[i * NV + k] * Lb[j * NV + k];
                    acc2 += Lb[i * NV + k] * Xb[k * NV + j];
                }
                errL = fmax(errL, fabs(acc - Ab[i * NV + j]));
                errX = fmax(errX, fabs(acc2 - (i == j ? 1.0 : 0.0)));
            }
    }
    double c0 = 0, c1 = 0;
    for (int b = 0; b < batch; ++b) c0 += C[2 * b], c1 += C[2 * b + 1];
    printf("err LL'-A %.2e  L X - I %.2e   cycles: cholesky %.0f  inverse %.0f\n", errL, errX, c0 / batch, c1 / batch);
    return 0;
}
