// Operand / result layout of v_mfma_f64_4x4x4f64 (four independent 4x4x4 products per instruction) found by one-hot probing,
// and its issue cost next to v_mfma_f64_16x16x4f64.   hipcc --offload-arch=gfx950 -O3 mfma444_probe.hip -o mfma444_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void probe(int* tab)
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            if (d != 0.0) tab[la * 64 + lb] = lane; // D lane that sees A(la) * B(lb)
        }
}
__global__ __launch_bounds__(64) void timing(double* out, long long* t)
{
    const int lane = threadIdx.x;
    double a = 1.0 + lane * 1e-9, b = 0.999;
    double d = 0.0;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < 512; ++i) d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d, 0, 0, 0);
    long long t1 = __builtin_readcyclecounter();
    if (lane == 0) t[0] = t1 - t0;
    typedef double v4 __attribute__((ext_vector_type(4)));
    v4 c = { 0, 0, 0, 0 };
    t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < 512; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    t1 = __builtin_readcyclecounter();
    if (lane == 0) t[1] = t1 - t0;
    // B operand = previous result (the chained recursion)
    double e = a;
    t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < 512; ++i) e = __builtin_amdgcn_mfma_f64_4x4x4f64(b, e, 0.0, 0, 0, 0);
    t1 = __builtin_readcyclecounter();
    if (lane == 0) t[2] = t1 - t0;
    out[blockIdx.x * 64 + lane] = d + c[0] + c[1] + c[2] + c[3] + e;
}
int main()
{
    int* tab;
    hipMalloc(&tab, 64 * 64 * 4);
    hipMemset(tab, 0xff, 64 * 64 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, tab);
    std::vector<int> h(64 * 64);
    hipMemcpy(h.data(), tab, 64 * 64 * 4, hipMemcpyDeviceToHost);
    printf("D lane that receives A(lane la) * B(lane lb); '.' = the two never meet\n     lb:");
    for (int lb = 0; lb < 64; ++lb) printf("%3d", lb);
    printf("\n");
    for (int la = 0; la < 64; ++la) {
        printf("la %2d:  ", la);
        for (int lb = 0; lb < 64; ++lb) {
            if (h[la * 64 + lb] < 0) printf("  .");
            else printf("%3d", h[la * 64 + lb]);
        }
        printf("\n");
    }
    double* out;
    long long* t;
    hipMalloc(&out, 4096 * 64 * 8);
    hipMalloc(&t, 64);
    for (int blocks : { 1, 2048 }) {
        hipLaunchKernelGGL(timing, dim3(blocks), dim3(64), 0, 0, out, t);
        hipLaunchKernelGGL(timing, dim3(blocks), dim3(64), 0, 0, out, t);
        hipDeviceSynchronize();
        long long ht[3];
        hipMemcpy(ht, t, 24, hipMemcpyDeviceToHost);
        printf("%d waves: dependent (srcC) mfma_f64_4x4x4 %.1f cycles, mfma_f64_16x16x4 %.1f cycles, 4x4x4 chained through B %.1f cycles\n", blocks, ht[0] / 512.0, ht[1] / 512.0, ht[2] / 512.0);
    }
    return 0;
}
