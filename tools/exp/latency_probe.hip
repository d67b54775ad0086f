// Dependent-chain latencies on gfx950 as seen by ONE wavefront (and by two per SIMD): what a latency-bound
// one-instance-per-wave kernel pays per dependent step.   hipcc --offload-arch=gfx950 -O3 latency_probe.hip -o latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 512
__device__ __forceinline__ long long clk() { return (long long)__builtin_readcyclecounter(); }
__global__ __launch_bounds__(64) void probe(double* out, long long* t, int waves_note)
{
    __shared__ double sm[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) sm[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    double x = 1.0 + lane * 1e-12, y = 0.999999;
    long long t0, t1;
    // 1. dependent v_fma_f64
    t0 = clk();
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = __builtin_fma(x, y, 1e-9);
    t1 = clk();
    if (lane == 0) t[0] = t1 - t0;
    // 2. independent v_fma_f64 (4 chains)
    double a = x, b = x + 1, c = x + 2, d = x + 3;
    t0 = clk();
#pragma unroll 4
    for (int i = 0; i < N / 4; ++i) {
        a = __builtin_fma(a, y, 1e-9);
        b = __builtin_fma(b, y, 1e-9);
        c = __builtin_fma(c, y, 1e-9);
        d = __builtin_fma(d, y, 1e-9);
    }
    t1 = clk();
    x = a + b + c + d;
    if (lane == 0) t[1] = t1 - t0;
    // 3. readlane -> fma chain (value goes lane 0 -> SGPR -> all lanes)
    t0 = clk();
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        const int lo = __builtin_amdgcn_readlane((int)__double2loint(x), 3), hi = __builtin_amdgcn_readlane(__double2hiint(x), 3);
        x = __builtin_fma(__hiloint2double(hi, lo), y, 1e-9);
    }
    t1 = clk();
    if (lane == 0) t[2] = t1 - t0;
    // 4. dependent LDS read chain (pointer chase through indices)
    int idx = lane;
    t0 = clk();
#pragma unroll 16
    for (int i = 0; i < N; ++i) idx = ((int)sm[idx & 1023]) + lane;
    t1 = clk();
    if (lane == 0) t[3] = t1 - t0;
    // 5. LDS write -> wave fence -> read of another lane's value -> fma (the sync step of the kernels)
    t0 = clk();
#pragma unroll 8
    for (int i = 0; i < N; ++i) {
        sm[lane] = x;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        x = __builtin_fma(sm[(lane + 1) & 63], y, 1e-9);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    t1 = clk();
    if (lane == 0) t[4] = t1 - t0;
    // 6. v_rcp_f64 + two Newton steps, dependent
    t0 = clk();
#pragma unroll 8
    for (int i = 0; i < N; ++i) {
        double r = __builtin_amdgcn_rcp(x);
        r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
        r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
        x = r + 0.5;
    }
    t1 = clk();
    if (lane == 0) t[5] = t1 - t0;
    // 7. dependent v_cndmask pair + add (32-bit ALU latency)
    int q = idx;
    t0 = clk();
#pragma unroll 16
    for (int i = 0; i < N; ++i) q = q * 3 + 1;
    t1 = clk();
    if (lane == 0) t[6] = t1 - t0;
    out[blockIdx.x * 64 + lane] = x + idx + q;
}
int main()
{
    double* out;
    long long* t;
    hipMalloc(&out, 4096 * 64 * 8);
    hipMalloc(&t, 64 * 8);
    const char* names[] = { "dependent v_fma_f64", "4 independent v_fma_f64 chains (per fma)", "v_readlane x2 -> v_fma_f64", "dependent ds_read_b64", "ds_write + fence + ds_read + fma + fence", "v_rcp_f64 + 2 Newton + add (5 dependent ops + rcp)", "dependent v_mul_lo/add i32" };
    for (int blocks : { 1, 256 * 8, 256 * 16 }) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, out, t, 0);
        hipDeviceSynchronize();
        std::vector<long long> h(8);
        hipMemcpy(h.data(), t, 64, hipMemcpyDeviceToHost);
        printf("---- %d workgroups of one wave (%s) ----\n", blocks, blocks == 1 ? "alone on the chip" : blocks == 2048 ? "2 waves per SIMD" : "4 waves per SIMD");
        for (int k = 0; k < 7; ++k) printf("%-55s %7.1f cycles per step\n", names[k], (double)h[k] / N);
    }
    return 0;
}
