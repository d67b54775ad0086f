// mfma_chain_probe.hip -- what ONE wave pays for dependent chains built from v_mfma_f64_4x4x4 on gfx950: accumulator chains,
// result -> B operand chains (the hand-over the Riccati kernels rely on), MFMA -> DPP row broadcast -> MFMA, and the
// LDS write -> read hand-over in between.  Build:  hipcc -O3 --offload-arch=gfx950 -o mfma_chain_probe mfma_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_STEPS 1024
__device__ __forceinline__ double mf(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double bc0(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(64) void probe(double* out, long long* cyc, int mode)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    double a = 1.0 + 1e-9 * lane, b = 1.0 - 1e-9 * lane, c = 0.0, d = 0.5, e = 0.25;
    lds[lane] = a;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (mode == 0) { // accumulator chain
        for (int i = 0; i < N_STEPS; ++i) c = mf(a, b, c);
    } else if (mode == 1) { // result -> B operand
        for (int i = 0; i < N_STEPS; ++i) c = mf(a, c, d);
    } else if (mode == 2) { // result -> A operand
        for (int i = 0; i < N_STEPS; ++i) c = mf(c, b, d);
    } else if (mode == 3) { // three independent accumulator chains (per MFMA)
        for (int i = 0; i < N_STEPS; ++i) c = mf(a, b, c), d = mf(a, b, d), e = mf(a, b, e);
    } else if (mode == 4) { // MFMA -> DPP row broadcast -> B operand of the next
        for (int i = 0; i < N_STEPS; ++i) c = mf(a, bc0(c), d);
    } else if (mode == 5) { // MFMA -> LDS write -> read -> A operand of the next
        for (int i = 0; i < N_STEPS; ++i) {
            lds[lane] = c;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            c = mf(lds[lane ^ 5], b, d);
        }
    } else if (mode == 6) { // chain of 3 accumulating MFMAs whose B operand is the previous group's result (a matrix-vector stage)
        for (int i = 0; i < N_STEPS; ++i) {
            double r = d;
            r = mf(a, c, r), r = mf(b, c, r), r = mf(a, c, r);
            c = r;
        }
    } else if (mode == 8) { // one stage of the backward vector sweep of lmpc_riccati_mfma.hpp (10 MFMAs, 4 row broadcasts), registers only
        double p0 = c, p1 = d, p2 = e;
        for (int i = 0; i < N_STEPS; ++i) {
            double hv = a, hbv = b;
            hv = mf(a, p0, hv), hbv = mf(b, p0, hbv);
            hv = mf(a, p1, hv), hbv = mf(b, p1, hbv);
            hv = mf(a, p2, hv), hbv = mf(b, p2, hbv);
            const double kvb = mf(a, hbv, 0.0);
            const double hp = mf(b, hbv, hv);
            const double ha = bc0(hp);
            const double kva = mf(a, ha, 0.0);
            const double pn = mf(b, ha, hp);
            d += kva + kvb;
            p0 = bc0(pn), p1 = bc0(pn + 1.0), p2 = bc0(pn - 1.0);
        }
        c = p0 + p1 + p2;
    } else if (mode == 9) { // one stage of the forward sweep (12 MFMAs, 3 row broadcasts)
        double x0 = c, x1 = d, x2 = e;
        for (int i = 0; i < N_STEPS; ++i) {
            double ua = a, ub = b, xn = 0.0;
            ua = mf(a, x0, ua), ub = mf(b, x0, ub), xn = mf(a, x0, xn);
            ua = mf(a, x1, ua), ub = mf(b, x1, ub), xn = mf(a, x1, xn);
            ua = mf(a, x2, ua), ub = mf(b, x2, ub), xn = mf(a, x2, xn);
            ub = mf(a, ua, ub);
            xn = mf(b, ua, xn);
            xn = mf(a, ub, xn);
            x0 = bc0(xn), x1 = bc0(xn + 1.0), x2 = bc0(xn - 1.0);
        }
        c = x0 + x1 + x2;
    } else if (mode == 7) { // v_fma_f64 dependent (reference)
        for (int i = 0; i < N_STEPS; ++i) c = fma(a, c, d);
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = c + d + e;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    double* out;
    long long* cyc;
    const int maxg = 4096;
    hipMalloc(&out, maxg * 64 * sizeof(double));
    hipMalloc(&cyc, maxg * sizeof(long long));
    const char* names[] = { "accumulator chain", "result -> B operand", "result -> A operand", "3 independent accumulator chains (per MFMA)",
        "MFMA -> row_bcast (2 DPP) -> B operand", "MFMA -> LDS write -> read -> A operand", "group of 3 accumulating MFMAs on the previous result (per group)",
        "dependent v_fma_f64", "backward vector sweep stage: 10 MFMAs + 4 row broadcasts (registers only)",
        "forward sweep stage: 12 MFMAs + 3 row broadcasts (registers only)" };
    for (int grid : { 1, 512 }) {
        printf("---- %d workgroups of one wave ----\n", grid);
        for (int mode = 0; mode < 10; ++mode) {
            std::vector<long long> h(grid);
            probe<<<grid, 64, 1024>>>(out, cyc, mode);
            probe<<<grid, 64, 1024>>>(out, cyc, mode);
            hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
            double s = 0;
            for (long long v : h) s += v;
            const double per = s / grid / N_STEPS / (mode == 3 ? 3 : 1);
            printf("%-70s %7.1f cycles per step\n", names[mode], per);
        }
    }
    return 0;
}
