// wave_prims.hpp -- the few wavefront-level primitives the kernels are written against (gfx950, wave64).
// One MPC instance is solved by ONE 64-lane wavefront (a 64-thread workgroup), so every cross-lane exchange is
// either an LDS round trip fenced by wave_sync() or a wave shuffle.
// (tests/emu/ provides a CPU stand-in for this header so that the kernel bodies can run under sanitizers.)
#ifndef COPRA_WAVE_PRIMS_HPP
#define COPRA_WAVE_PRIMS_HPP
#include <hip/hip_runtime.h>

#define COPRA_DEV __device__ __forceinline__

namespace copra_hip {

COPRA_DEV int lane_id() { return (int)threadIdx.x; }
COPRA_DEV int instance_id() { return (int)blockIdx.x; }
COPRA_DEV int instance_stride() { return (int)gridDim.x; }

// workgroup == one wave: the barrier only has to order LDS traffic
COPRA_DEV void wave_sync() { __syncthreads(); }

COPRA_DEV double* lds_base()
{
    extern __shared__ __attribute__((aligned(16))) double copra_lds[];
    return copra_lds;
}

COPRA_DEV double shfl_xor_f64(double v, int mask) { return __shfl_xor(v, mask, 64); }
COPRA_DEV int shfl_xor_i32(int v, int mask) { return __shfl_xor(v, mask, 64); }
COPRA_DEV double shfl_f64(double v, int src) { return __shfl(v, src, 64); }
COPRA_DEV int shfl_i32(int v, int src) { return __shfl(v, src, 64); }
// value of lane (lane+delta), 0 beyond the end of the wave
COPRA_DEV double shfl_down0_f64(double v, int delta)
{
    const double t = __shfl_down(v, (unsigned)delta, 64);
    return ((int)threadIdx.x + delta < 64) ? t : 0.0;
}
// value of lane (lane-delta), 0 before the start of the wave
COPRA_DEV double shfl_up0_f64(double v, int delta)
{
    const double t = __shfl_up(v, (unsigned)delta, 64);
    return ((int)threadIdx.x - delta >= 0) ? t : 0.0;
}

COPRA_DEV double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
    return v;
}
COPRA_DEV double wave_max(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, shfl_xor_f64(v, m));
    return v;
}

} // namespace copra_hip
#endif // COPRA_WAVE_PRIMS_HPP
