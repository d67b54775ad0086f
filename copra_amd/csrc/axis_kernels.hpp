// axis_kernels.hpp -- the __global__ template of the one-(instance, axis)-per-lane solver (lmpc_axis.hpp), shared by the translation unit that
// instantiates it (copra_hip_axis.hip) and the one that launches it (copra_hip.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "lmpc_axis.hpp"

using namespace copra_hip;

// One wave = 64 / NU instances x NU axes.  K, 1 / M_uu, U and the recursion's t of every stage live in registers (5 NMAX doubles), next to the
// lane's active set (S: QMAX (QMAX + 1) / 2 doubles).
#ifndef COPRA_AXIS_WAVES
#define COPRA_AXIS_WAVES 1 // waves per SIMD the register budget is cut for
#endif
template <int NXA, int NU, int NMAX, int QMAX, bool EXACT, bool CT, int RPA>
__global__ __launch_bounds__(64, COPRA_AXIS_WAVES) void copra_lmpc_axis_kernel(const FusedPlan P)
{
    lmpc_axis_body<NXA, NU, NMAX, QMAX, EXACT, CT, RPA>(P, (int)blockIdx.x);
}
// (NXA, NU, NMAX, QMAX, EXACT, CT, RPA): the headline's horizon exactly; every horizon up to 20 and -- two axes -- up to 31.  With the tables in
// registers (FusedPlan::axis_const) for one and for two rows per axis and step; with the tables read from LDS stage by stage
#define COPRA_AXIS_KERNELS(X)                                                                                                    \
    X(2, 3, 20, 6, true, true, 1) X(2, 3, 20, 6, false, true, 1) X(2, 2, 20, 6, false, true, 1) X(2, 2, 31, 6, false, true, 1)    \
    X(2, 3, 20, 6, true, true, 2) X(2, 3, 20, 6, false, true, 2) X(2, 2, 20, 6, false, true, 2) X(2, 2, 31, 6, false, true, 2)    \
    X(2, 3, 20, 6, false, false, 2) X(2, 2, 20, 6, false, false, 2) X(2, 2, 31, 6, false, false, 2)
// ... and chains of THREE states per control (the jerk-controlled CoM model: position, velocity, acceleration per axis) in two and three
// dimensions: tables in registers with one row per axis and step, or read from LDS  (copra_hip_axis3.hip).  (Single-control systems stay on the
// packed kernels -- 2 or 4 instances per wave, copra_hip_packed16/32.hip --: BASELINE configs[1]'s workload ends with every bound active.)
#define COPRA_AXIS_KERNELS_MORE(X) X(1, 2, 31, 6, false, true, 1) X(1, 2, 31, 6, false, false, 2) X(1, 2, 20, 6, false, true, 1) X(1, 2, 20, 6, false, false, 2) X(1, 3, 20, 6, false, true, 1) X(1, 3, 20, 6, false, false, 2) X(2, 3, 21, 6, false, true, 1) X(2, 3, 21, 6, false, false, 2) X(3, 2, 20, 6, false, true, 1) X(3, 2, 20, 6, false, false, 2) X(3, 3, 20, 6, false, true, 1) X(3, 3, 20, 6, false, false, 2)

// The second chance of what that launch lists: the same solver with room for kAxisQmaxBig active constraints per lane, instances taken from the
// list (a grid-stride loop over it: the launch does not know its length).  One wave per CU at most (its lanes' LDS): a handful of waves.
template <int NXA, int NU, int NMAX, int QMAX, bool CT, int RPA>
__global__ __launch_bounds__(64, 1) void copra_lmpc_axis_list_kernel(const FusedPlan P)
{
    constexpr int IPW = 64 / NU;
    const int count = *P.axis_list_count;
    for (int g = (int)blockIdx.x; g * IPW < count; g += (int)gridDim.x) {
        lmpc_axis_body<NXA, NU, NMAX, QMAX, false, CT, RPA, true>(P, g);
        __syncthreads();
    }
}
#define COPRA_AXIS_LIST_KERNELS_MORE(X) X(1, 2, 31, 16, false, 2) X(1, 2, 20, 16, false, 2) X(1, 3, 20, 16, false, 2) X(2, 3, 21, 16, false, 2) X(3, 2, 20, 16, false, 2) X(3, 3, 20, 16, false, 2)
#define COPRA_AXIS_LIST_KERNELS(X) X(2, 3, 20, 16, true, 2) X(2, 2, 20, 16, true, 2) X(2, 2, 31, 16, true, 2) X(2, 3, 20, 16, false, 2) X(2, 2, 20, 16, false, 2) X(2, 2, 31, 16, false, 2)
