// ric_kernels.hpp -- the __global__ templates of the Riccati-factor tier and of the one-instance-per-lane pass, shared by the translation
// units that instantiate them: copra_hip.hip (compile-time horizons of the BASELINE shapes) and copra_hip_ric.hip (run-time horizon,
// NH == 0, for the shapes of plan_builder.hpp::ric_aot_shape).
#pragma once
#include <hip/hip_runtime.h>

#include "lmpc_fused_ric.hpp"
#include "lmpc_lane.hpp"

using namespace copra_hip;

// The same tier with the factor in Riccati form (lmpc_fused_ric.hpp): controllers whose costs are all per-step entries.
// (three waves per SIMD: 168 VGPRs; compact variant: 13.9 KB of LDS per instance, eleven instances share a CU; general
//  variant: 17.7 KB + the rows' share, seven to nine -- the LDS allocation granule is 1280 B, profiles/r02/lds_granule_probe.txt)
template <int NX, int NU, int NH, int QR, bool SREFS = false>
__global__ __launch_bounds__(64, 3) void copra_lmpc_fused_ric_kernel(const FusedPlan P)
{
    if (P.ovf_zero && blockIdx.x == 0 && threadIdx.x == 0) *P.ovf_zero = 0; // (the NEXT solve's overflow counter: begin_overflow_queue)
    int inst;
    bool lane_failed;
    if (!tier_instance(P, (int)blockIdx.x, inst, lane_failed)) return;
    lmpc_fused_ric_body<NX, NU, NH, 6, QR, SREFS>(P, inst, lane_failed);
}
// One instance per LANE (lmpc_lane.hpp): the pass in front of the Riccati-factor tier -- LQ roll-out and qpgen2's first scan for every
// instance; those that violate nothing are finished here.  One wave per SIMD: the lane's matrices live in up to 512 registers.
template <int NX, int NU, bool SREFS = false, bool SPEC = true>
__global__ __launch_bounds__(64, 1) void copra_lmpc_lane_kernel(const FusedPlan P)
{
    lmpc_lane_body<NX, NU, SREFS, SPEC>(P, (int)blockIdx.x);
}
