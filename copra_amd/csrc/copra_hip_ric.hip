// copra_hip_ric.hip -- the Riccati-factor tier (lmpc_fused_ric.hpp) and the one-instance-per-lane pass (lmpc_lane.hpp) with a RUN-TIME
// horizon, for the shapes the library covers at every horizon (plan_builder.hpp::ric_aot_shape: the double integrators in one, two and
// three dimensions).  Round 3 shipped these kernels for (6, 3) at N = 10, 15, 20 only; every other horizon ran on the round-1 tiers unless
// the user's box had hipcc for copra_batch_specialise.  A translation unit of its own: it compiles next to copra_hip.hip (make -j).
#include "ric_kernels.hpp"

#define COPRA_RIC_RT_INST(NX, NU)                                                                                      \
    template __global__ void copra_lmpc_fused_ric_kernel<NX, NU, 0, kFusedQ1Regs, false>(const FusedPlan);             \
    template __global__ void copra_lmpc_fused_ric_kernel<NX, NU, 0, 0, false>(const FusedPlan);                        \
    template __global__ void copra_lmpc_fused_ric_kernel<NX, NU, 0, kFusedQ1Regs, true>(const FusedPlan);              \
    template __global__ void copra_lmpc_fused_ric_kernel<NX, NU, 0, 0, true>(const FusedPlan);
COPRA_RIC_RT_INST(6, 3)
COPRA_RIC_RT_INST(4, 2)
COPRA_RIC_RT_INST(2, 1)
template __global__ void copra_lmpc_lane_kernel<4, 2, false, true>(const FusedPlan);
template __global__ void copra_lmpc_lane_kernel<4, 2, true, true>(const FusedPlan);
template __global__ void copra_lmpc_lane_kernel<4, 2, false, false>(const FusedPlan);
template __global__ void copra_lmpc_lane_kernel<4, 2, true, false>(const FusedPlan);
