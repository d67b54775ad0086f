// copra_hip_axis3.hip -- more instantiations of the one-(instance, axis)-per-lane solver (lmpc_axis.hpp): one chain of two states, and chains
// of three states per control (the jerk-controlled CoM model) in one, two and three dimensions -- plan_builder.hpp::axis_solver_nmax.
// A translation unit of its own: it compiles next to copra_hip_axis.hip (make -j).
#include "axis_kernels.hpp"

#define COPRA_AXIS_INST(NXA, NU, NMAX, QMAX, EXACT, CT, RPA) template __global__ void copra_lmpc_axis_kernel<NXA, NU, NMAX, QMAX, EXACT, CT, RPA>(const FusedPlan);
COPRA_AXIS_KERNELS_MORE(COPRA_AXIS_INST)
#define COPRA_AXIS_LIST_INST(NXA, NU, NMAX, QMAX, CT, RPA) template __global__ void copra_lmpc_axis_list_kernel<NXA, NU, NMAX, QMAX, CT, RPA>(const FusedPlan);
COPRA_AXIS_LIST_KERNELS_MORE(COPRA_AXIS_LIST_INST)
