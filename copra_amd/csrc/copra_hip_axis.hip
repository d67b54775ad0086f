// copra_hip_axis.hip -- instantiations of the one-(instance, axis)-per-lane solver (lmpc_axis.hpp) for the shapes of
// plan_builder.hpp::axis_solver_nmax: chains of two states and one control in one, two and three dimensions, horizons up to 20 and up to 31.
// A translation unit of its own: it compiles next to copra_hip.hip (make -j).
#include "axis_kernels.hpp"

#define COPRA_AXIS_INST(NXA, NU, NMAX, QMAX, EXACT, CT, RPA) template __global__ void copra_lmpc_axis_kernel<NXA, NU, NMAX, QMAX, EXACT, CT, RPA>(const FusedPlan);
COPRA_AXIS_KERNELS(COPRA_AXIS_INST)
#define COPRA_AXIS_LIST_INST(NXA, NU, NMAX, QMAX, CT, RPA) template __global__ void copra_lmpc_axis_list_kernel<NXA, NU, NMAX, QMAX, CT, RPA>(const FusedPlan);
COPRA_AXIS_LIST_KERNELS(COPRA_AXIS_LIST_INST)
