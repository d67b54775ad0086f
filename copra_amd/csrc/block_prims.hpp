// block_prims.hpp -- workgroup-level primitives of the LARGE-n kernels (n > 64: one MPC instance / QP per WORKGROUP of
// several 64-lane waves, J and R in HBM/L2 instead of LDS).  tests/emu/wave_prims.hpp provides the CPU stand-ins.
#ifndef COPRA_BLOCK_PRIMS_HPP
#define COPRA_BLOCK_PRIMS_HPP
#include "wave_prims.hpp"

namespace copra_hip {

COPRA_DEV int bt_tid() { return (int)threadIdx.x; }
COPRA_DEV int bt_size() { return (int)blockDim.x; }
COPRA_DEV int bt_lane() { return (int)(threadIdx.x & 63u); }
COPRA_DEV int bt_wave() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
COPRA_DEV int bt_nwaves() { return (int)(blockDim.x >> 6); }
COPRA_DEV void bt_sync() { __syncthreads(); } // orders LDS and global traffic of the workgroup

} // namespace copra_hip
#endif // COPRA_BLOCK_PRIMS_HPP
