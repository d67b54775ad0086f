// packed_launch.hpp -- interface between copra_hip.hip and the packed translation units (packed_impl.inc): several small
// problems per wavefront, 4 x <= 16 variables or 2 x <= 32 variables
#pragma once
#include <hip/hip_runtime.h>

#include "plan.hpp"

namespace copra_hip {

// first-tier launch of the fused / shared-model / InitialStateLMPC body; `lds_bytes_per_instance` = bytes of P.lds
hipError_t packed_launch_w16(const FusedPlan& P, bool shared, size_t lds_bytes_per_instance, hipStream_t s);
hipError_t packed_launch_w32(const FusedPlan& P, bool shared, size_t lds_bytes_per_instance, hipStream_t s);
// batched dense QPs (qp_dense.hpp body)
hipError_t packed_dense_launch_w16(const DensePlan& P, size_t lds_bytes_per_instance, hipStream_t s);
hipError_t packed_dense_launch_w32(const DensePlan& P, size_t lds_bytes_per_instance, hipStream_t s);

// lanes per instance the packed build would use for `nvar` decision variables (0: the ordinary one-wave kernels).
// Packing only pays where wave slots, not LDS, limit how many instances a CU holds: the one-wave kernels run at most 8
// waves per CU (two per SIMD at their register budget), so a CU holds min(8, 160 KiB / lds) instances unpacked and
// min(8 x 64 / w, 160 KiB / lds) packed; the packed build is used when that at least doubles (measured: config 2,
// lds 3 KB, w = 16: 35 -> 150 M solves/s; n = 30, lds 14 KB, w = 32: 11 vs 8 instances per CU and rows that diverge
// -> 18.1 -> 15.4 M solves/s, hence the threshold).  Full-size COST entries need the 64-lane MFMA operand layout and
// stay on the one-wave kernels.
// `build_width` = elements one step of the preview recursion produces, xDim (xDim + uDim + 1) (0 for a dense QP): the
// condense phase is lane-parallel over those, so a group should not be much narrower (CoM system, xDim 6, uDim 3, 15
// variables: 73 M solves/s on 64 lanes, 60 M on 16 -- the recursion then takes four rounds per step).
inline int packed_width(int nvar, int build_width, bool has_full_size_costs, size_t lds_bytes_per_instance)
{
    if (has_full_size_costs || lds_bytes_per_instance == 0) return 0;
    int w = 0;
    if (nvar <= 16 && build_width <= 32)
        w = 16;
    else if (nvar <= 32 && build_width <= 64)
        w = 32;
    if (w == 0) return 0;
    const long long by_lds = (long long)(160u * 1024u / lds_bytes_per_instance);
    const long long unpacked = by_lds < 8 ? by_lds : 8;
    const long long slots = 8LL * (64 / w);
    const long long packed = by_lds < slots ? by_lds : slots;
    if ((size_t)(64 / w) * lds_bytes_per_instance > 160u * 1024u || packed < 2 * unpacked) return 0;
    return w;
}

} // namespace copra_hip
