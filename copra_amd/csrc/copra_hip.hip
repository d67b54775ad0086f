// copra_hip.hip -- kernels + C ABI (include/copra_hip.h) of the MI355X-native batched linear-MPC engine.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared copra_hip.hip -o libcopra_hip.so  (see Makefile)
#include "engine.hpp"
#include "islmpc_fused.hpp"
#include "lmpc_fused.hpp"
#include "lmpc_fused_ric.hpp"
#include "lmpc_lane.hpp"
#include "lmpc_large.hpp"
#include "lmpc_riccati.hpp"
#include "lmpc_riccati_mfma.hpp"
#include "lmpc_shared.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// ------------------------------------------------------------------------------------------------
// kernels: one 64-lane wavefront (= one workgroup) per MPC instance.  The hardware workgroup dispatcher is the
// work queue: instances with long active-set loops simply hold their CU slot longer.
// ------------------------------------------------------------------------------------------------
// <NX, NU, N, RP> = compile-time (xDim, uDim, nrUStep, padded cost rows); <0,0,0,0> is the generic (run-time
// shape) instantiation.
template <int NX, int NU, int NH, int RP>
__global__ __launch_bounds__(64) void copra_lmpc_fused_kernel(const FusedPlan P)
{
    int inst;
    bool lane_failed; // (a first tier that does not take the pass's factor over finds a failed factorisation itself)
    if (!tier_instance(P, (int)blockIdx.x, inst, lane_failed)) return;
    lmpc_fused_body<NX, NU, NH, RP>(P, inst);
}

// Run-time shapes whose (compact) LDS layout lets more than 8 instances share a CU: the same bodies at four waves per
// SIMD (128 VGPRs) -- at the 256-VGPR budget of the kernels above a CU holds 8 waves, whatever the LDS would allow.
__global__ __launch_bounds__(64, 4) void copra_lmpc_fused_kernel_w4(const FusedPlan P)
{
    int inst;
    bool lane_failed;
    if (!tier_instance(P, (int)blockIdx.x, inst, lane_failed)) return;
    lmpc_fused_body<0, 0, 0, 0>(P, inst);
}
__global__ __launch_bounds__(64, 4) void copra_lmpc_shared_kernel_w4(const FusedPlan P)
{
    lmpc_shared_body<0, 0, 0>(P, (int)blockIdx.x);
}

// Factor-only first tier (LdsLayout::tri): the packed Cholesky factor is the only O(n^2) object in LDS, so six instances
// share a CU at the headline shape (two waves on two of the four SIMDs: 256 VGPRs each) instead of four.
template <int NX, int NU, int NH, int RP, int QR = 0>
__global__ __launch_bounds__(64, 2) void copra_lmpc_fused_tri_kernel(const FusedPlan P)
{
    int inst;
    bool lane_failed;
    if (!tier_instance(P, (int)blockIdx.x, inst, lane_failed)) return;
    lmpc_fused_body<NX, NU, NH, RP, true, QR>(P, inst);
}

#include "ric_kernels.hpp" // copra_lmpc_fused_ric_kernel, copra_lmpc_lane_kernel
#include "axis_kernels.hpp" // copra_lmpc_axis_kernel: instantiated in copra_hip_axis.hip
#define COPRA_AXIS_DECL(NXA, NU, NMAX, QMAX, EXACT, CT, RPA) extern template __global__ void copra_lmpc_axis_kernel<NXA, NU, NMAX, QMAX, EXACT, CT, RPA>(const FusedPlan);
COPRA_AXIS_KERNELS(COPRA_AXIS_DECL)
COPRA_AXIS_KERNELS_MORE(COPRA_AXIS_DECL)
#define COPRA_AXIS_LIST_DECL(NXA, NU, NMAX, QMAX, CT, RPA) extern template __global__ void copra_lmpc_axis_list_kernel<NXA, NU, NMAX, QMAX, CT, RPA>(const FusedPlan);
COPRA_AXIS_LIST_KERNELS(COPRA_AXIS_LIST_DECL)
COPRA_AXIS_LIST_KERNELS_MORE(COPRA_AXIS_LIST_DECL)
// run-time-horizon builds (NH == 0) for the shapes of ric_aot_shape: instantiated in copra_hip_ric.hip, a translation unit of its own
#define COPRA_RIC_RT_DECL(NX, NU)                                                                                      \
    extern template __global__ void copra_lmpc_fused_ric_kernel<NX, NU, 0, kFusedQ1Regs, false>(const FusedPlan);      \
    extern template __global__ void copra_lmpc_fused_ric_kernel<NX, NU, 0, 0, false>(const FusedPlan);                 \
    extern template __global__ void copra_lmpc_fused_ric_kernel<NX, NU, 0, kFusedQ1Regs, true>(const FusedPlan);       \
    extern template __global__ void copra_lmpc_fused_ric_kernel<NX, NU, 0, 0, true>(const FusedPlan);
COPRA_RIC_RT_DECL(6, 3)
COPRA_RIC_RT_DECL(4, 2)
COPRA_RIC_RT_DECL(2, 1)
extern template __global__ void copra_lmpc_lane_kernel<4, 2, false, true>(const FusedPlan);
extern template __global__ void copra_lmpc_lane_kernel<4, 2, true, true>(const FusedPlan);
extern template __global__ void copra_lmpc_lane_kernel<4, 2, false, false>(const FusedPlan);
extern template __global__ void copra_lmpc_lane_kernel<4, 2, true, false>(const FusedPlan);
// ... and its shared-model form: the stage records are those of the whole batch (wave-uniform: scalar operands), only the roll-out
// from each instance's x0 is left
template <int NX, int NU, bool SPEC = true>
__global__ __launch_bounds__(64, 4) void copra_lmpc_lane_shared_kernel(const FusedPlan P)
{
    lmpc_lane_shared_body<NX, NU, SPEC>(P, (int)blockIdx.x);
}
// Second tier of the two-tier scheme (own symbol so that profiles keep the two apart): the same body with the full LDS
// layout, run only for the instances whose active set outgrew the compact layout's R (queue filled by the first tier).
template <int NX, int NU, int NH, int RP>
__global__ __launch_bounds__(64) void copra_lmpc_fused_tier2_kernel(const FusedPlan P)
{
    if (P.seen_out && blockIdx.x == 0 && threadIdx.x == 0) { // (FusedPlan::seen_out: the lists are complete before this launch starts)
        P.seen_out[0] = *P.seen_src0;
        P.seen_out[2] = *P.seen_src1;
    }
    const int count = *P.ovf_count;
    for (int k = (int)blockIdx.x; k < count; k += (int)gridDim.x) {
        lmpc_fused_body<NX, NU, NH, RP>(P, P.ovf_list[k]);
        __syncthreads();
    }
    // ... and what the first tier's grid did not reach of the list in front of it (FusedPlan::lane_cap: that grid follows the lengths of the
    // last solves' lists; a list that grew past it -- the workload changed -- is finished here, from scratch)
    if (P.lane_rest >= 0) {
        const int total = *P.lane_count;
        for (int k = P.lane_rest + (int)blockIdx.x; k < total; k += (int)gridDim.x) {
            lmpc_fused_body<NX, NU, NH, RP>(P, P.lane_list[k] & 0x7fffffff);
            __syncthreads();
        }
    }
}

// Shared-model fast path (lmpc_shared.hpp): the factorisation comes from a one-workgroup prepare launch of the fused
// kernel; same two-tier scheme (own symbols so that profiles keep the kernels apart).
template <int NX, int NU, int NH>
__global__ __launch_bounds__(64) void copra_lmpc_shared_kernel(const FusedPlan P)
{
    lmpc_shared_body<NX, NU, NH>(P, (int)blockIdx.x);
}
template <int NX, int NU, int NH>
__global__ __launch_bounds__(64, 2) void copra_lmpc_shared_tri_kernel(const FusedPlan P) // factor-only first tier
{
    lmpc_shared_body<NX, NU, NH, true>(P, (int)blockIdx.x);
}
template <int NX, int NU, int NH>
__global__ __launch_bounds__(64) void copra_lmpc_shared_tier2_kernel(const FusedPlan P)
{
    const int count = *P.ovf_count;
    for (int k = (int)blockIdx.x; k < count; k += (int)gridDim.x) {
        lmpc_shared_body<NX, NU, NH>(P, P.ovf_list[k]);
        __syncthreads();
    }
}




static fused_kernel_t select_shared_kernel(const FusedPlan& P, bool tier2)
{
    if (P.lds.tri && !tier2) {
        if (P.nx == 6 && P.nu == 3 && P.N == 20) return copra_lmpc_shared_tri_kernel<6, 3, 20>;
        return copra_lmpc_shared_tri_kernel<0, 0, 0>;
    }
    if (P.nx == 6 && P.nu == 3 && P.N == 20)
        return tier2 ? copra_lmpc_shared_tier2_kernel<6, 3, 20> : copra_lmpc_shared_kernel<6, 3, 20>;
    if (P.nx == 2 && P.nu == 1 && P.N == 10)
        return tier2 ? copra_lmpc_shared_tier2_kernel<2, 1, 10> : copra_lmpc_shared_kernel<2, 1, 10>;
    if (!tier2 && (size_t)P.lds.total * sizeof(double) * 9 <= 160u * 1024u) return copra_lmpc_shared_kernel_w4;
    return tier2 ? copra_lmpc_shared_tier2_kernel<0, 0, 0> : copra_lmpc_shared_kernel<0, 0, 0>;
}
// the BASELINE.json shapes get their own instantiation
fused_kernel_t select_fused_kernel(const FusedPlan& P)
{
    const int rp = specialised_cost_rows(P.nx, P.nu, P.N, P.rmax, P.rfull);
    if (P.lds.tri) {
        if (P.lds.ric && !ric_aot_exact(P.nx, P.nu, P.N)) { // run-time-horizon builds of the shape (copra_hip_ric.hip)
#define COPRA_RIC_RT_PICK(NX, NU)                                                                                      \
    if (P.nx == NX && P.nu == NU)                                                                                      \
        return P.stage_refs ? (P.lds.q1regs ? copra_lmpc_fused_ric_kernel<NX, NU, 0, kFusedQ1Regs, true> : copra_lmpc_fused_ric_kernel<NX, NU, 0, 0, true>) \
                            : (P.lds.q1regs ? copra_lmpc_fused_ric_kernel<NX, NU, 0, kFusedQ1Regs, false> : copra_lmpc_fused_ric_kernel<NX, NU, 0, 0, false>);
            COPRA_RIC_RT_PICK(6, 3)
            COPRA_RIC_RT_PICK(4, 2)
            COPRA_RIC_RT_PICK(2, 1)
#undef COPRA_RIC_RT_PICK
        }
        if (P.lds.ric) { // (plan_builder.hpp: only these shapes get the layout; Q1 in registers, or in LDS further down the ladder)
            if (P.stage_refs) { // (reference trajectories: the builds with the stage-varying affine term)
                if (P.N == 10) return P.lds.q1regs ? copra_lmpc_fused_ric_kernel<6, 3, 10, kFusedQ1Regs, true> : copra_lmpc_fused_ric_kernel<6, 3, 10, 0, true>;
                if (P.N == 15) return P.lds.q1regs ? copra_lmpc_fused_ric_kernel<6, 3, 15, kFusedQ1Regs, true> : copra_lmpc_fused_ric_kernel<6, 3, 15, 0, true>;
                return P.lds.q1regs ? copra_lmpc_fused_ric_kernel<6, 3, 20, kFusedQ1Regs, true> : copra_lmpc_fused_ric_kernel<6, 3, 20, 0, true>;
            }
            if (P.N == 10) return P.lds.q1regs ? copra_lmpc_fused_ric_kernel<6, 3, 10, kFusedQ1Regs> : copra_lmpc_fused_ric_kernel<6, 3, 10, 0>;
            if (P.N == 15) return P.lds.q1regs ? copra_lmpc_fused_ric_kernel<6, 3, 15, kFusedQ1Regs> : copra_lmpc_fused_ric_kernel<6, 3, 15, 0>;
            return P.lds.q1regs ? copra_lmpc_fused_ric_kernel<6, 3, 20, kFusedQ1Regs> : copra_lmpc_fused_ric_kernel<6, 3, 20, 0>;
        }
        if (P.nx == 6 && rp == 6 && P.lds.q1regs == kFusedQ1Regs) return copra_lmpc_fused_tri_kernel<6, 3, 20, 6, kFusedQ1Regs>;
        if (P.nx == 6 && rp == 6) return copra_lmpc_fused_tri_kernel<6, 3, 20, 6>;
        if (P.rfull > 0 && P.nx == 6 && P.nu == 3 && P.N == 20) // headline shape, full-size costs
            return P.lds.q1regs == kFusedQ1Regs ? copra_lmpc_fused_tri_kernel<6, 3, 20, 0, kFusedQ1Regs> : copra_lmpc_fused_tri_kernel<6, 3, 20, 0>;
        return copra_lmpc_fused_tri_kernel<0, 0, 0, 0>;
    }
    if (P.nx == 6 && rp == 6) return copra_lmpc_fused_kernel<6, 3, 20, 6>;
    if (P.nx == 2 && rp == 2) return copra_lmpc_fused_kernel<2, 1, 10, 2>;
    if (P.rfull > 0 && P.nx == 6 && P.nu == 3 && P.N == 20) return copra_lmpc_fused_kernel<6, 3, 20, 0>; // headline shape, full-size costs
    if ((size_t)P.lds.total * sizeof(double) * 9 <= 160u * 1024u) return copra_lmpc_fused_kernel_w4; // > 8 per CU
    return copra_lmpc_fused_kernel<0, 0, 0, 0>;
}
fused_kernel_t select_tier2_kernel(const FusedPlan& P)
{
    const int rp = specialised_cost_rows(P.nx, P.nu, P.N, P.rmax, P.rfull);
    if (P.nx == 6 && rp == 6) return copra_lmpc_fused_tier2_kernel<6, 3, 20, 6>;
    if (P.nx == 2 && rp == 2) return copra_lmpc_fused_tier2_kernel<2, 1, 10, 2>;
    if (P.rfull > 0 && P.nx == 6 && P.nu == 3 && P.N == 20) return copra_lmpc_fused_tier2_kernel<6, 3, 20, 0>;
    return copra_lmpc_fused_tier2_kernel<0, 0, 0, 0>;
}


__global__ __launch_bounds__(64) void copra_islmpc_fused_kernel(const FusedPlan P)
{
    islmpc_fused_body(P, P.inst_offset + (int)blockIdx.x);
}


// more than 64 decision variables: one MPC instance per workgroup (lmpc_large.hpp), persistent grid over the batch
__global__ __launch_bounds__(kLargeMaxN) void copra_lmpc_large_kernel(const FusedPlan P) { lmpc_large_body(P); }
// Same body with the register budget of FOUR waves per SIMD (128 VGPRs, some spilling): on gfx950 a second five- to
// eight-wave workgroup is only placed on a CU when every SIMD still has room for its share of the waves, which in
// practice needs four wave slots per SIMD (tools/exp/coresidency.hip: 168 VGPRs = three slots never co-schedules
// two five-wave workgroups although the occupancy API reports two).  Each workgroup is a chain of HBM round trips, so
// two resident workgroups per CU win despite the spills: config 5 2.18 k -> 2.76 k solves/s, 300-step fixture
// 35.0 k -> 41.2 k solves/s.  Used whenever the LDS footprint lets two workgroups share a CU.
__global__ __launch_bounds__(kLargeMaxN, 4) void copra_lmpc_large_kernel_w4(const FusedPlan P) { lmpc_large_body(P); }

// Long horizons whose pieces are all stage-wise (stage_plan.hpp): Riccati interior-point method, one instance per
// wavefront, persistent grid (lmpc_riccati.hpp); the instances it does not converge on are queued for the kernel above.
template <int NXT, int NUT>
__global__ __launch_bounds__(64, 2) void copra_lmpc_riccati_kernel(const FusedPlan P, const StagePlan S) // (256 VGPRs: the prefetched operands of the next stage and the register-resident factorisation spill below that; LDS admits 10 waves per CU, this gives 8)
{
    lmpc_riccati_body<NXT, NUT>(P, S);
}
// The same method with the iterate resident on the CU and the stage algebra on the matrix cores (lmpc_riccati_mfma.hpp): what
// BASELINE config 5 runs on.  512 VGPRs (the per-row state of the instance lives in registers), <= 80 KB of LDS: two per CU.
__global__ __launch_bounds__(64, 1) void copra_lmpc_riccati_mfma_kernel(const FusedPlan P, const StagePlan S)
{
    lmpc_riccati_mfma_body(P, S);
}
typedef void (*riccati_kernel_t)(const FusedPlan, const StagePlan);
// shapes with their own instantiation: BASELINE config 5 (12, 6) and the reference's test fixtures (2, 1)
static riccati_kernel_t select_riccati_kernel(int nx, int nu)
{
    if (nx == 12 && nu == 6) return copra_lmpc_riccati_kernel<12, 6>;
    if (nx == 2 && nu == 1) return copra_lmpc_riccati_kernel<2, 1>;
    return copra_lmpc_riccati_kernel<0, 0>;
}


// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
thread_local std::string g_copra_err; // copra_last_error()

// resident workgroups per CU the runtime reports for a kernel variant (at least 1, at most 8)
int large_per_cu(const void* kernel, int threads, size_t lds_bytes)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds_bytes) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    return per_cu > 8 ? 8 : per_cu;
}

// persistent grid of the workgroup-per-instance kernels: as many workgroups as the device keeps resident
int large_grid(const copra_options_t& opt, const void* kernel, int batch, int threads, size_t lds_bytes)
{
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    int per_cu = large_per_cu(kernel, threads, lds_bytes);
    if (opt.debug)
        fprintf(stderr, "[copra] large grid: %d CUs x %d workgroups of %d threads, %zu B LDS\n", cus, per_cu, threads, lds_bytes);
    long long g = (long long)cus * per_cu;
    return (int)(g < batch ? g : batch);
}

// The 128-VGPR build of a kernel is worth its spills only where it puts MORE workgroups on a CU than the full-budget
// build.  (The occupancy API is trusted for one to four waves per SIMD; for five- to eight-wave workgroups at THREE
// waves per SIMD it over-reports, tools/exp/coresidency.hip -- that budget is not used.)
bool prefer_w4(const copra_options_t& opt, const void* full, const void* w4, int threads, size_t lds_bytes)
{
    if (lds_bytes > 48 * 1024) { // the occupancy query honours the opt-in limit of each symbol
        (void)hipFuncSetAttribute(full, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute(w4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    }
    return large_per_cu(w4, threads, lds_bytes) > large_per_cu(full, threads, lds_bytes);
}

typedef void (*large_kernel_t)(const FusedPlan);
static large_kernel_t choose_large_kernel(const HostPlan& hp)
{
    if (prefer_w4(hp.opt, reinterpret_cast<const void*>(copra_lmpc_large_kernel), reinterpret_cast<const void*>(copra_lmpc_large_kernel_w4),
            hp.plan.large.threads, hp.lds_bytes))
        return copra_lmpc_large_kernel_w4;
    return copra_lmpc_large_kernel;
}

// Dynamic LDS beyond 48 KiB is an opt-in PER KERNEL SYMBOL: raise the limit of exactly the function that is about to be
// launched, to the size it is launched with (several symbols -- first tier, second tier, parity dump, shared-model
// prepare, run-time-compiled kernels -- run with different layouts of the same controller).  Remembered per function.
hipError_t lds_opt_in(const void* fn, size_t bytes)
{
    if (bytes <= 48u * 1024u) return hipSuccess;
    static std::mutex mu;
    static std::map<const void*, size_t> granted;
    std::lock_guard<std::mutex> lock(mu);
    auto it = granted.find(fn);
    if (it != granted.end() && it->second >= bytes) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) granted[fn] = bytes;
    return e;
}



FusedPlan device_plan(const copra_batch* h)
{
    FusedPlan P = h->hp.plan;
    P.row_step = h->d_row_step;
    P.row_ekind = h->d_row_ekind;
    P.row_eoff = h->d_row_eoff;
    P.row_gkind = h->d_row_gkind;
    P.row_goff = h->d_row_goff;
    P.row_f = h->d_row_f;
    P.row_prev = h->d_row_prev;
    P.warm_set = h->shared ? h->d_warm : nullptr;
    P.params = h->d_params;
    P.lb = h->d_lb;
    P.ub = h->d_ub;
    P.A = h->A;
    P.B = h->B;
    P.d = h->d;
    P.x0 = h->x0;
    P.control = h->ext_control ? h->ext_control : h->d_control;
    P.trajectory = h->ext_traj ? h->ext_traj : h->d_traj;
    P.status = h->ext_status ? h->ext_status : h->d_status;
    P.iter = h->ext_iter ? h->ext_iter : h->d_iter;
    P.inst_offset = 0;
    P.dump_instance = -1;
    P.dump_only = 0;
    P.dumpQ = P.dumpc = P.dumpA = P.dumpb = nullptr;
    P.prof = h->d_prof;
    P.prof_fine = h->d_prof_fine;
    P.is_R = h->d_isR;
    P.is_r = h->d_isr;
    P.x0lb = h->x0lb;
    P.x0ub = h->x0ub;
    P.x0_opt = h->d_x0opt;
    P.ovf_count = h->d_ovf_count + h->ovf_cur;
    P.ovf_zero = nullptr;
    P.ovf_list = h->d_ovf_list;
    P.from_list = 0;
    P.lane_from_list = 0;
    if (h->axis_order == 1) { // (the systems' states are in axis-major order: that order's tables -- plan_builder.hpp)
        P.axis_order = 1;
        P.axis_tab = h->hp.axis1_tab;
        P.axis_cref = h->hp.axis1_cref;
        P.axis_rpa = h->hp.axis1_rpa;
        P.axis_const = h->hp.axis1_const;
    }
    P.lane_cap = 0;
    P.lane_rest = -1;
    P.seen_out = nullptr;
    P.seen_src0 = P.seen_src1 = nullptr;
    P.lane_handover = 0;
    P.lane_spec = 0;
    P.lane_ws = nullptr;
    P.lane_ws2 = nullptr;
    P.lane_list = nullptr;
    P.lane_count = P.lane_zero = nullptr;
    P.lane_hist = nullptr;
    P.lane_bp = 0;
    P.ws = h->d_ws;
    P.model_out = nullptr;
    P.model = nullptr;
    for (int k = 0; k < kMaxCosts; ++k) P.cost_p[k] = h->cost_p[k];
    P.row_f_inst = h->d_row_f_inst;
    P.lb_inst = h->d_lb_inst;
    P.ub_inst = h->d_ub_inst;
    return P;
}

// The overflow queue of a two-tier solve needs a counter that is zero when the first tier starts.  There are two, used in
// turn: a first-tier kernel that can do so (`self_reset`: the Riccati-factor tier) zeroes the OTHER one while it runs -- nobody
// reads it then (the second tier of the previous solve, which did, is behind in the stream) --, so that the next solve finds
// a clean counter and no hipMemsetAsync has to sit in the stream between two solves (a dispatch of its own: ~ 6 us with its
// gap, 1 % of the headline step).  Otherwise: one memset, as before.
static hipError_t begin_overflow_queue(copra_batch* h, hipStream_t s, bool self_reset, FusedPlan& P)
{
    int cur = h->ovf_clean[0] ? 0 : h->ovf_clean[1] ? 1 : -1;
    if (cur < 0) {
        cur = 0;
        const hipError_t e = hipMemsetAsync(h->d_ovf_count, 0, sizeof(int), s);
        if (e != hipSuccess) return e;
    }
    h->ovf_clean[cur] = false; // (appended to from now on)
    h->ovf_cur = cur;
    P.ovf_count = h->d_ovf_count + cur;
    P.ovf_zero = nullptr;
    if (self_reset) {
        P.ovf_zero = h->d_ovf_count + (cur ^ 1);
        h->ovf_clean[cur ^ 1] = true;
    }
    return hipSuccess;
}

// ---- the one-instance-per-lane pass in front of the Riccati-factor tier (lmpc_lane.hpp) ----
static fused_kernel_t select_lane_kernel(const FusedPlan& P)
{
    // <NX, NU, reference trajectories, takes the first steps of the iteration itself (FusedPlan::lane_spec)>
#define COPRA_LANE_PICK(NX, NU)                                                                                                                   \
    if (P.nx == NX && P.nu == NU)                                                                                                                  \
        return P.lane_spec ? (P.stage_refs ? copra_lmpc_lane_kernel<NX, NU, true, true> : copra_lmpc_lane_kernel<NX, NU, false, true>)            \
                           : (P.stage_refs ? copra_lmpc_lane_kernel<NX, NU, true, false> : copra_lmpc_lane_kernel<NX, NU, false, false>);
    COPRA_LANE_PICK(6, 3) // (the CoM system)
    COPRA_LANE_PICK(2, 1) // (the reference's falling-mass system: BASELINE configs[1])
    COPRA_LANE_PICK(4, 2) // (a planar point mass; copra_hip_ric.hip)
#undef COPRA_LANE_PICK
    return nullptr;
}
static fused_kernel_t select_lane_shared_kernel(const FusedPlan& P)
{
    if (P.nx == 6 && P.nu == 3) return P.lane_spec ? copra_lmpc_lane_shared_kernel<6, 3, true> : copra_lmpc_lane_shared_kernel<6, 3, false>;
    if (P.nx == 4 && P.nu == 2) return P.lane_spec ? copra_lmpc_lane_shared_kernel<4, 2, true> : copra_lmpc_lane_shared_kernel<4, 2, false>; // (the other shapes of the tier's run-time-horizon builds)
    if (P.nx == 2 && P.nu == 1) return P.lane_spec ? copra_lmpc_lane_shared_kernel<2, 1, true> : copra_lmpc_lane_shared_kernel<2, 1, false>;
    return nullptr;
}
// ---- the one-(instance, axis)-per-lane solver (lmpc_axis.hpp): the whole solve of a controller whose axes are decoupled ----
static fused_kernel_t select_axis_kernel(const FusedPlan& P)
{
    const int nmax = axis_solver_nmax(P.nx, P.nu, P.N);
    if (nmax == 31 && P.nx == P.nu) // (one state per control in the plane, horizons up to 31)
        return (P.axis_const && P.axis_rpa <= 1) ? copra_lmpc_axis_kernel<1, 2, 31, kAxisQmax, false, true, 1> : copra_lmpc_axis_kernel<1, 2, 31, kAxisQmax, false, false, 2>;
    if (nmax == 20 && (P.nx == 3 * P.nu || P.nx == P.nu)) { // (copra_hip_axis3.hip: tables in registers with one row per axis and step, or read from LDS)
        const bool ct = P.axis_const && P.axis_rpa <= 1;
#define COPRA_AXIS_PICK3(NXA, NU) (ct ? copra_lmpc_axis_kernel<NXA, NU, 20, kAxisQmax, false, true, 1> : copra_lmpc_axis_kernel<NXA, NU, 20, kAxisQmax, false, false, 2>)
        if (P.nx == P.nu) return P.nu == 2 ? COPRA_AXIS_PICK3(1, 2) : COPRA_AXIS_PICK3(1, 3);
        return P.nu == 2 ? COPRA_AXIS_PICK3(3, 2) : COPRA_AXIS_PICK3(3, 3);
#undef COPRA_AXIS_PICK3
    }
    if (nmax == 21) // (three axes at the last horizon of the one-wave kernels: tables in registers with one row per axis and step, or read from LDS)
        return (P.axis_const && P.axis_rpa <= 1) ? copra_lmpc_axis_kernel<2, 3, 21, kAxisQmax, false, true, 1> : copra_lmpc_axis_kernel<2, 3, 21, kAxisQmax, false, false, 2>;
#define COPRA_AXIS_PICK(NU, NMAX, EXACT)                                                                                                     \
    (P.axis_const ? (P.axis_rpa <= 1 ? copra_lmpc_axis_kernel<2, NU, NMAX, kAxisQmax, EXACT, true, 1> : copra_lmpc_axis_kernel<2, NU, NMAX, kAxisQmax, EXACT, true, 2>) \
                  : copra_lmpc_axis_kernel<2, NU, NMAX, kAxisQmax, false, false, 2>)
    if (nmax == 20) return P.nu == 3 ? (P.N == 20 && !P.stage_refs ? COPRA_AXIS_PICK(3, 20, true) : COPRA_AXIS_PICK(3, 20, false)) : COPRA_AXIS_PICK(2, 20, false); // (reference trajectories: the run-time-horizon builds)
    if (nmax == 31) return COPRA_AXIS_PICK(2, 31, false);
#undef COPRA_AXIS_PICK
    return nullptr;
}
// ... and the second chance of what it lists: room for kAxisQmaxBig active constraints per lane
static fused_kernel_t select_axis_list_kernel(const FusedPlan& P)
{
    const int nmax = axis_solver_nmax(P.nx, P.nu, P.N);
    if (nmax == 31 && P.nx == P.nu) return copra_lmpc_axis_list_kernel<1, 2, 31, kAxisQmaxBig, false, 2>;
    if (nmax == 20 && P.nx == P.nu)
        return P.nu == 2 ? copra_lmpc_axis_list_kernel<1, 2, 20, kAxisQmaxBig, false, 2> : copra_lmpc_axis_list_kernel<1, 3, 20, kAxisQmaxBig, false, 2>;
    if (nmax == 20 && P.nx == 3 * P.nu)
        return P.nu == 2 ? copra_lmpc_axis_list_kernel<3, 2, 20, kAxisQmaxBig, false, 2> : copra_lmpc_axis_list_kernel<3, 3, 20, kAxisQmaxBig, false, 2>;
    if (nmax == 21) return copra_lmpc_axis_list_kernel<2, 3, 21, kAxisQmaxBig, false, 2>;
#define COPRA_AXIS_LPICK(NU, NMAX) (P.axis_const ? copra_lmpc_axis_list_kernel<2, NU, NMAX, kAxisQmaxBig, true, 2> : copra_lmpc_axis_list_kernel<2, NU, NMAX, kAxisQmaxBig, false, 2>)
    if (nmax == 20) return P.nu == 3 ? COPRA_AXIS_LPICK(3, 20) : COPRA_AXIS_LPICK(2, 20);
    if (nmax == 31) return COPRA_AXIS_LPICK(2, 31);
#undef COPRA_AXIS_LPICK
    return nullptr;
}
static size_t axis_list_lds_bytes(const FusedPlan& P)
{
    int oB = 0, oR = 0, rcs = 0;
    return (size_t)axis_lds_doubles(P.nx, P.nu, P.N, P.axis_rpa, kAxisQmaxBig, oB, oR, rcs) * sizeof(double);
}
static size_t axis_lds_bytes(const FusedPlan& P)
{
    int oB = 0, oR = 0, rcs = 0;
    return (size_t)axis_lds_doubles(P.nx, P.nu, P.N, P.axis_rpa, kAxisQmax, oB, oR, rcs) * sizeof(double);
}
static bool axis_solver_wanted(const copra_batch* h, const FusedPlan& P)
{
    const copra_options_t& opt = h->hp.opt;
    if (h->ad.axis_off || opt.no_axis_solver || opt.no_lane_pass || P.axis_tab < 0 || P.prof_fine) return false;
    if (opt.lane_min_batch > 0 && P.batch < opt.lane_min_batch) return false;
    if (h->packed || h->shared || h->hp.large || P.initial_state) return false;
    // (per-instance limits: the builds that keep bounds and right-hand sides in registers take this lane's own -- where they are the same
    //  along the horizon, else the instance goes to the tier: lmpc_axis.hpp)
    if ((P.row_f_inst || P.lb_inst || P.ub_inst) && (!P.axis_const || (P.lb_inst == nullptr) != (P.ub_inst == nullptr))) return false;
    for (int t = 0; t < kMaxCosts; ++t) // (per-instance references: a lane rebuilds the affine terms of its axis from them -- FusedPlan::axis_cref)
        if (h->cost_p[t] && (P.axis_cref < 0 || t >= P.ncost)) return false;
    if (P.stage_refs) { // reference trajectories: the stages' h wait in the lane's sparse array for the sweep (lmpc_axis.hpp)
        int oB = 0, oR = 0, rcs = 0;
        (void)axis_lds_doubles(P.nx, P.nu, P.N, P.axis_rpa, kAxisQmax, oB, oR, rcs);
        if (P.axis_cref < 0 || P.N * (P.nx / P.nu + 1) > rcs) return false;
    }
    return select_axis_kernel(P) != nullptr;
}
static size_t lane_lds_bytes(const FusedPlan& P)
{
    int oH = 0;
    return (size_t)(lane_lds_doubles(P.nx, P.nu, oH) + P.lane_tlds) * sizeof(double);
}
// does the next solve run it?  (the per-instance references and right-hand sides can be set at any time: checked per solve)
// The pass holds 64 instances per wave at ONE wave per SIMD, and a wave of it runs ~ 90 us whatever the batch: it pays from about a third
// of the machine's SIMDs on (measured at the headline shape, first tier alone -> with the pass: batch 2048: 0.051 -> 0.174 ms, 8192: 0.125 ->
// 0.212, 16384: 0.208 -> 0.241, 24576: 0.293 -> 0.276, 32768: 0.383 -> 0.309, 65536: 0.71 -> 0.49).  Smaller batches -- the single problem of
// copra::LMPC::solve() above all -- keep the wave-per-instance tier alone.
// In front of the OTHER one-wave first tiers (ten times slower per instance than the Riccati-factor tier) it pays from a few thousand
// instances on (tools/exp/lane_threshold.py: falling mass N = 32: 2048: 0.49 -> 0.38 ms, 16384: 2.78 -> 2.06; N = 64: 1.27 -> 0.08 ms).
// (shared-model controllers with per-instance references, refs: without the pass they run lmpc_shared.hpp, not the tier alone -- the pass
//  with its delta sweep pays from about 10 k instances on: 8192: 0.129 vs 0.132 ms, 16384: 0.198 vs 0.160, profiles/r04/shared_goals_batches.txt)
static bool lane_batch_ok(const copra_options_t& opt, int batch, bool ric_tier, bool refs = false)
{
    int least = ric_tier ? (refs ? 10240 : 20480) : 4096;
    if (opt.lane_min_batch != 0) least = opt.lane_min_batch < 0 ? 0 : opt.lane_min_batch;
    return batch >= least;
}
static bool lane_pass_wanted(const copra_batch* h, const FusedPlan& P, bool jit_launch)
{
    const copra_options_t& opt = h->hp.opt;
    if (h->ad.lane_off || opt.no_lane_pass || !lane_batch_ok(opt, P.batch, P.lds.ric != 0)) return false;
    if (P.prof_fine) return false; // (the fine-grained stamps of the profiling build follow ONE kernel through a whole solve)
    // (in front of the Riccati-factor tier, which takes the factor over, or of any other one-wave first tier, where it only filters)
    if (P.lane_tab < 0 || (jit_launch && !h->jit_ric) || h->packed || h->shared || h->hp.large || P.initial_state) return false;
    for (int t = 0; t < kMaxCosts; ++t)
        if (h->cost_p[t] && P.lane_cref < 0) return false; // (per-instance references: the pass rebuilds its affine terms per lane)
    if (P.stage_refs && P.lane_cref < 0) return false; // (reference trajectories: ... per lane and stage)
    return (jit_launch ? h->jit_lane != nullptr : select_lane_kernel(P) != nullptr);
}
static copra_status_t ensure_lane_buffers(copra_batch* h, bool need_ws)
{
    const FusedPlan& P = h->hp.plan;
    const size_t bp = ((size_t)P.batch + kWave - 1) / kWave * kWave + kWave; // (+ 64 spare columns: what lanes without an instance write)
    // every buffer is tested on its own, and a failed attempt leaves NONE behind (round-3 advisor finding: with the counters allocated
    // and the list not, the next call returned COPRA_OK with a null list)
    hipError_t e = hipSuccess;
    if (!h->d_lane_count || !h->d_lane_list || !h->d_lane_hist) {
        (void)hipFree(h->d_lane_count);
        (void)hipFree(h->d_lane_list);
        (void)hipFree(h->d_lane_hist);
        h->d_lane_count = h->d_lane_list = h->d_lane_hist = nullptr;
        e = hipMalloc((void**)&h->d_lane_count, 4 * sizeof(int)); // [cur | other]: instances left to the tier; [2 + cur | 2 + other]: instances ended by the pass's own steps
        if (e == hipSuccess) e = hipMemset(h->d_lane_count, 0, 4 * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void**)&h->d_lane_list, bp * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void**)&h->d_lane_hist, kLaneHistBins * sizeof(int));
        if (e == hipSuccess) e = hipMemset(h->d_lane_hist, 0, kLaneHistBins * sizeof(int));
        h->lane_cur = 1;
    }
    if (e == hipSuccess && need_ws && !h->d_lane_ws) // (the shared-model form of the pass has no sweep: no workspace)
        e = hipMalloc((void**)&h->d_lane_ws, (size_t)P.N * lane_ws_rows(P.nx, P.nu) * bp * sizeof(double));
    const bool hand_over = P.lds.ricC && !h->hp.opt.no_lane_handover && (h->hp.opt.no_lane_spec || h->ad.lane_form_handover); // (solve_one_wave: the form of the pass that hands blocks over)
    if (e == hipSuccess && need_ws && hand_over && !h->d_lane_ws2) { // (the hand-over blocks: what only the first tier reads, instance-major)
        // (round-5 advisor: 126 MB at 65 536 instances -- without room for them the pass keeps its speculating form, which needs none, instead of
        //  being switched off for good with everything else freed)
        if (hipMalloc((void**)&h->d_lane_ws2, bp * (size_t)lane_ws2_doubles(P.nx, P.nu, P.N) * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            h->d_lane_ws2 = nullptr;
            if (h->hp.opt.no_lane_spec) e = hipErrorOutOfMemory; // (no other form allowed: the tier alone)
            else h->ad.lane_form_handover = false, h->ad.lane_ws2_failed = true;
        }
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(h->d_lane_count);
        (void)hipFree(h->d_lane_list);
        (void)hipFree(h->d_lane_hist);
        (void)hipFree(h->d_lane_ws);
        (void)hipFree(h->d_lane_ws2);
        h->d_lane_count = h->d_lane_list = h->d_lane_hist = nullptr;
        h->d_lane_ws = h->d_lane_ws2 = nullptr;
        return fail(COPRA_ERR_HIP, std::string("one-instance-per-lane pass: ") + hipGetErrorString(e));
    }
    return COPRA_OK;
}
// The pass pays when a fair share of the batch ends in it.  After each of the first solves that ran it: if fewer than one instance in
// eight did, it is switched off for this controller (it costs about a tenth of the first tier per instance).
static copra_status_t adapt_lane_pass(copra_batch* h)
{
    // The decision is not final (round-3 advisor finding: a controller whose first ticks are a constrained transient lost the pass for
    // good, one that became constraint-heavy later kept paying for it): every kLaneResample solves the share is sampled again -- a pass
    // that was switched off by THIS function (not for lack of memory) runs once more, a running one is checked again.
    constexpr int kLaneResample = 256;
    h->ad.lane_solves += 1;
    if (h->ad.lane_solves % kLaneResample == 0 && h->ad.lane_adapt_left <= 0) {
        h->ad.lane_adapt_left = 1;
        if (h->ad.lane_off && h->ad.lane_off_by_share) h->ad.lane_off = h->ad.lane_off_by_share = false;
        h->ad.lane_form_handover = false; // (the speculating form is tried again)
        h->ad.lane_spec_shared_off = false;
    }
    if (!h->ad.lane_ran || h->ad.lane_adapt_left <= 0) return COPRA_OK;
    // In front of the compact variant of the Riccati-factor tier the pass has two forms (solve_one_wave).  The one that hands the factor
    // over pays for every instance -- it is the tier's sweep, done at several times the efficiency: nothing to decide.  The speculating form
    // (the default) hands nothing over: it pays where most instances END in it; where fewer than one in four do -- constraint-heavy
    // workloads: the first picks are state rows, or the iteration goes on after two bounds on u_0 -- the controller moves to the hand-over
    // form (measured at v_max 0.25 / u_max 1.2, where nothing ends in the pass: 23.7 M solves/s with the hand-over, 20.2 M with the tier
    // sweeping for itself).
    if (!h->shared && h->hp.plan.lds.ricC && !h->hp.opt.no_lane_handover) {
        if (h->hp.opt.no_lane_spec || h->ad.lane_form_handover) return COPRA_OK;
        h->ad.lane_adapt_left -= 1;
        int left_over = 0, by_steps = 0;
        HIP_TRY(hipStreamSynchronize(h->last_stream));
        HIP_TRY(hipMemcpy(&left_over, h->d_lane_count + h->lane_cur, sizeof(int), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&by_steps, h->d_lane_count + 2 + h->lane_cur, sizeof(int), hipMemcpyDeviceToHost));
        const long long ended = (long long)h->hp.plan.batch - left_over;
        // (... and where the steps it takes itself end fewer than one instance in twelve, they do not pay for the two trajectories that ride
        //  along -- + 45 us per 65 536 instances against 8 ns per instance the tier is spared: the hand-over form does not carry them)
        if ((ended * 4 < (long long)h->hp.plan.batch || (long long)by_steps * 12 < (long long)h->hp.plan.batch) && !h->ad.lane_ws2_failed) h->ad.lane_form_handover = true;
        if (h->hp.opt.debug)
            fprintf(stderr, "[copra] one-instance-per-lane pass (speculating form): %lld of %d instances ended in it, %d of them by its own steps%s\n", ended,
                h->hp.plan.batch, by_steps, h->ad.lane_form_handover ? " -- the hand-over form from now on" : "");
        return COPRA_OK;
    }
    if (h->shared && !h->ad.lane_spec_shared_off) { // the shared-model form's own steps: kept while they end one instance in eight (measured: a tracking
        int by_steps = 0;                           // controller whose instances end at their minimiser ran 0.247 ms with them, 0.200 ms without)
        HIP_TRY(hipStreamSynchronize(h->last_stream));
        HIP_TRY(hipMemcpy(&by_steps, h->d_lane_count + 2 + h->lane_cur, sizeof(int), hipMemcpyDeviceToHost));
        if ((long long)by_steps * 8 < (long long)h->hp.plan.batch) h->ad.lane_spec_shared_off = true;
        if (h->hp.opt.debug)
            fprintf(stderr, "[copra] shared-model pass: %d of %d instances ended by its own steps%s\n", by_steps, h->hp.plan.batch,
                h->ad.lane_spec_shared_off ? " -- the plain form from now on" : "");
    }
    // (... and a shared-model controller with per-instance references: the pass carries their delta sweep, without it the controller runs
    //  lmpc_shared.hpp, which the tier beats by 2 x whatever the share that ends in the pass)
    if (h->shared && h->shared_ric)
        for (int t = 0; t < kMaxCosts; ++t)
            if (h->cost_p[t]) {
                h->ad.lane_adapt_left -= 1; // (the decision about its own steps above was this sample's)
                return COPRA_OK;
            }
    h->ad.lane_adapt_left -= 1;
    int left = 0;
    HIP_TRY(hipStreamSynchronize(h->last_stream));
    HIP_TRY(hipMemcpy(&left, h->d_lane_count + h->lane_cur, sizeof(int), hipMemcpyDeviceToHost));
    const long long done = (long long)h->hp.plan.batch - left;
    const long long share = h->shared ? 4 : 8;
    if (done * share < (long long)h->hp.plan.batch) h->ad.lane_off = h->ad.lane_off_by_share = true;
    if (h->hp.opt.debug)
        fprintf(stderr, "[copra] one-instance-per-lane pass: %lld of %d instances ended in it%s\n", done, h->hp.plan.batch,
            h->ad.lane_off ? " -- switched off" : "");
    return COPRA_OK;
}

// The one-(instance, axis)-per-lane solver pays where it FINISHES instances: what it lists is solved again, from scratch, by the tier.  After
// each of a controller's first two solves on it: if it left more than half of the batch over (systems whose axes are coupled after all, active
// sets beyond its lanes' room) the controller goes back to the one-instance-per-lane pass + tier; sampled again every 256 solves.
static copra_status_t adapt_axis_solver(copra_batch* h)
{
    constexpr int kResample = 256;
    h->ad.axis_solves += 1;
    if (h->ad.axis_solves % kResample == 0 && h->ad.axis_adapt_left <= 0) {
        h->ad.axis_adapt_left = 1;
        if (h->ad.axis_off && h->ad.axis_off_by_share) h->ad.axis_off = h->ad.axis_off_by_share = false;
    }
    if (!h->ad.axis_ran || h->ad.axis_adapt_left <= 0) return COPRA_OK;
    h->ad.axis_adapt_left -= 1;
    int left = 0;
    HIP_TRY(hipStreamSynchronize(h->last_stream));
    HIP_TRY(hipMemcpy(&left, h->d_lane_count + h->lane_cur, sizeof(int), hipMemcpyDeviceToHost));
    if ((long long)left * 2 > (long long)h->hp.plan.batch) h->ad.axis_off = h->ad.axis_off_by_share = true;
    if (h->hp.opt.debug)
        fprintf(stderr, "[copra] (instance, axis)-per-lane solver: %d of %d instances left to the tier%s\n", left, h->hp.plan.batch,
            h->ad.axis_off ? " -- switched off" : "");
    return COPRA_OK;
}

// Compact LDS layouts (R capped, overflow finished by the second tier) bet on small active sets.  After each of the
// first solves the overflow queue tells whether the bet holds; if more than one instance in eight had to be redone by
// the second tier, step to the next safer layout: dense -> safe (quarter-CU compact or full) -> full.
static copra_status_t adapt_layout(copra_batch* h)
{
    if (!h->hp.two_tier || h->hp.large || h->ad.adapt_left <= 0 || !h->ad.solved_once) return COPRA_OK;
    h->ad.adapt_left -= 1;
    int count = 0;
    HIP_TRY(hipStreamSynchronize(h->last_stream));
    HIP_TRY(hipMemcpy(&count, h->d_ovf_count + h->ovf_cur, sizeof(int), hipMemcpyDeviceToHost));
    // (the Riccati-factor tier is so much faster than its second tier -- the square-layout kernel -- that it pays to step down
    //  the ladder until only one instance in 64 is left over -- measured in round 3 over five constraint levels: 64 and 128 equal, 32
    //  leaves a mid-constrained workload on five columns at 42.8 instead of 50.4 M solves/s, 512 goes too far; the other first tiers
    //  keep the round-1 threshold of one in 8)
    const long long share = h->hp.plan.lds.ric ? 64 : 8;
    if ((long long)count * share <= (long long)h->hp.plan.batch) return COPRA_OK;
    LdsLayout roomier {};
    if (next_tri_layout(h->hp.plan, h->hp.plan.lds, roomier, h->hp.opt.no_ladder != 0)) { // factor-only: one instance per CU fewer, more columns
        h->hp.plan.lds = roomier;
        h->hp.lds_bytes = (size_t)roomier.total * sizeof(double);
        h->lds_attr_set = false;
        h->ad.adapt_left += 1; // (a step down the ladder does not use up the budget of attempts)
        if (h->hp.opt.debug)
            fprintf(stderr, "[copra] %d of %d instances overflowed the factor-only layout: next %zu B, %d columns\n", count,
                h->hp.plan.batch, h->hp.lds_bytes, roomier.rcap);
        return COPRA_OK;
    }
    if (h->hp.dense && h->hp.safe_two_tier) { // dense -> quarter-CU compact
        h->hp.plan.lds = h->hp.lds_safe;
        h->hp.two_tier = true;
    } else { // -> full layout, single tier
        h->hp.plan.lds = h->hp.lds_full;
        h->hp.two_tier = false;
    }
    h->hp.dense = false;
    h->hp.lds_bytes = (size_t)h->hp.plan.lds.total * sizeof(double);
    h->lds_attr_set = false;
    h->shared_attr_set = false;
    if (h->shared_ric || h->has_lds_ric) {
        // the ladder of Riccati-factor layouts is exhausted: a shared-model controller leaves that tier for good (its first
        // tier is lmpc_shared.hpp from now on, whose model -- J, G, Phi, xi -- has to be prepared again for the new layout)
        h->shared_ric = false;
        h->has_lds_ric = false;
        h->model_dirty = true;
    }
    const FusedPlan& P = h->hp.plan;
    h->packed = (P.lds.tri || h->hp.opt.no_packed) ? 0
        : packed_width(P.n, P.nx * (P.nx + P.nu + 1), P.rfull > 0, h->hp.lds_bytes);
    if (h->hp.opt.debug)
        fprintf(stderr, "[copra] %d of %d instances overflowed the compact LDS layout: next layout %zu B, %s\n", count, P.batch,
            h->hp.lds_bytes, h->hp.two_tier ? "two-tier" : "single tier");
    return COPRA_OK;
}

// The ladder only leads down, and the first solve's choice rests on a proxy (rows violated by the unconstrained minimiser: an instance
// that must brake for ten steps violates ten rows and ends with one or two active).  Once a solve is behind the controller the SIZES of
// the final active sets are known exactly -- adds minus drops, from the iteration counters every instance reports anyway.  After the
// first solve, and every 256 solves from then on, the first tier's layout is chosen again FROM THE TOP of its ladder: the densest step
// that leaves at most one instance in `share` to the second tier.  One synchronisation and one copy of 8 B per instance each time;
// adapt_layout keeps checking the overflow counts of the next solves behind it (an active set can peak above its final size).
static copra_status_t rechoose_layout(copra_batch* h)
{
    constexpr long long kPeriod = 256;
    const FusedPlan& P = h->hp.plan;
    if (!h->hp.two_tier || h->hp.large || h->shared || !P.lds.tri || h->hp.opt.no_ladder || P.batch <= 0 || P.initial_state) return COPRA_OK;
    if (!h->ad.lds_top_set) { // (first solve of this ladder: remember where it starts)
        h->ad.lds_top = P.lds;
        h->ad.lds_top_set = true;
        h->ad.layout_solves = 0;
        return COPRA_OK;
    }
    h->ad.layout_solves += 1; // solves completed on this ladder
    if (h->ad.layout_solves != 1 && h->ad.layout_solves % kPeriod != 0) return COPRA_OK;
    if (!(h->ad.lds_top.tri && h->ad.lds_top.ric == P.lds.ric && h->ad.lds_top.ricC == P.lds.ricC)) return COPRA_OK;
    const int* d_it = h->ad.last_iter; // (where the LAST solve wrote: with rotating result slabs not the buffers set for the next one)
    const int* d_st = h->ad.last_status;
    if (!d_it || !d_st) return COPRA_OK;
    std::vector<int> it((size_t)P.batch * 2), st((size_t)P.batch);
    HIP_TRY(hipStreamSynchronize(h->last_stream));
    HIP_TRY(hipMemcpy(it.data(), d_it, it.size() * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(st.data(), d_st, st.size() * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<long long> larger((size_t)P.n + 2, 0); // larger[a] = instances whose final active set holds MORE than a rows
    for (int b = 0; b < P.batch; ++b) {
        if (st[(size_t)b] != 0) continue;
        int a = it[2 * (size_t)b] - 1 - it[2 * (size_t)b + 1];
        a = a < 0 ? 0 : (a > P.n ? P.n : a);
        if (a > 0) larger[(size_t)a - 1] += 1;
    }
    for (int a = P.n - 1; a >= 0; --a) larger[(size_t)a] += larger[(size_t)a + 1];
    const long long share = P.lds.ric ? 64 : 8;
    LdsLayout pick = h->ad.lds_top;
    for (;;) {
        const int cap = pick.rcap < P.n ? pick.rcap : P.n;
        LdsLayout roomier {};
        if (larger[(size_t)cap] * share <= (long long)P.batch || !next_tri_layout(P, pick, roomier)) break;
        pick = roomier;
    }
    if (pick.total != P.lds.total || pick.rcap != P.lds.rcap) {
        if (h->hp.opt.debug)
            fprintf(stderr, "[copra] final active sets of the last solve: first tier from %d to %d columns (%zu B)\n", P.lds.rcap, pick.rcap,
                (size_t)pick.total * sizeof(double));
        h->hp.plan.lds = pick;
        h->hp.lds_bytes = (size_t)pick.total * sizeof(double);
        h->lds_attr_set = false;
        if (h->ad.adapt_left < 2) h->ad.adapt_left = 2;
    }
    return COPRA_OK;
}

copra_status_t ensure_lds_attr(copra_batch* h)
{
    if (h->hp.ric_only) return COPRA_OK; // (the Riccati kernels opt in to their LDS where they are launched)
    if (h->hp.large) {
        if (!h->lds_attr_set && h->hp.lds_bytes > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(h->large_fn),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->hp.lds_bytes));
        h->lds_attr_set = true;
        return COPRA_OK;
    }
    const size_t need = h->hp.lds_full_bytes > h->hp.lds_bytes ? h->hp.lds_full_bytes : h->hp.lds_bytes;
    if (!h->lds_attr_set && need > 48 * 1024) {
        const void* fn = h->hp.plan.initial_state ? reinterpret_cast<const void*>(copra_islmpc_fused_kernel)
                                                  : reinterpret_cast<const void*>(select_fused_kernel(h->hp.plan));
        HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
        if (!h->hp.plan.initial_state && h->hp.two_tier)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(select_tier2_kernel(h->hp.plan)),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    }
    h->lds_attr_set = true;
    return COPRA_OK;
}

// (Re)build the stage plan of a controller and put its tables on the device.  The plan depends on whether per-instance
// control bounds exist (then every control gets both bound rows, infinite ones are switched off per instance).
copra_status_t prepare_riccati(copra_batch* h)
{
    const bool all_bounds = h->d_lb_inst != nullptr;
    bool refs = false; // per-instance cost references: q_k differs per instance, which only the streaming kernel evaluates
    for (int t = 0; t < kMaxCosts; ++t) refs = refs || h->cost_p[t] != nullptr;
    if (h->ric_built && h->ric_all_bounds == all_bounds && h->ric_refs == refs) return COPRA_OK;
    h->ric_refs = refs;
    for (void* q : h->ric_dev) (void)hipFree(q);
    h->ric_dev.clear();
    (void)hipFree(h->d_ric_ws);
    h->d_ric_ws = nullptr;
    build_stage_plan(h->hp, h->hs, all_bounds);
    h->ric_built = false; // (set once every table and the workspace are on the device: a failed attempt must not look prepared)
    h->ric_all_bounds = all_bounds;
    if (!h->hs.eligible) {
        h->ric_built = true;
        return COPRA_OK;
    }
    HostStagePlan& hs = h->hs;
    hipError_t e = hipSuccess;
    auto upi = [&](const std::vector<int>& v) -> const int* {
        int* dptr = nullptr;
        hipError_t r = upload(&dptr, v);
        if (r != hipSuccess && e == hipSuccess) e = r;
        h->ric_dev.push_back(dptr);
        return dptr;
    };
    auto upd = [&](const std::vector<double>& v) -> const double* {
        double* dptr = nullptr;
        hipError_t r = upload(&dptr, v);
        if (r != hipSuccess && e == hipSuccess) e = r;
        h->ric_dev.push_back(dptr);
        return dptr;
    };
    StagePlan& sp = hs.sp;
    sp.cls_of_stage = upi(hs.cls_of_stage);
    sp.stage_row0 = upi(hs.stage_row0);
    sp.cls_W = upi(hs.cls_W);
    sp.cls_crow0 = upi(hs.cls_crow0);
    sp.cls_row0 = upi(hs.cls_row0);
    sp.cls_ndense = upi(hs.cls_ndense);
    sp.cr_aoff = upi(hs.cr_aoff);
    sp.cr_cost = upi(hs.cr_cost);
    sp.cr_pidx = upi(hs.cr_pidx);
    sp.cr_w = upd(hs.cr_w);
    sp.r_kind = upi(hs.r_kind);
    sp.r_aoff = upi(hs.r_aoff);
    sp.r_sign = upd(hs.r_sign);
    sp.r_eq = upi(hs.r_eq);
    sp.r_src = upi(hs.r_src);
    sp.r_sidx = upi(hs.r_sidx);
    sp.r_sstride = upi(hs.r_sstride);
    sp.blob = upd(hs.blob);
    sp.iblob = upi(hs.iblob);
    sp.cls_rptr = upi(hs.cls_rptr), sp.cls_rcol = upi(hs.cls_rcol), sp.cls_gptr = upi(hs.cls_gptr), sp.cls_grow = upi(hs.cls_grow);
    sp.cls_eptr = upi(hs.cls_eptr), sp.cls_erow = upi(hs.cls_erow);
    sp.cls_rval = upi(hs.cls_rval), sp.cls_gval = upi(hs.cls_gval), sp.cls_eval = upi(hs.cls_eval);
    h->ric_fast = sp.fast_ok && !refs && !h->hp.opt.no_ric_fast;
    if (h->ric_fast) { // fixed-width tables of the LDS-resident kernel
        sp.f_rinfo = upi(hs.f_rinfo), sp.f_rcomp = upi(hs.f_rcomp), sp.f_rval = upd(hs.f_rval);
        sp.f_gcnt = upi(hs.f_gcnt), sp.f_grow = upi(hs.f_grow), sp.f_gval = upd(hs.f_gval);
        sp.f_wcol = upi(hs.f_wcol), sp.f_wval = upd(hs.f_wval), sp.f_Wp = upi(hs.f_Wp);
        sp.f_tptr = upi(hs.f_tptr), sp.f_tent = upi(hs.f_tent), sp.f_trow = upi(hs.f_trow), sp.f_tval = upd(hs.f_tval);
        sp.f_qcnt = upi(hs.f_qcnt), sp.f_qoff = upi(hs.f_qoff), sp.f_qa = upd(hs.f_qa), sp.f_q = upd(hs.f_q);
    }
    // persistent grid: as many one-wave workgroups as the device keeps resident
    const size_t lds_bytes = (size_t)(h->ric_fast ? sp.fast_lds_doubles : sp.lds_doubles) * sizeof(double);
    const void* ric_fn = h->ric_fast ? reinterpret_cast<const void*>(copra_lmpc_riccati_mfma_kernel)
                                     : reinterpret_cast<const void*>(select_riccati_kernel(sp.nx, sp.nu));
    if (e == hipSuccess) e = lds_opt_in(ric_fn, lds_bytes);
    int dev = 0, cus = 256, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ric_fn, 64, lds_bytes)
            != hipSuccess
        || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    long long g = (long long)cus * per_cu;
    const int batch = h->hp.plan.batch > 0 ? h->hp.plan.batch : 1;
    h->ric_grid = (int)(g < batch ? g : batch);
    if (e == hipSuccess) // the streaming kernel's workspace | the LDS-resident kernel's stage records (all but a ring of four wait there: N x 107 doubles per wave)
        e = hipMalloc((void**)&h->d_ric_ws, (size_t)h->ric_grid * (h->ric_fast ? (size_t)sp.N * kRfKStride : (size_t)sp.ws_total) * sizeof(double));
    sp.ws = h->ric_fast ? nullptr : h->d_ric_ws;
    sp.rec_ws = h->ric_fast ? h->d_ric_ws : nullptr;
    if (e == hipSuccess && !h->d_ric_next) e = hipMalloc((void**)&h->d_ric_next, sizeof(int));
    sp.next_instance = h->d_ric_next;
    if (h->hp.opt.debug)
        fprintf(stderr, "[copra] riccati path (%s): %d classes, %d rows, grid %d (%d per CU), %zu B LDS, %lld B workspace per wave\n",
            h->ric_fast ? "LDS-resident, MFMA" : hs.fast_why.c_str(), sp.ncls, sp.m, h->ric_grid, per_cu, lds_bytes,
            h->ric_fast ? 0LL : sp.ws_total * 8LL);
    if (e != hipSuccess) {
        for (void* q : h->ric_dev) (void)hipFree(q);
        h->ric_dev.clear();
        (void)hipFree(h->d_ric_ws);
        h->d_ric_ws = nullptr;
        h->hs.eligible = false;
        h->hs.why = "device tables of the stage plan could not be allocated";
        (void)hipGetLastError();
        return fail(COPRA_ERR_HIP, std::string("riccati path: ") + hipGetErrorString(e));
    }
    h->ric_built = true;
    return COPRA_OK;
}

// does the next solve of this controller run the Riccati interior-point kernel?
bool use_riccati(copra_batch* h)
{
    if (h->hp.ric_only) return prepare_riccati(h) == COPRA_OK && h->hs.eligible;
    if (h->solver == COPRA_SOLVER_QUADPROG_DENSE || h->shared) return false;
    if (h->solver == COPRA_SOLVER_DEFAULT && (!h->hp.large || h->hp.opt.no_riccati)) return false;
    if (prepare_riccati(h) != COPRA_OK) return false;
    return h->hs.eligible;
}

extern "C" {

int copra_abi_version(void) { return 5; } // 5: + copra_options_t, copra_options_init, copra_set_default_options, copra_batch_create_with_options; 3: + copra_batch_last_first_tier_seconds, copra_batch_set_system_rowmajor_async; 4: + copra_batch_lane_pass_info, copra_batch_set_cost_reference_all





const char* copra_last_error(void) { return g_copra_err.c_str(); }

// (copra_source_hash: copra_hip_hash.hip -- a translation unit of its own that is compiled again whenever ANY source of the library changes;
//  compiled into this one it went stale whenever make rebuilt only another unit, and bench.py's roofline.traffic_stale said so)


static copra_status_t create_common(copra_batch_t** out, const copra_dims_t* dims, int n_costs,
    const copra_cost_desc_t* costs, int n_cstrs, const copra_cstr_desc_t* cstrs, const copra_initial_state_desc_t* is,
    const copra_options_t* opts);

void copra_options_init(copra_options_t* opts)
{
    if (opts) *opts = default_options();
}

copra_status_t copra_set_default_options(const copra_options_t* opts)
{
    copra_options_t builtin {};
    builtin.struct_size = (int)sizeof(copra_options_t);
    default_options() = builtin;
    if (opts) default_options() = resolve_options(opts);
    return COPRA_OK;
}

copra_status_t copra_batch_create(copra_batch_t** out, const copra_dims_t* dims, int n_costs,
    const copra_cost_desc_t* costs, int n_cstrs, const copra_cstr_desc_t* cstrs)
{
    return create_common(out, dims, n_costs, costs, n_cstrs, cstrs, nullptr, nullptr);
}

copra_status_t copra_batch_create_with_options(copra_batch_t** out, const copra_dims_t* dims, int n_costs,
    const copra_cost_desc_t* costs, int n_cstrs, const copra_cstr_desc_t* cstrs, const copra_initial_state_desc_t* is,
    const copra_options_t* opts)
{
    return create_common(out, dims, n_costs, costs, n_cstrs, cstrs, is, opts);
}

copra_status_t copra_batch_create_initial_state(copra_batch_t** out, const copra_dims_t* dims, int n_costs,
    const copra_cost_desc_t* costs, int n_cstrs, const copra_cstr_desc_t* cstrs, const copra_initial_state_desc_t* is)
{
    if (!is) return fail(COPRA_ERR_ARG, "copra_batch_create_initial_state: null descriptor");
    return create_common(out, dims, n_costs, costs, n_cstrs, cstrs, is, nullptr);
}

static copra_status_t create_common(copra_batch_t** out, const copra_dims_t* dims, int n_costs,
    const copra_cost_desc_t* costs, int n_cstrs, const copra_cstr_desc_t* cstrs, const copra_initial_state_desc_t* is,
    const copra_options_t* opts)
{
    if (!out || !dims || n_costs < 0 || n_cstrs < 0 || (n_costs > 0 && !costs) || (n_cstrs > 0 && !cstrs))
        return fail(COPRA_ERR_ARG, "copra_batch_create: null / negative argument");
    *out = nullptr;
    copra_batch* h = new copra_batch();
    h->hp.opt = resolve_options(opts); // (everything the engine consults from here on: no environment variable is read on any path)
    copra_status_t rc = build_plan(h->hp, *dims, n_costs, costs, n_cstrs, cstrs, is);
    if (rc != COPRA_OK) {
        g_copra_err = h->hp.error;
        delete h;
        return rc;
    }
    const FusedPlan& P = h->hp.plan;
    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t r) {
        if (e == hipSuccess) e = r;
    };
    chk(upload(&h->d_row_step, h->hp.row_step));
    chk(upload(&h->d_row_ekind, h->hp.row_ekind));
    chk(upload(&h->d_row_eoff, h->hp.row_eoff));
    chk(upload(&h->d_row_gkind, h->hp.row_gkind));
    chk(upload(&h->d_row_goff, h->hp.row_goff));
    chk(upload(&h->d_row_f, h->hp.row_f));
    chk(upload(&h->d_row_prev, h->hp.row_prev));
    chk(upload(&h->d_params, h->hp.params));
    chk(upload(&h->d_lb, h->hp.lb));
    chk(upload(&h->d_ub, h->hp.ub));
    const size_t b = (size_t)(P.batch > 0 ? P.batch : 1);
    { // result slab [U | X | status | iter], every part 256-byte aligned.  (The slab a multi-GPU caller hands in through
      //  copra_batch_set_outputs is laid out [U | status | iter | X] instead -- copra_amd/sharding.py: there the small parts come first so
      //  that controls + status are a contiguous prefix a gather may send alone; this engine-owned slab is only ever fetched whole.)
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        h->off_traj = up(b * P.n * sizeof(double));
        h->off_status = h->off_traj + up(b * P.X * sizeof(double));
        h->off_iter = h->off_status + up(b * sizeof(int));
        h->results_bytes = h->off_iter + up(b * 2 * sizeof(int));
        chk(hipMalloc((void**)&h->d_results, h->results_bytes));
        if (h->d_results) {
            h->d_control = reinterpret_cast<double*>(h->d_results);
            h->d_traj = reinterpret_cast<double*>(h->d_results + h->off_traj);
            h->d_status = reinterpret_cast<int*>(h->d_results + h->off_status);
            h->d_iter = reinterpret_cast<int*>(h->d_results + h->off_iter);
        }
        if (h->results_bytes <= kSmallSlab) chk(hipHostMalloc((void**)&h->h_results, h->results_bytes, hipHostMallocDefault));
    }
    if (is) {
        chk(upload(&h->d_isR, h->hp.isR));
        chk(upload(&h->d_isr, h->hp.isr));
        chk(hipMalloc((void**)&h->d_x0opt, b * P.nx * sizeof(double)));
    }
    if (h->hp.ric_only) { // only the Riccati interior-point kernels cover this size: the controller must be stage-wise
        const copra_status_t rr = prepare_riccati(h);
        if (rr != COPRA_OK || !h->hs.eligible) {
            const std::string why = rr != COPRA_OK ? g_copra_err : h->hs.why;
            copra_batch_destroy(h);
            *out = nullptr;
            fail(COPRA_ERR_UNSUPPORTED, "more than 512 decision variables (or InitialStateLMPC with xDim > 16) need a stage-wise controller for the Riccati interior-point kernel: " + why);
            return COPRA_ERR_UNSUPPORTED;
        }
    } else if (h->hp.large) {
        h->large_fn = choose_large_kernel(h->hp);
        if (h->hp.lds_bytes > 48 * 1024) // (the occupancy query below needs the attribute as well)
            chk(hipFuncSetAttribute(reinterpret_cast<const void*>(h->large_fn),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->hp.lds_bytes));
        h->large_grid = large_grid(h->hp.opt, reinterpret_cast<const void*>(h->large_fn), P.batch > 0 ? P.batch : 1,
            P.large.threads, h->hp.lds_bytes);
        chk(hipMalloc((void**)&h->d_ws, (size_t)h->large_grid * (size_t)P.large.ws_total * sizeof(double)));
    }
    h->packed = (h->hp.large || h->hp.plan.lds.tri || h->hp.opt.no_packed) ? 0 // (the packed bodies are square-layout)
        : packed_width(is ? P.nx + P.n : P.n, P.nx * (P.nx + P.nu + 1), P.rfull > 0, h->hp.lds_bytes);
    chk(hipMalloc((void**)&h->d_ovf_count, 2 * sizeof(int)));
    chk(hipMalloc((void**)&h->d_ovf_list, b * sizeof(int)));
    chk(hipEventCreate(&h->ev0));
    chk(hipEventCreate(&h->ev1));
    chk(hipEventCreate(&h->evm));
    if (e != hipSuccess) {
        g_copra_err = std::string("copra_batch_create: ") + hipGetErrorString(e);
        copra_batch_destroy(h);
        return COPRA_ERR_HIP;
    }
    *out = h;
    return COPRA_OK;
}

void copra_batch_destroy(copra_batch_t* h)
{
    if (!h) return;
    (void)hipFree(h->d_row_step);
    (void)hipFree(h->d_row_ekind);
    (void)hipFree(h->d_row_eoff);
    (void)hipFree(h->d_row_gkind);
    (void)hipFree(h->d_row_goff);
    (void)hipFree(h->d_row_f);
    (void)hipFree(h->d_row_prev);
    (void)hipFree(h->d_warm);
    (void)hipFree(h->d_params);
    (void)hipFree(h->d_lb);
    (void)hipFree(h->d_ub);
    (void)hipFree(h->own_A);
    (void)hipFree(h->own_B);
    (void)hipFree(h->own_d);
    (void)hipFree(h->own_x0);
    (void)hipFree(h->d_results);
    if (h->h_results) (void)hipHostFree(h->h_results);
    (void)hipFree(h->d_prof);
    (void)hipFree(h->d_isR);
    (void)hipFree(h->d_isr);
    (void)hipFree(h->d_x0opt);
    (void)hipFree(h->own_x0lb);
    (void)hipFree(h->own_x0ub);
    if (h->jit_module) (void)hipModuleUnload(h->jit_module);
    for (int k = 0; k < kMaxCosts; ++k) (void)hipFree(h->d_cost_p[k]);
    (void)hipFree(h->d_row_f_inst);
    (void)hipFree(h->d_lb_inst);
    (void)hipFree(h->d_ub_inst);
    (void)hipFree(h->d_ws);
    for (void* q : h->ric_dev) (void)hipFree(q);
    (void)hipFree(h->d_ric_ws);
    (void)hipFree(h->d_ric_next);
    (void)hipFree(h->d_ric_model);
    (void)hipFree(h->d_shA);
    (void)hipFree(h->d_shB);
    (void)hipFree(h->d_shd);
    (void)hipFree(h->d_model);
    (void)hipFree(h->d_ovf_count);
    (void)hipFree(h->d_ovf_list);
    (void)hipFree(h->d_lane_count);
    (void)hipFree(h->d_lane_list);
    (void)hipFree(h->d_lane_hist);
    (void)hipFree(h->d_axis_acc);
    (void)hipFree(h->d_axis_list2);
    (void)hipFree(h->d_axis_count2);
    if (h->h_lane_seen) (void)hipHostFree(h->h_lane_seen);
    (void)hipFree(h->d_lane_ws);
    (void)hipFree(h->d_lane_ws2);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->evm) (void)hipEventDestroy(h->evm);
    delete h;
}





copra_status_t copra_plan_check(const copra_dims_t* dims, int n_costs, const copra_cost_desc_t* costs, int n_cstrs,
    const copra_cstr_desc_t* cstrs, const copra_initial_state_desc_t* is)
{
    if (!dims) return fail(COPRA_ERR_ARG, "copra_plan_check: null dims");
    HostPlan hp; // (the process-wide default options)
    const copra_status_t rc = build_plan(hp, *dims, n_costs, costs, n_cstrs, cstrs, is);
    if (rc != COPRA_OK) g_copra_err = hp.error;
    if (rc == COPRA_OK && hp.ric_only) { // beyond the condensed kernels' sizes: covered if (and only if) the controller is stage-wise
        HostStagePlan hs;
        build_stage_plan(hp, hs, false);
        if (!hs.eligible)
            return fail(COPRA_ERR_UNSUPPORTED, "more than 512 decision variables (or InitialStateLMPC with xDim > 16) need a stage-wise controller for the Riccati interior-point kernel: " + hs.why);
    }
    return rc;
}

// The order of the states in this controller's systems, for the (instance, axis)-per-lane solver (FusedPlan::axis_order): from the zero pattern of
// the FIRST system the controller is given (one instance's A and B: 432 bytes from the device where the caller's arrays live there -- once per
// controller).  Systems in neither order, or that change their order later, are what the solver's per-instance check and the adaptation catch.
extern "C++" void see_axis_order(copra_batch* h, const double* A, const double* B, bool on_device)
{
    const FusedPlan& P = h->hp.plan;
    if (h->axis_order_seen || P.batch <= 0 || (P.axis_tab < 0 && h->hp.axis1_tab < 0)) return;
    h->axis_order_seen = true;
    const size_t nA = (size_t)P.nx * P.nx, nB = (size_t)P.nx * P.nu;
    std::vector<double> a(nA), b(nB);
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost;
    if (hipMemcpy(a.data(), A, nA * sizeof(double), kind) != hipSuccess || hipMemcpy(b.data(), B, nB * sizeof(double), kind) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    const int ord = axis_order_of(a.data(), b.data(), P.nx, P.nu);
    if (ord == 1 && h->hp.axis1_tab >= 0) h->axis_order = 1;
    if (h->hp.opt.debug) fprintf(stderr, "[copra] state order of the first system: %d (0: state i on axis i %% nu, 1: axis-major, -1: neither)\n", ord);
}

// instance 0 of an array of `batch` blocks of `per` doubles, repeated into every other block
__global__ void copra_repeat_first_block_kernel(double* __restrict__ a, int per, long long total)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x + per;
    if (e < total) a[e] = a[e % per];
}

// One model for the batch on a controller the (instance, axis)-per-lane solver takes: that solver, reading every instance's system from memory,
// is faster than the shared-model path at EVERY batch size (profiles/r06/shared_model_against_instance_by_instance.txt: 256 instances 21 vs
// 34 us, 65 536: 0.093 vs 0.191 ms; a goal per instance: 0.107 vs 0.247 ms) -- so the model is written out `batch` times and the controller
// stays on the per-instance path.  Only where that solver would take every solve: its tables and reference coefficients exist, nothing
// has switched it off.
static bool shared_model_runs_as_batch(const copra_batch* h)
{
    const copra_options_t& opt = h->hp.opt;
    FusedPlan P = h->hp.plan;
    if (h->axis_order == 1) P.axis_order = 1, P.axis_tab = h->hp.axis1_tab, P.axis_cref = h->hp.axis1_cref, P.axis_rpa = h->hp.axis1_rpa, P.axis_const = h->hp.axis1_const;
    if (opt.no_axis_solver || opt.no_lane_pass || h->ad.axis_off || P.axis_tab < 0 || P.axis_cref < 0 || P.batch <= 0) return false;
    if (opt.lane_min_batch > 0 && P.batch < opt.lane_min_batch) return false;
    if (h->packed || h->hp.large || P.initial_state || h->jit_fused) return false;
    if (P.stage_refs) {
        int oB = 0, oR = 0, rcs = 0;
        (void)axis_lds_doubles(P.nx, P.nu, P.N, P.axis_rpa, kAxisQmax, oB, oR, rcs);
        if (P.N * (P.nx / P.nu + 1) > rcs) return false;
    }
    return select_axis_kernel(P) != nullptr;
}

copra_status_t copra_batch_set_shared_system(copra_batch_t* h, const double* A, const double* B, const double* d,
    int on_device)
{
    if (!h || !A || !B || !d) return fail(COPRA_ERR_ARG, "copra_batch_set_shared_system: null argument");
    const FusedPlan& P = h->hp.plan;
    if (P.initial_state || h->hp.large)
        return fail(COPRA_ERR_UNSUPPORTED, "the shared-model fast path covers LMPC with at most 64 decision variables");
    const size_t nA = (size_t)P.nx * P.nx, nB = (size_t)P.nx * P.nu, nd = (size_t)P.nx;
    h->shA.resize(nA), h->shB.resize(nB), h->shd.resize(nd);
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost;
    HIP_TRY(hipMemcpy(h->shA.data(), A, nA * sizeof(double), kind));
    HIP_TRY(hipMemcpy(h->shB.data(), B, nB * sizeof(double), kind));
    HIP_TRY(hipMemcpy(h->shd.data(), d, nd * sizeof(double), kind));
    h->shared_as_batch = false;
    see_axis_order(h, h->shA.data(), h->shB.data(), false);
    if (!h->shared && shared_model_runs_as_batch(h)) { // (a handle that is in shared-model mode stays there: its layouts have moved)
        const size_t b = (size_t)P.batch;
        if (!h->own_A) {
            HIP_TRY(hipMalloc((void**)&h->own_A, b * nA * sizeof(double)));
            HIP_TRY(hipMalloc((void**)&h->own_B, b * nB * sizeof(double)));
            HIP_TRY(hipMalloc((void**)&h->own_d, b * nd * sizeof(double)));
        }
        HIP_TRY(hipMemcpy(h->own_A, h->shA.data(), nA * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->own_B, h->shB.data(), nB * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->own_d, h->shd.data(), nd * sizeof(double), hipMemcpyHostToDevice));
        double* const arr[3] = { h->own_A, h->own_B, h->own_d };
        const size_t per[3] = { nA, nB, nd };
        for (int i = 0; i < 3; ++i) {
            const long long total = (long long)(b * per[i]), rest = total - (long long)per[i];
            if (rest > 0) hipLaunchKernelGGL(copra_repeat_first_block_kernel, dim3((unsigned)((rest + 255) / 256)), dim3(256), 0, 0, arr[i], (int)per[i], total);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(0)); // (the solves run on the caller's stream)
        h->A = h->own_A;
        h->B = h->own_B;
        h->d = h->own_d;
        h->shared_as_batch = true;
        return COPRA_OK;
    }
    h->shared = true;
    h->model_dirty = true;
    // (general rows as well: an instance whose rows go through the free response of the preview rebuilds it from its own x0 and the model's A,
    //  lmpc_fused_ric.hpp -- until that was there, the random differential test of the engine's modes, tests/fuzz/fuzz_modes.py, had statuses and
    //  U off on (6, 3) with a dense state row)
    if (h->hp.plan.lds.ric && h->hp.plan.lds.q1regs && !h->jit_ric && (ric_aot_exact(P.nx, P.nu, P.N) || ric_aot_shape(P.nx, P.nu))) { // (the Riccati-factor tier has a shared-model mode of its own: copra_batch_solve;
                                                                        //  only the library's instantiations: a run-time-compiled one has no prepare kernel)
        h->has_lds_ric = true;
        h->lds_ric = h->hp.plan.lds;
    }
    h->ad.shared_ric_off = false; // (a new model: the tier is tried again)
    h->ad.shared_ric_solves = 0;
    h->shared_ric = false;
    LdsLayout lq {};
    if (tri_layout_with_lds_q1(h->hp.plan, h->hp.plan.lds, lq)) { // the shared-model kernels keep Q1 in LDS
        h->hp.plan.lds = lq;
        h->hp.lds_bytes = (size_t)lq.total * sizeof(double);
        h->lds_attr_set = false;
        h->shared_attr_set = false;
    }
    return COPRA_OK;
}

// One-off work of the shared-model path: nx + 1 probe instances (x0 = 0, e_0 .. e_{nx-1}) of the ordinary fused kernel
// give c(x0) = c0 + C1 x0; one more launch stores J = R^-1, G, Phi, xi and the row norms (FusedPlan::model_out).
static copra_status_t prepare_shared_model(copra_batch* h, hipStream_t s)
{
    const FusedPlan& HP = h->hp.plan;
    const int nx = HP.nx, nu = HP.nu, N = HP.N, n = HP.n, X = HP.X;
    // c is affine in (x0, p of the costs with per-instance references): probe 0 = everything zero, then one unit probe
    // per component of x0 and per row of every such cost
    int rtot = 0;
    for (int t = 0; t < kMaxCosts; ++t) {
        h->model_ref_off[t] = -1;
        if (t < HP.ncost && h->cost_p[t]) {
            h->model_ref_off[t] = rtot;
            rtot += HP.cost[t].prows; // (a reference trajectory: one column per row AND step -- the gradient is affine in all of them)
        }
    }
    const int np = 1 + nx + rtot;
    const ModelLayout m = model_layout(nx, nu, N, n, X, h->hp.lds_full.ldj, HP.mgen);
    const size_t need = (size_t)m.C2 + 2 * (size_t)n * (rtot > 0 ? rtot : 1);
    if (!h->d_model || h->model_doubles < need) {
        (void)hipFree(h->d_model);
        h->d_model = nullptr;
        HIP_TRY(hipMalloc((void**)&h->d_model, need * sizeof(double)));
        h->model_doubles = need;
    }
    const size_t nA = (size_t)nx * nx, nB = (size_t)nx * nu;
    std::vector<double> Ap(nA * np), Bp(nB * np), dp((size_t)nx * np), xp((size_t)nx * np, 0.0);
    for (int a = 0; a < np; ++a) {
        std::copy(h->shA.begin(), h->shA.end(), Ap.begin() + (size_t)a * nA);
        std::copy(h->shB.begin(), h->shB.end(), Bp.begin() + (size_t)a * nB);
        std::copy(h->shd.begin(), h->shd.end(), dp.begin() + (size_t)a * nx);
        if (a >= 1 && a <= nx) xp[(size_t)a * nx + (a - 1)] = 1.0;
    }
    std::vector<void*> owned;
    hipError_t e = hipSuccess;
    auto up = [&](const std::vector<double>& src) -> double* {
        double* dst = nullptr;
        hipError_t r = hipMalloc((void**)&dst, (src.empty() ? 1 : src.size()) * sizeof(double));
        if (r == hipSuccess && !src.empty()) r = hipMemcpy(dst, src.data(), src.size() * sizeof(double), hipMemcpyHostToDevice);
        if (r != hipSuccess && e == hipSuccess) e = r;
        owned.push_back(dst);
        return dst;
    };
    auto release = [&]() {
        for (void* q : owned) (void)hipFree(q);
    };
    FusedPlan P = device_plan(h);
    P.A = up(Ap), P.B = up(Bp), P.d = up(dp), P.x0 = up(xp);
    for (int t = 0; t < HP.ncost; ++t) { // probe references of the costs that have per-instance ones
        P.cost_p[t] = nullptr;
        if (h->model_ref_off[t] < 0) continue;
        const int r = HP.cost[t].prows;
        std::vector<double> pp((size_t)np * r, 0.0);
        for (int i = 0; i < r; ++i) pp[(size_t)(1 + nx + h->model_ref_off[t] + i) * r + i] = 1.0;
        P.cost_p[t] = up(pp);
    }
    P.row_f_inst = nullptr; // (c, J and the norms do not depend on right-hand sides or bounds)
    P.lb_inst = P.ub_inst = nullptr;
    double* dQ = up(std::vector<double>((size_t)n * n));
    double* dC = up(std::vector<double>((size_t)n * np));
    if (e != hipSuccess) {
        release();
        return fail(COPRA_ERR_HIP, std::string("shared-model prepare: ") + hipGetErrorString(e));
    }
    P.batch = np;
    P.lds = h->hp.lds_full;
    P.prof = nullptr;
    P.prof_fine = nullptr;
    for (int a = 0; a < np && e == hipSuccess; ++a) { // c(probe a) through the parity hook of the fused kernel
        P.inst_offset = a;
        P.dump_instance = a;
        P.dump_only = 1;
        P.dumpQ = dQ;
        P.dumpc = dC + (size_t)a * n;
        e = lds_opt_in(reinterpret_cast<const void*>(select_fused_kernel(P)), h->hp.lds_full_bytes);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(select_fused_kernel(P), dim3(1), dim3(64), h->hp.lds_full_bytes, s, P);
        e = hipGetLastError();
    }
    P.inst_offset = 0;
    P.dump_instance = 0;
    P.dump_only = 0;
    P.dumpQ = P.dumpc = nullptr;
    P.model_out = h->d_model;
    if (e == hipSuccess) e = lds_opt_in(reinterpret_cast<const void*>(select_fused_kernel(P)), h->hp.lds_full_bytes);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(select_fused_kernel(P), dim3(1), dim3(64), h->hp.lds_full_bytes, s, P);
        e = hipGetLastError();
    }
    std::vector<double> C((size_t)n * np);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipMemcpy(C.data(), dC, C.size() * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
        std::vector<double> lin((size_t)n * np); // c0 | C1 (n x nx) | C2 (n x rtot), column-major
        for (int j = 0; j < n; ++j) lin[(size_t)j] = C[(size_t)j];
        for (int a = 1; a < np; ++a)
            for (int j = 0; j < n; ++j) lin[(size_t)a * n + j] = C[(size_t)a * n + j] - C[(size_t)j];
        e = hipMemcpy(h->d_model + m.c0, lin.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess)
            e = hipMemcpy(h->d_model + m.C1, lin.data() + n, (size_t)n * nx * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess && rtot > 0)
            e = hipMemcpy(h->d_model + m.C2, lin.data() + (size_t)n * (1 + nx), (size_t)n * rtot * sizeof(double), hipMemcpyHostToDevice);
        // the unconstrained minimiser x = -Qinv (c0 + C1 x0 + C2 p) as an affine map of (x0, p): one n x (1 + nx + R)
        // product here instead of an n x n one in every instance
        const int ld = h->hp.lds_full.ldj;
        std::vector<double> Qi((size_t)n * ld), K((size_t)n * np);
        if (e == hipSuccess) e = hipMemcpy(Qi.data(), h->d_model + m.Qinv, Qi.size() * sizeof(double), hipMemcpyDeviceToHost);
        for (int a = 0; a < np; ++a)
            for (int i = 0; i < n; ++i) {
                double acc = 0.0;
                for (int j = 0; j < n; ++j) acc += Qi[(size_t)j * ld + i] * lin[(size_t)a * n + j]; // (same order as the kernel had)
                K[(size_t)a * n + i] = -acc;
            }
        if (e == hipSuccess) e = hipMemcpy(h->d_model + m.xu0, K.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess)
            e = hipMemcpy(h->d_model + m.K1, K.data() + n, (size_t)n * nx * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess && rtot > 0)
            e = hipMemcpy(h->d_model + m.C2 + (size_t)n * rtot, K.data() + (size_t)n * (1 + nx), (size_t)n * rtot * sizeof(double),
                hipMemcpyHostToDevice);
        h->model_rtot = rtot;
    }
    release();
    if (e != hipSuccess) return fail(COPRA_ERR_HIP, std::string("shared-model prepare: ") + hipGetErrorString(e));
    if (h->shared_ric) { // the stage records of the shared system: one run of the Riccati-factor body (FusedPlan::ric_model_out)
        int oBk, oG, oNb;
        const size_t count = (size_t)ric_model_offsets(nx, nu, N, HP.mgen, oBk, oG, oNb);
        if (!h->d_ric_model) HIP_TRY(hipMalloc((void**)&h->d_ric_model, count * sizeof(double)));
        std::vector<void*> own2;
        hipError_t e2 = hipSuccess;
        auto up2 = [&](const std::vector<double>& src) -> double* {
            double* dst = nullptr;
            hipError_t r = hipMalloc((void**)&dst, src.size() * sizeof(double));
            if (r == hipSuccess) r = hipMemcpy(dst, src.data(), src.size() * sizeof(double), hipMemcpyHostToDevice);
            if (r != hipSuccess && e2 == hipSuccess) e2 = r;
            own2.push_back(dst);
            return dst;
        };
        FusedPlan R = device_plan(h);
        R.A = up2(h->shA), R.B = up2(h->shB), R.d = up2(h->shd), R.x0 = up2(std::vector<double>((size_t)nx, 0.0));
        for (int t = 0; t < kMaxCosts; ++t) R.cost_p[t] = nullptr;
        R.row_f_inst = nullptr;
        R.lb_inst = R.ub_inst = nullptr;
        R.batch = 1;
        R.dump_instance = 0;
        R.lds = h->lds_ric;
        R.prof = nullptr;
        R.prof_fine = nullptr;
        R.ric_model_out = h->d_ric_model;
        const size_t lb = (size_t)h->lds_ric.total * sizeof(double);
        if (e2 == hipSuccess) e2 = lds_opt_in(reinterpret_cast<const void*>(select_fused_kernel(R)), lb);
        if (e2 == hipSuccess) {
            hipLaunchKernelGGL(select_fused_kernel(R), dim3(1), dim3(64), lb, s, R);
            e2 = hipGetLastError();
        }
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(s);
        for (void* q : own2) (void)hipFree(q);
        if (e2 != hipSuccess) return fail(COPRA_ERR_HIP, std::string("shared-model prepare (Riccati records): ") + hipGetErrorString(e2));
    }
    h->model_dirty = false;
    return COPRA_OK;
}













// ---- copra_batch_solve: ONE function per launch path (round-4 verdict: it was one 335-line function with eight paths and four adaptive
//      controllers inline).  What the engine has learnt about the controller lives in AdaptState (engine.hpp); the controllers run in
//      learn_from_the_last_solve, before anything of the next solve is chosen. ----

// the adaptive controllers, in the order they depend on each other; all of them read what the LAST solve left behind
static copra_status_t learn_from_the_last_solve(copra_batch* h)
{
    copra_status_t rc = adapt_lane_pass(h);
    if (rc == COPRA_OK) rc = adapt_axis_solver(h);
    if (rc == COPRA_OK) rc = adapt_layout(h);
    if (rc == COPRA_OK) rc = rechoose_layout(h);
    h->ad.solved_once = true;
    h->ad.lane_ran = false;
    h->ad.axis_ran = false;
    return rc;
}
// ... and where this solve will leave its counters, for the next call of the above
static void remember_outputs(copra_batch* h, const FusedPlan& P)
{
    h->ad.last_iter = P.iter;
    h->ad.last_status = P.status;
}

// Shared-model tick: WHICH first tier -- the Riccati-factor tier in shared-model mode, or lmpc_shared.hpp.  May move the plan's layout.
static copra_status_t choose_shared_tier(copra_batch* h)
{
    // which first tier: the Riccati-factor tier in shared-model mode (cold starts, controller-wide references), or lmpc_shared.hpp
    // The tier is taken by SHAPE.  Where the condensed problem is small (two controls, at most 32 variables) and the active-set path
    // long, the dense triangular solves of lmpc_shared.hpp are cheaper than N stages of the recursion per iteration (planar point
    // mass N = 8, 20 iterations per solve: 6.8 vs 13.5 ms, profiles/r04/tier_choice_map_before_switch.txt; at 5 iterations the tier is twice as
    // fast): after the first solve on the tier its iteration counters decide -- one synchronisation in the controller's life.
    if (h->shared_ric && h->ad.shared_ric_solves == 1 && !h->ad.shared_ric_off && h->hp.plan.nu <= 2 && h->hp.plan.n <= 32) {
        const int* d_it = h->ad.last_iter;
        const int* d_st = h->ad.last_status;
        if (d_it && d_st) {
            const size_t nb = (size_t)h->hp.plan.batch;
            std::vector<int> it(nb * 2), st(nb);
            HIP_TRY(hipStreamSynchronize(h->last_stream));
            HIP_TRY(hipMemcpy(it.data(), d_it, it.size() * sizeof(int), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(st.data(), d_st, st.size() * sizeof(int), hipMemcpyDeviceToHost));
            long long sum = 0, cnt = 0;
            for (size_t b = 0; b < nb; ++b)
                if (st[b] == 0) sum += it[2 * b], cnt += 1;
            if (cnt > 0 && sum > 12 * cnt) {
                h->ad.shared_ric_off = true;
                if (h->hp.opt.debug)
                    fprintf(stderr, "[copra] shared-model tick: %.1f iterations per solve at %d variables: lmpc_shared.hpp from now on\n",
                        (double)sum / (double)cnt, h->hp.plan.n);
            }
        }
        h->ad.shared_ric_solves = 2; // (decided)
    }
    bool want = h->has_lds_ric && !h->d_warm && !h->hp.opt.no_ric_shared && !h->ad.shared_ric_off;
    // Per-instance cost references (one model, every instance its own goal / reference trajectory): the records were swept with the
    // controller-wide references, an instance's own feed-forward terms come from the DELTA sweep of the shared lane pass
    // (lmpc_lane_shared_body) -- so the records form takes them where that pass runs; elsewhere lmpc_shared.hpp (reference columns of
    // the shared model), as for every such controller before round 4 (96 vs 200 M solves/s at the headline shape).
    bool refs = false;
    for (int t = 0; t < kMaxCosts; ++t) refs = refs || h->cost_p[t];
    if (want && refs) {
        const FusedPlan Pw = device_plan(h);
        want = !h->ad.lane_off && !h->hp.opt.no_lane_pass && lane_batch_ok(h->hp.opt, Pw.batch, true, true) && Pw.lane_tab >= 0 && Pw.lane_cref >= 0
            && h->lds_ric.ricC && !Pw.prof && !Pw.prof_fine && !Pw.row_f_inst && select_lane_shared_kernel(Pw) != nullptr;
        if (want && ensure_lane_buffers(h, true) != COPRA_OK) { // (no room for the delta terms: lmpc_shared.hpp)
            (void)hipGetLastError();
            want = false;
        }
    }
    if (want != h->shared_ric) {
        LdsLayout lq {};
        if (want) {
            h->hp.plan.lds = h->lds_ric;
        } else if (tri_layout_with_lds_q1(h->hp.plan, h->lds_ric, lq)) {
            h->hp.plan.lds = lq;
        }
        h->hp.lds_bytes = (size_t)h->hp.plan.lds.total * sizeof(double);
        h->hp.two_tier = true;
        h->lds_attr_set = false;
        h->shared_attr_set = false;
        h->shared_ric = want;
        h->model_dirty = true;
    }
    return COPRA_OK;
}

// copra_batch_set_shared_system: one (A, B, d) for the batch -- prepare launch when the model changed, [shared lane pass +] first tier, second tier
static copra_status_t solve_shared_model(copra_batch* h, hipStream_t s)
{
    if (!h->x0) return fail(COPRA_ERR_RUNTIME, "copra_batch_solve: no initial states set (copra_batch_set_x0)");
    h->last_stream = s;
    if (h->hp.plan.batch == 0) return COPRA_OK;
    copra_status_t rcc = choose_shared_tier(h);
    if (rcc != COPRA_OK) return rcc;
    copra_status_t rc = ensure_lds_attr(h);
    if (rc != COPRA_OK) return rc;
    if (!h->shared_attr_set && h->hp.lds_full_bytes > 48 * 1024) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(select_shared_kernel(h->hp.plan, false)),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->hp.lds_full_bytes));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(select_shared_kernel(h->hp.plan, true)),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->hp.lds_full_bytes));
    }
    h->shared_attr_set = true;
    if (h->model_dirty) {
        rc = prepare_shared_model(h, s);
        if (rc != COPRA_OK) return rc;
    }
    FusedPlan P = device_plan(h);
    P.model = h->d_model;
    for (int k = 0; k < kMaxCosts; ++k) P.model_ref_off[k] = h->model_ref_off[k];
    P.model_rtot = h->model_rtot;
    HIP_TRY(hipEventRecord(h->ev0, s));
    if (h->hp.two_tier) HIP_TRY(begin_overflow_queue(h, s, false, P));
    if (h->shared_ric && P.lds.ric) { // first tier: the Riccati-factor body, records copied from the prepare launch instead of swept
        if (h->ad.shared_ric_solves < 1) h->ad.shared_ric_solves = 1;
        FusedPlan Pr = P;
        Pr.ric_model = h->d_ric_model;
        // in front of it the one-instance-per-lane pass in its shared-model form (lmpc_lane.hpp): the roll-out of every instance from
        // the batch-wide records; the tier solves what it leaves over, starting from the U and X it wrote
        unsigned g1 = (unsigned)P.batch;
        bool refs_now = false; // (then the pass MUST run: the choice of this tier above has checked that it can)
        for (int t = 0; t < kMaxCosts; ++t) refs_now = refs_now || h->cost_p[t];
        bool pass_ran = false;
        if (!h->ad.lane_off && !h->hp.opt.no_lane_pass && lane_batch_ok(h->hp.opt, P.batch, true, refs_now) && P.lane_tab >= 0 && P.lds.ricC && !P.prof && !P.prof_fine
            && !P.row_f_inst && select_lane_shared_kernel(P) && (ensure_lane_buffers(h, refs_now) == COPRA_OK || (h->ad.lane_off = true, false))) {
            // (no room for the pass's list: the tier alone, from now on -- as on the per-instance path below)
            h->lane_cur ^= 1;
            h->ad.lane_ran = true;
            Pr.lane_list = h->d_lane_list;
            Pr.lane_count = h->d_lane_count + h->lane_cur;
            Pr.lane_zero = h->d_lane_count + (h->lane_cur ^ 1);
            Pr.lane_bp = (int)(((size_t)P.batch + kWave - 1) / kWave * kWave);
            Pr.lane_ws = refs_now ? h->d_lane_ws : nullptr; // (the delta feed-forward terms of instances with their own references)
            Pr.lane_spec = (h->hp.opt.no_lane_spec || h->ad.lane_spec_shared_off) ? 0 : 1; // (the first steps of the iteration in the pass, lmpc_lane_shared_body: the tier then rolls out again)
            hipLaunchKernelGGL(select_lane_shared_kernel(Pr), dim3((unsigned)(Pr.lane_bp / kWave)), dim3(64), lane_lds_bytes(Pr), s, Pr);
            HIP_TRY(hipGetLastError());
            Pr.lane_from_list = 1;
            Pr.lane_handover = 1;
            Pr.lane_zero = nullptr;
            g1 = ((unsigned)P.batch + 7u) & ~7u; // (the list is dealt out in eighths: ric_tier_instance)
            pass_ran = true;
        }
        if (refs_now && !pass_ran)
            return fail(COPRA_ERR_RUNTIME, "copra_batch_solve: per-instance references on the shared-model records tier need the lane pass in front");
        LDS_OPT_IN(select_fused_kernel(Pr), h->hp.lds_bytes);
        hipLaunchKernelGGL(select_fused_kernel(Pr), dim3(g1), dim3(64), h->hp.lds_bytes, s, Pr);
        HIP_TRY(hipGetLastError());
    } else if (h->jit_shared && h->jit_lanes == (h->packed ? h->packed : 64) && h->jit_tri == P.lds.tri) {
        FusedPlan Pj = P;
        void* args[] = { &Pj };
        const unsigned per = 64u / (unsigned)h->jit_lanes;
        LDS_OPT_IN(h->jit_shared, (size_t)per * h->hp.lds_bytes);
        HIP_TRY(hipModuleLaunchKernel(h->jit_shared, ((unsigned)P.batch + per - 1) / per, 1, 1, 64, 1, 1,
            per * (unsigned)h->hp.lds_bytes, s, args, nullptr));
    } else if (h->packed) {
        HIP_TRY(h->packed == 16 ? packed_launch_w16(P, true, h->hp.lds_bytes, s) : packed_launch_w32(P, true, h->hp.lds_bytes, s));
    } else {
        LDS_OPT_IN(select_shared_kernel(P, false), h->hp.lds_bytes);
        hipLaunchKernelGGL(select_shared_kernel(P, false), dim3((unsigned)P.batch), dim3(64), h->hp.lds_bytes, s, P);
        HIP_TRY(hipGetLastError());
    }
    if (h->hp.two_tier) {
        FusedPlan P2 = P;
        P2.lds = h->hp.lds_full;
        P2.from_list = 1;
        const unsigned g2 = (unsigned)(P.batch < 1024 ? P.batch : 1024);
        LDS_OPT_IN(select_shared_kernel(P2, true), h->hp.lds_full_bytes);
        hipLaunchKernelGGL(select_shared_kernel(P2, true), dim3(g2), dim3(64), h->hp.lds_full_bytes, s, P2);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(h->ev1, s));
    h->timed = true;
    h->tier_timed = false;
    return COPRA_OK;
}

// more than 64 decision variables: the stage-wise interior-point kernel with the workgroup Goldfarb-Idnani kernel behind it, or the latter alone
static copra_status_t solve_large(copra_batch* h, FusedPlan& P, hipStream_t s)
{
    if (use_riccati(h)) {
        // first tier: stage-wise interior-point kernel; second tier: Goldfarb-Idnani for the instances it queued
        const size_t ric_lds = (size_t)(h->ric_fast ? h->hs.sp.fast_lds_doubles : h->hs.sp.lds_doubles) * sizeof(double);
        HIP_TRY(begin_overflow_queue(h, s, false, P));
        HIP_TRY(hipMemsetAsync(h->d_ric_next, 0, sizeof(int), s));
        const riccati_kernel_t ric_fn = h->ric_fast ? copra_lmpc_riccati_mfma_kernel : select_riccati_kernel(P.nx, P.nu);
        LDS_OPT_IN(ric_fn, ric_lds);
        hipLaunchKernelGGL(ric_fn, dim3((unsigned)h->ric_grid), dim3(64), ric_lds, s, P, h->hs.sp);
        HIP_TRY(hipGetLastError());
        if (!h->hp.ric_only) { // (beyond the condensed kernels' sizes the instances that did not converge keep status 3)
            FusedPlan P2 = P;
            P2.from_list = 1;
            LDS_OPT_IN(h->large_fn, h->hp.lds_bytes);
            hipLaunchKernelGGL(h->large_fn, dim3((unsigned)h->large_grid), dim3((unsigned)P.large.threads),
                h->hp.lds_bytes, s, P2);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(h->ev1, s));
        h->timed = true;
        h->tier_timed = false;
        return COPRA_OK;
    }
    LDS_OPT_IN(h->large_fn, h->hp.lds_bytes);
    hipLaunchKernelGGL(h->large_fn, dim3((unsigned)h->large_grid), dim3((unsigned)P.large.threads),
        h->hp.lds_bytes, s, P);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev1, s));
    h->timed = true;
    h->tier_timed = false;
    return COPRA_OK;
}

// InitialStateLMPC up to 64 variables: one wave (or 16 / 32 lanes) per instance
static copra_status_t solve_initial_state(copra_batch* h, FusedPlan& P, hipStream_t s)
{
    if (h->packed) {
        HIP_TRY(h->packed == 16 ? packed_launch_w16(P, false, h->hp.lds_bytes, s) : packed_launch_w32(P, false, h->hp.lds_bytes, s));
    } else {
        LDS_OPT_IN(copra_islmpc_fused_kernel, h->hp.lds_bytes);
        hipLaunchKernelGGL(copra_islmpc_fused_kernel, dim3((unsigned)P.batch), dim3(64), h->hp.lds_bytes, s, P);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(h->ev1, s));
    h->timed = true;
    h->tier_timed = false;
    return COPRA_OK;
}

// The one-wave kernels: [one-instance-per-lane pass ->] first tier [-> second tier].  ext_timed: the launches carry their events in their
// dispatch packets (hipExtLaunchKernelGGL): start of the first launch, end of the first tier (copra_batch_last_first_tier_seconds), end of
// the solve -- no barrier packets in the stream: two hipEventRecord per solve cost ~ 20 us between consecutive solves, 3 % of a headline step.
static copra_status_t solve_one_wave(copra_batch* h, FusedPlan& P, hipStream_t s, bool jit_launch, bool ext_timed)
{
    // the one-instance-per-lane pass (lmpc_lane.hpp): every instance whose unconstrained minimiser violates nothing ends in it, the
    // first tier below runs for the others only
    bool lane_pass = lane_pass_wanted(h, P, jit_launch);
    // ... or, where the controller's axes are decoupled, the one-(instance, axis)-per-lane solver (lmpc_axis.hpp): it finishes every instance
    // whose axes keep their active sets within its lanes' room; the first tier below runs for what it lists
    bool axis_pass = axis_solver_wanted(h, P), axis_quiet = false;
    if (axis_pass && ensure_lane_buffers(h, false) != COPRA_OK) {
        (void)hipGetLastError();
        h->ad.axis_off = true;
        axis_pass = false;
    }
    int axis_spare = 0;
    const int axis_waves = axis_pass ? axis_grid(P.nu, P.batch, axis_spare) : 0;
    if (axis_pass && !h->d_axis_list2) { // the second chance's own list (the first tier's, then) and its length
        hipError_t e = hipMalloc((void**)&h->d_axis_list2, ((size_t)P.batch + 64) * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void**)&h->d_axis_count2, 4 * sizeof(int)); // ([0]: entries; [2]: instances its own steps ended, as d_lane_count)
        if (e == hipSuccess) e = hipMemset(h->d_axis_count2, 0, 4 * sizeof(int));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(h->d_axis_list2);
            (void)hipFree(h->d_axis_count2);
            h->d_axis_list2 = h->d_axis_count2 = nullptr;
            h->ad.axis_off = true;
            axis_pass = false;
        }
    }
    if (axis_pass && !h->d_axis_acc) { // (one word per instance on spare lanes, at most one per nu waves: sized for the batch once)
        const size_t words = (size_t)axis_waves / (size_t)P.nu + 2;
        hipError_t e = hipMalloc((void**)&h->d_axis_acc, words * sizeof(int));
        if (e == hipSuccess) e = hipMemset(h->d_axis_acc, 0, words * sizeof(int));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(h->d_axis_acc);
            h->d_axis_acc = nullptr;
            h->ad.axis_off = true;
            axis_pass = false;
        }
    }
    if (axis_pass) {
        lane_pass = false;
        h->lane_cur ^= 1;
        h->ad.axis_ran = true;
        P.lane_list = h->d_lane_list;
        P.lane_count = h->d_lane_count + h->lane_cur;
        P.lane_zero = h->d_lane_count + (h->lane_cur ^ 1);
        const unsigned ga = (unsigned)axis_waves;
        P.axis_count2 = h->d_axis_count2;
        P.axis_list_in = nullptr;
        P.axis_list_count = nullptr;
        P.axis_waves = axis_waves;
        P.axis_pf = axis_waves > 1024 ? 1024 : 0; // (one wave per SIMD, 1024 SIMDs: the wave 1024 further on is the next on this one's SIMD, give or take)
        P.axis_acc = h->d_axis_acc;
        const fused_kernel_t ak = select_axis_kernel(P);
        LDS_OPT_IN(ak, axis_lds_bytes(P));
        if (h->hp.opt.debug) {
            int per_cu = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(ak), 64, axis_lds_bytes(P));
            fprintf(stderr, "[copra] (instance, axis)-per-lane solver: %u waves, %zu B LDS per wave, %d instances on spare lanes, occupancy API: %d waves per CU\n", ga,
                axis_lds_bytes(P), axis_spare, per_cu);
        }
        const bool quiet = h->hp.two_tier && !jit_launch && !h->packed && h->h_lane_seen && h->lane_seen_solves >= 4 && h->lane_seen_first_max == 0
            && h->lane_seen_max == 0; // (see below)
        if (ext_timed) // (quiet: this IS the first launch and the last but one -- its end is what copra_batch_last_first_tier_seconds reports)
            hipExtLaunchKernelGGL(ak, dim3(ga), dim3(64), axis_lds_bytes(P), s, h->ev0, quiet ? h->evm : nullptr, 0, P);
        else
            hipLaunchKernelGGL(ak, dim3(ga), dim3(64), axis_lds_bytes(P), s, P);
        HIP_TRY(hipGetLastError());
        // A controller whose solver has listed NOTHING in its recent solves (the headline: all 65 536 instances end in it) launches neither the
        // second chance nor the first tier -- three launches that find nothing cost 13 us of a 100 us step --: the tier-2 launch, a grid-stride
        // loop, walks the solver's own list from its first entry.  Should the list fill after all (a workload changes) that launch solves it,
        // slowly, and the next solves see the length that has travelled to the host in the meantime.
        // the second chance of what it listed (active sets beyond its lanes' six constraints): the same solver with room for sixteen, instances
        // from the list -- a few waves that walk it, however long it is -- appending what IT cannot finish to the list the first tier takes
        if (!quiet) {
            FusedPlan Pl = P;
            Pl.axis_list_in = h->d_lane_list;
            Pl.axis_list_count = h->d_lane_count + h->lane_cur;
            Pl.lane_list = h->d_axis_list2;
            Pl.lane_count = h->d_axis_count2;
            Pl.lane_zero = nullptr;
            Pl.lane_hist = nullptr;
            Pl.axis_pf = 0;
            Pl.prof = nullptr;
            const fused_kernel_t lk = select_axis_list_kernel(P);
            const long long want = (4LL * h->lane_seen_first_max) / (64 / P.nu) + 8;
            const unsigned gl = (unsigned)(want < 8 ? 8 : want > 512 ? 512 : want); // (one wave per CU at most: 256 CUs)
            LDS_OPT_IN(lk, axis_list_lds_bytes(P));
            hipLaunchKernelGGL(lk, dim3(gl), dim3(64), axis_list_lds_bytes(P), s, Pl);
            HIP_TRY(hipGetLastError());
        }
        if (!quiet) {
            P.lane_list = h->d_axis_list2;
            P.lane_count = h->d_axis_count2;
        } // (else: the solver's own list and counter, as they are)
        axis_quiet = quiet;
        h->ad.axis_quiet = quiet;
        P.lane_from_list = 1;
        P.lane_handover = 0; // (nothing is handed over: the tier sweeps for itself)
        P.lane_spec = P.lds.ricC ? 1 : 0;
        P.lane_zero = nullptr;
        // The tier's grid: one workgroup per list entry -- and 65 536 workgroups that find none cost 15 us, a sixth of the headline's step.  The
        // length of a solve's list travels to the host behind it (no synchronisation: the next solve takes whatever has arrived); four times the
        // longest of the last lists + 256 workgroups are launched, entry w by workgroup w, and the second launch walks whatever lies beyond
        // (tier-2 kernel: a grid-stride loop -- slow, and only ever busy on the solve in which a workload changes).
        if (h->hp.two_tier && !jit_launch && !h->packed && h->h_lane_seen) {
            long long cap = 4LL * h->lane_seen_max + 256;
            if (h->lane_seen_solves >= 2 && cap < (long long)P.batch) P.lane_cap = (int)((cap + 7) & ~7LL);
            P.lane_rest = quiet ? 0 : P.lane_cap > 0 ? P.lane_cap : -1;
        }
    }
    if (lane_pass && ensure_lane_buffers(h, true) != COPRA_OK) { // (no room for its workspace: the tier alone, from now on)
        (void)hipGetLastError();
        h->ad.lane_off = true;
        lane_pass = false;
    }
    if (lane_pass) {
        h->lane_cur ^= 1;
        h->ad.lane_ran = true;
        P.lane_ws = h->d_lane_ws;
        P.lane_ws2 = h->d_lane_ws2;
        P.lane_list = h->d_lane_list;
        P.lane_count = h->d_lane_count + h->lane_cur;
        P.lane_zero = h->d_lane_count + (h->lane_cur ^ 1);
        P.lane_bp = (int)(((size_t)P.batch + kWave - 1) / kWave * kWave) + kWave;
        // (64 instances per wave.  Half-waves -- twice the waves for a shard of BASELINE configs[3] -- were built in round 4 and measured no
        //  faster, profiles/r04/lane_half_waves.txt: the switch and its code are gone.)
        const unsigned g0 = (unsigned)(((long long)P.batch + kWave - 1) / kWave);
        // Two forms of the pass in front of the compact variant of the Riccati-factor tier (results are the same):
        //   speculation (default): it takes the first steps of the active-set iteration itself where bounds on u_0 are the picks -- at the
        //     headline's constraint level 96 % of the batch ends in it -- and hands NOTHING over: the few instances it leaves sweep for themselves
        //     (the hand-over blocks of 65 536 instances were 157 MB of writes for the 2 437 that read them);
        //   hand-over (copra_options_t::no_lane_spec): Lam^-1 | kv | norm sums of every instance for a tier that takes the factor over.
        P.lane_spec = (P.lds.ricC && !h->hp.opt.no_lane_spec && !h->ad.lane_form_handover) ? 1 : 0; // (adapt_lane_pass chooses between them by workload)
        // (the speculating build has no hand-over code: a code object without the plain build -- an older cache entry -- only filters)
        const bool plain_missing = jit_launch && !h->jit_lane_plain;
        P.lane_handover = (P.lds.ricC && !h->hp.opt.no_lane_handover && !P.lane_spec && !plain_missing) ? 1 : 0;
        // first solve of a controller on a factor-only tier with a layout ladder: the pass also counts, per instance it leaves over, the
        // rows its unconstrained minimiser violates; the layout the tier STARTS on is chosen from that histogram (below)
        const bool predict = h->ad.lane_predict_left > 0 && h->hp.two_tier && P.lds.tri && !h->shared && !h->hp.opt.no_ladder;
        if (predict) {
            P.lane_hist = h->d_lane_hist;
            HIP_TRY(hipMemsetAsync(h->d_lane_hist, 0, kLaneHistBins * sizeof(int), s)); // (the pass ADDS to it: a re-armed prediction -- copra_batch_specialise -- must not count the earlier solve's instances again)
        }
        if (jit_launch) {
            FusedPlan Pl = P;
            void* largs[] = { &Pl };
            const hipFunction_t jl = (P.lane_spec || !h->jit_lane_plain) ? h->jit_lane : h->jit_lane_plain;
            LDS_OPT_IN(jl, lane_lds_bytes(P));
            HIP_TRY(hipModuleLaunchKernel(jl, g0, 1, 1, 64, 1, 1, (unsigned)lane_lds_bytes(P), s, largs, nullptr));
        } else {
            LDS_OPT_IN(select_lane_kernel(P), lane_lds_bytes(P));
            if (ext_timed)
                hipExtLaunchKernelGGL(select_lane_kernel(P), dim3(g0), dim3(64), lane_lds_bytes(P), s, h->ev0, nullptr, 0, P);
            else
                hipLaunchKernelGGL(select_lane_kernel(P), dim3(g0), dim3(64), lane_lds_bytes(P), s, P);
            HIP_TRY(hipGetLastError());
        }
        if (P.lane_hist) {
            // The size of an instance's final active set goes with the number of rows its unconstrained minimiser violates (~ 1.1 x,
            // correlation 0.9 from the headline's to the tightest workload of the tests): step down the ladder until at most one instance
            // in `share` is expected to outgrow the first tier -- BEFORE its first launch, with ONE synchronisation in the controller's
            // life, instead of after each of the first solves (round 3: the first solve of the tight workload took 10.4 ms, the
            // steady state 2.8 ms).  adapt_layout keeps checking the real overflow counts of the first solves behind this.
            h->ad.lane_predict_left -= 1;
            int hist[kLaneHistBins];
            HIP_TRY(hipStreamSynchronize(s));
            HIP_TRY(hipMemcpy(hist, h->d_lane_hist, sizeof hist, hipMemcpyDeviceToHost));
            const long long share = h->hp.plan.lds.ric ? 64 : 8;
            for (;;) {
                long long over = 0;
                for (int b = 0; b < kLaneHistBins; ++b)
                    if (b + (b + 3) / 4 > h->hp.plan.lds.rcap) over += hist[b]; // (a little above the mean: the tail decides)
                LdsLayout roomier {};
                if (over * share <= (long long)P.batch || !next_tri_layout(h->hp.plan, h->hp.plan.lds, roomier)) break;
                h->hp.plan.lds = roomier;
                h->hp.lds_bytes = (size_t)roomier.total * sizeof(double);
                h->lds_attr_set = false;
                if (h->hp.opt.debug)
                    fprintf(stderr, "[copra] %lld of %d instances are expected to outgrow the first tier: starting on %zu B, %d columns\n", over,
                        P.batch, h->hp.lds_bytes, roomier.rcap);
            }
            P.lds = h->hp.plan.lds;
            P.lane_hist = nullptr;
        }
        P.lane_from_list = 1;
        P.lane_zero = nullptr;
    }
    if (h->hp.two_tier) HIP_TRY(begin_overflow_queue(h, s, P.lds.ric && (!jit_launch || h->jit_ric) && !h->packed, P));
    if (jit_launch) {
        FusedPlan Pj = P;
        void* args[] = { &Pj };
        const unsigned per = 64u / (unsigned)h->jit_lanes;
        const hipFunction_t jfn = (h->jit_ric && P.lds.q1regs == 0) ? h->jit_fused_q0 : h->jit_fused;
        LDS_OPT_IN(jfn, (size_t)per * h->hp.lds_bytes);
        const unsigned gj = (lane_pass || axis_pass) ? (((unsigned)P.batch + 7u) & ~7u) : ((unsigned)P.batch + per - 1) / per; // (the list is dealt out in eighths)
        HIP_TRY(hipModuleLaunchKernel(jfn, gj, 1, 1, 64, 1, 1, per * (unsigned)h->hp.lds_bytes, s, args, nullptr));
    } else if (h->packed) {
        HIP_TRY(h->packed == 16 ? packed_launch_w16(P, false, h->hp.lds_bytes, s) : packed_launch_w32(P, false, h->hp.lds_bytes, s));
    } else if (axis_quiet) {
        // (no first tier: see above -- the tier-2 launch below walks the list)
    } else {
        LDS_OPT_IN(select_fused_kernel(P), h->hp.lds_bytes);
        if (h->hp.opt.debug) {
            int per_cu = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(select_fused_kernel(P)), 64, h->hp.lds_bytes);
            fprintf(stderr, "[copra] fused first tier: %zu B LDS per instance, %d columns, q1regs %d, occupancy API: %d instances per CU\n",
                h->hp.lds_bytes, P.lds.rcap, P.lds.q1regs, per_cu);
        }
        const unsigned g1 = P.lane_cap > 0 ? (unsigned)P.lane_cap : (lane_pass || axis_pass) ? (((unsigned)P.batch + 7u) & ~7u) : (unsigned)P.batch; // (the list is dealt out in eighths)
        if (ext_timed) // (start | end of the first launch; with a second launch the solve ends with THAT kernel's packet)
            hipExtLaunchKernelGGL(select_fused_kernel(P), dim3(g1), dim3(64), h->hp.lds_bytes, s,
                (lane_pass || axis_pass) ? nullptr : h->ev0, h->hp.two_tier ? h->evm : h->ev1, 0, P);
        else
            hipLaunchKernelGGL(select_fused_kernel(P), dim3(g1), dim3(64), h->hp.lds_bytes, s, P);
        HIP_TRY(hipGetLastError());
    }
    // the lengths of this solve's lists, for the grids of the next ones: pinned host words nobody waits for -- written by the tier-2 launch itself
    // where there is one (two 4-byte copies behind the solve are two dispatches of the runtime's copy kernel, 4 us each: 8 % of the headline's step)
    bool seen_by_kernel = false;
    if (axis_pass) {
        if (!h->h_lane_seen) {
            if (hipHostMalloc((void**)&h->h_lane_seen, 4 * sizeof(int), hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                h->h_lane_seen = nullptr;
            } else {
                h->h_lane_seen[0] = h->h_lane_seen[1] = h->h_lane_seen[2] = h->h_lane_seen[3] = 0;
                if (hipHostGetDevicePointer((void**)&h->d_lane_seen, h->h_lane_seen, 0) != hipSuccess) {
                    (void)hipGetLastError();
                    h->d_lane_seen = nullptr;
                }
            }
        }
        if (h->h_lane_seen) {
            if (h->lane_seen_solves > 0) { // (what the LAST solve left there)
                const int seen = h->h_lane_seen[h->lane_seen_slot], seen1 = h->h_lane_seen[2 + h->lane_seen_slot];
                h->lane_seen_max = h->lane_seen_solves % 64 == 0 ? seen : (seen > h->lane_seen_max ? seen : h->lane_seen_max); // (the window starts again every 64 solves)
                h->lane_seen_first_max = h->lane_seen_solves % 64 == 0 ? seen1 : (seen1 > h->lane_seen_first_max ? seen1 : h->lane_seen_first_max);
            }
            h->lane_seen_slot ^= 1;
            h->lane_seen_solves += 1;
        }
    }
    if (h->hp.two_tier) {
        // second tier: same kernel, full LDS layout, instances taken from the overflow queue (usually empty)
        FusedPlan P2 = P;
        P2.lds = h->hp.lds_full;
        P2.from_list = 1;
        P2.lane_from_list = 0; // (its body takes the instance it is given)
        if (axis_pass && h->h_lane_seen && h->d_lane_seen && !jit_launch) {
            P2.seen_out = h->d_lane_seen + h->lane_seen_slot;
            P2.seen_src0 = axis_quiet ? h->d_lane_count + h->lane_cur : h->d_axis_count2; // (what the tier got)
            P2.seen_src1 = h->d_lane_count + h->lane_cur; // (what the second chance got)
            seen_by_kernel = true;
        }
        const unsigned g2 = (unsigned)(P.batch < 1024 ? P.batch : 1024);
        LDS_OPT_IN(select_tier2_kernel(P2), h->hp.lds_full_bytes);
        if (ext_timed)
            hipExtLaunchKernelGGL(select_tier2_kernel(P2), dim3(g2), dim3(64), h->hp.lds_full_bytes, s, nullptr, h->ev1, 0, P2);
        else
            hipLaunchKernelGGL(select_tier2_kernel(P2), dim3(g2), dim3(64), h->hp.lds_full_bytes, s, P2);
        HIP_TRY(hipGetLastError());
    }
    if (axis_pass && h->h_lane_seen && !seen_by_kernel) {
        HIP_TRY(hipMemcpyAsync(h->h_lane_seen + h->lane_seen_slot, axis_quiet ? h->d_lane_count + h->lane_cur : h->d_axis_count2, sizeof(int), hipMemcpyDeviceToHost, s)); // (what the tier got)
        HIP_TRY(hipMemcpyAsync(h->h_lane_seen + 2 + h->lane_seen_slot, h->d_lane_count + h->lane_cur, sizeof(int), hipMemcpyDeviceToHost, s)); // (what the second chance got)
    }
    if (!ext_timed) HIP_TRY(hipEventRecord(h->ev1, s));
    h->tier_timed = ext_timed && h->hp.two_tier;
    h->timed = true;
    return COPRA_OK;
}

copra_status_t copra_batch_solve(copra_batch_t* h, void* hip_stream)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_solve: null handle");
    copra_status_t rc = learn_from_the_last_solve(h);
    if (rc != COPRA_OK) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    if (h->shared) {
        rc = solve_shared_model(h, s);
        if (rc == COPRA_OK) remember_outputs(h, device_plan(h));
        return rc;
    }
    if (h->shared_as_batch && !h->x0) return fail(COPRA_ERR_RUNTIME, "copra_batch_solve: no initial states set (copra_batch_set_x0)");
    if (!h->A || !h->B || !h->d || !h->x0)
        return fail(COPRA_ERR_RUNTIME, "copra_batch_solve: no preview system set (copra_batch_set_system)");
    FusedPlan P = device_plan(h);
    h->last_stream = s;
    if (P.batch == 0) return COPRA_OK;
    rc = ensure_lds_attr(h);
    if (rc != COPRA_OK) return rc;
    remember_outputs(h, P);
    const bool jit_launch = h->jit_fused && h->jit_lanes == (h->packed ? h->packed : 64) && h->jit_tri == P.lds.tri
        && h->jit_ric == (P.lds.ric != 0);
    const bool ext_timed = !h->hp.large && !P.initial_state && !jit_launch && !h->packed;
    if (!ext_timed) HIP_TRY(hipEventRecord(h->ev0, s));
    if (h->hp.large) return solve_large(h, P, s);
    if (P.initial_state) return solve_initial_state(h, P, s);
    return solve_one_wave(h, P, s, jit_launch, ext_timed);
}







copra_status_t copra_batch_dump_qp(copra_batch_t* h, int instance, double* Q, double* c, double* Aeq, double* beq,
    double* Aineq, double* bineq, double* lb, double* ub)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_dump_qp: null handle");
    const FusedPlan& HP = h->hp.plan;
    if (instance < 0 || instance >= HP.batch) return fail(COPRA_ERR_ARG, "copra_batch_dump_qp: bad instance");
    if (!h->A) return fail(COPRA_ERR_RUNTIME, "copra_batch_dump_qp: no preview system set");
    if (h->hp.ric_only)
        return fail(COPRA_ERR_UNSUPPORTED, "copra_batch_dump_qp: the condensed QP of a controller this size is never formed on the device (stage-wise Riccati path)");
    const int n = HP.initial_state ? HP.nx + HP.n : HP.n, mg = HP.mgen;
    double *dQ = nullptr, *dc = nullptr, *dA = nullptr, *db = nullptr;
    HIP_TRY(hipMalloc((void**)&dQ, (size_t)n * n * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&dc, (size_t)n * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&dA, (size_t)(mg ? mg : 1) * n * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&db, (size_t)(mg ? mg : 1) * sizeof(double)));
    FusedPlan P = device_plan(h);
    P.inst_offset = instance;
    P.dump_instance = instance;
    P.dump_only = 1;
    P.dumpQ = dQ;
    P.dumpc = dc;
    P.dumpA = dA;
    P.dumpb = db;
    P.lds = h->hp.lds_full;
    copra_status_t rc = ensure_lds_attr(h);
    if (rc != COPRA_OK) return rc;
    if (h->hp.large) {
        LDS_OPT_IN(h->large_fn, h->hp.lds_bytes);
        hipLaunchKernelGGL(h->large_fn, dim3(1), dim3((unsigned)P.large.threads), h->hp.lds_bytes,
            h->last_stream, P);
    } else if (P.initial_state) {
        LDS_OPT_IN(copra_islmpc_fused_kernel, h->hp.lds_full_bytes);
        hipLaunchKernelGGL(copra_islmpc_fused_kernel, dim3(1), dim3(64), h->hp.lds_full_bytes, h->last_stream, P);
    } else { // (P.lds is the full layout here: NOT the symbol the solve launches when that runs the factor-only tier)
        LDS_OPT_IN(select_fused_kernel(P), h->hp.lds_full_bytes);
        hipLaunchKernelGGL(select_fused_kernel(P), dim3(1), dim3(64), h->hp.lds_full_bytes, h->last_stream, P);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->last_stream));
    std::vector<double> hA((size_t)(mg ? mg : 1) * n), hb((size_t)(mg ? mg : 1));
    if (Q) HIP_TRY(hipMemcpy(Q, dQ, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost));
    if (c) HIP_TRY(hipMemcpy(c, dc, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hA.data(), dA, hA.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hb.data(), db, hb.size() * sizeof(double), hipMemcpyDeviceToHost));
    // split the stacked rows (leading dimension mgen) into Aeq (ld meq) and Aineq (ld mineq)
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < HP.meq; ++i)
            if (Aeq) Aeq[(size_t)j * HP.meq + i] = hA[(size_t)j * mg + i];
        for (int i = 0; i < HP.mineq; ++i)
            if (Aineq) Aineq[(size_t)j * HP.mineq + i] = hA[(size_t)j * mg + HP.meq + i];
    }
    for (int i = 0; i < HP.meq; ++i)
        if (beq) beq[i] = hb[(size_t)i];
    for (int i = 0; i < HP.mineq; ++i)
        if (bineq) bineq[i] = hb[(size_t)HP.meq + i];
    { // bounds of the decision vector: [x0lb; lb], [x0ub; ub] for the InitialStateLMPC variant
        const int off = HP.initial_state ? HP.nx : 0;
        std::vector<double> l0((size_t)(off ? off : 1)), u0((size_t)(off ? off : 1));
        if (off) {
            HIP_TRY(hipMemcpy(l0.data(), (h->x0lb ? h->x0lb : h->x0) + (size_t)instance * off, off * sizeof(double), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(u0.data(), (h->x0ub ? h->x0ub : h->x0) + (size_t)instance * off, off * sizeof(double), hipMemcpyDeviceToHost));
        }
        for (int i = 0; i < n; ++i) {
            if (lb) lb[i] = (i < off) ? l0[(size_t)i] : h->hp.lb[(size_t)(i - off)];
            if (ub) ub[i] = (i < off) ? u0[(size_t)i] : h->hp.ub[(size_t)(i - off)];
        }
    }
    (void)hipFree(dQ);
    (void)hipFree(dc);
    (void)hipFree(dA);
    (void)hipFree(db);
    return COPRA_OK;
}







} // extern "C"
