// copra_hip.hip -- kernels + C ABI (include/copra_hip.h) of the MI355X-native batched linear-MPC engine.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared copra_hip.hip -o libcopra_hip.so  (see Makefile)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "../../include/copra_hip.h"
#include "islmpc_fused.hpp"
#include "lmpc_fused.hpp"
#include "lmpc_fused_ric.hpp"
#include "lmpc_lane.hpp"
#include "lmpc_large.hpp"
#include "lmpc_riccati.hpp"
#include "lmpc_riccati_mfma.hpp"
#include "lmpc_shared.hpp"
#include "packed_launch.hpp"
#include "plan_builder.hpp"
#include "qp_dense.hpp"
#include "qp_dense_large.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

extern char** environ;

#include <algorithm>
#include <map>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace copra_hip;

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
// run-time-horizon builds (NH == 0) for the shapes of ric_aot_shape: instantiated in copra_hip_ric.hip, a translation unit of its own
#define COPRA_RIC_RT_DECL(NX, NU)                                                                                      \
    extern template __global__ void copra_lmpc_fused_ric_kernel<NX, NU, 0, kFusedQ1Regs, false>(const FusedPlan);      \
    extern template __global__ void copra_lmpc_fused_ric_kernel<NX, NU, 0, 0, false>(const FusedPlan);                 \
    extern template __global__ void copra_lmpc_fused_ric_kernel<NX, NU, 0, kFusedQ1Regs, true>(const FusedPlan);       \
    extern template __global__ void copra_lmpc_fused_ric_kernel<NX, NU, 0, 0, true>(const FusedPlan);
COPRA_RIC_RT_DECL(6, 3)
COPRA_RIC_RT_DECL(4, 2)
COPRA_RIC_RT_DECL(2, 1)
extern template __global__ void copra_lmpc_lane_kernel<4, 2, false>(const FusedPlan);
extern template __global__ void copra_lmpc_lane_kernel<4, 2, true>(const FusedPlan);
// ... and its shared-model form: the stage records are those of the whole batch (wave-uniform: scalar operands), only the roll-out
// from each instance's x0 is left
template <int NX, int NU>
__global__ __launch_bounds__(64, 4) void copra_lmpc_lane_shared_kernel(const FusedPlan P)
{
    lmpc_lane_shared_body<NX, NU>(P, (int)blockIdx.x);
}
// Second tier of the two-tier scheme (own symbol so that profiles keep the two apart): the same body with the full LDS
// layout, run only for the instances whose active set outgrew the compact layout's R (queue filled by the first tier).
template <int NX, int NU, int NH, int RP>
__global__ __launch_bounds__(64) void copra_lmpc_fused_tier2_kernel(const FusedPlan P)
{
    const int count = *P.ovf_count;
    for (int k = (int)blockIdx.x; k < count; k += (int)gridDim.x) {
        lmpc_fused_body<NX, NU, NH, RP>(P, P.ovf_list[k]);
        __syncthreads();
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

// PreviewSystem::updateSystem (src/PreviewSystem.cpp:57-74) as matrices, for host-evaluated user subclasses of Constraint /
// CostFunction (copra_preview_update).  One workgroup runs the recursion Phi_i = A Phi_{i-1}, G_i = A G_{i-1} (G_0 = B),
// xi_i = A xi_{i-1} + d; a second launch spreads the first block column over Psi_{i,j} = G_{i-1-j}.
__global__ __launch_bounds__(256) void copra_preview_recursion_kernel(int nx, int nu, int N, const double* A, const double* B,
    const double* d, double* Phi, double* G, double* xi)
{
    const int X = nx * (N + 1), tid = (int)threadIdx.x, T = (int)blockDim.x;
    for (int e = tid; e < nx * nx; e += T) Phi[(e % nx) + (size_t)X * (e / nx)] = (e % nx == e / nx) ? 1.0 : 0.0; // Phi_0 = I (:51)
    for (int e = tid; e < nx * nu; e += T) G[e] = B[e]; // Psi_{1,0} = B (:60)
    for (int e = tid; e < nx; e += T) xi[e] = 0.0;
    __syncthreads();
    for (int i = 1; i <= N; ++i) {
        for (int e = tid; e < nx * (nx + nu + 1); e += T) {
            const int c = e / nx, r = e - c * nx;
            double acc = 0.0;
            if (c < nx) { // Phi_i = A Phi_{i-1} (:59, :64)
                for (int t = 0; t < nx; ++t) acc += A[r + nx * t] * Phi[((i - 1) * nx + t) + (size_t)X * c];
                Phi[(i * nx + r) + (size_t)X * c] = acc;
            } else if (c < nx + nu) { // G_i = A G_{i-1} (:65); G_N is not part of Psi
                if (i < N) {
                    const int cc = c - nx;
                    for (int t = 0; t < nx; ++t) acc += A[r + nx * t] * G[(size_t)(i - 1) * nx * nu + t + nx * cc];
                    G[(size_t)i * nx * nu + r + nx * cc] = acc;
                }
            } else { // xi_i = A xi_{i-1} + d (:61, :70)
                acc = d[r];
                for (int t = 0; t < nx; ++t) acc += A[r + nx * t] * xi[(i - 1) * nx + t];
                xi[i * nx + r] = acc;
            }
        }
        __syncthreads();
    }
}
__global__ void copra_preview_fill_kernel(int nx, int nu, int N, const double* G, double* Psi)
{
    const size_t X = (size_t)nx * (N + 1), U = (size_t)nu * N;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= X * U) return;
    const size_t col = e / X, row = e - col * X;
    const int i = (int)(row / nx), r = (int)(row - (size_t)i * nx), j = (int)(col / nu), c = (int)(col - (size_t)j * nu);
    Psi[e] = (j < i) ? G[(size_t)(i - 1 - j) * nx * nu + r + nx * c] : 0.0; // Psi_{i,j} = A^(i-1-j) B (:66-69), row block 0 is zero
}

// out[b][i] = out[0][i], b >= 1: one reference for every instance (copra_batch_set_cost_reference_all)
__global__ void copra_broadcast_reference_kernel(const double* p, double* out, int rows, long long total)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < total) out[e] = p[e % rows];
}

// out[b][row0 + s * r + i] = f[b][i] for the steps s of one constraint (copra_batch_set_constraint_rhs)
__global__ void copra_scatter_rhs_kernel(const double* f, double* out, int batch, int r, int steps, int row0, int mgen)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)r * steps;
    if (e >= per * batch) return;
    const long long b = e / per, rem = e - b * per;
    out[b * mgen + row0 + rem] = f[b * r + rem % r];
}

namespace {
using fused_kernel_t = void (*)(const FusedPlan);
fused_kernel_t select_shared_kernel(const FusedPlan& P, bool tier2)
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
        if (P.rfull > 0 && P.nx == 6 && P.nu == 3 && P.N == 20) return copra_lmpc_fused_tri_kernel<6, 3, 20, 0>; // headline shape, full-size costs
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
} // namespace

__global__ __launch_bounds__(64) void copra_islmpc_fused_kernel(const FusedPlan P)
{
    islmpc_fused_body(P, P.inst_offset + (int)blockIdx.x);
}

__global__ __launch_bounds__(64) void copra_qp_dense_kernel(const DensePlan P) { qp_dense_body(P, (int)blockIdx.x); }

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

// n > 64: one problem per workgroup (thread = row of J), persistent grid over the batch
__global__ __launch_bounds__(kLargeMaxN) void copra_qp_dense_large_kernel(const DensePlan P) { qp_dense_large_body(P); }
// more than four waves per workgroup: 128-VGPR build so that two workgroups share a CU (see copra_lmpc_large_kernel_w4)
__global__ __launch_bounds__(kLargeMaxN, 4) void copra_qp_dense_large_kernel_w4(const DensePlan P) { qp_dense_large_body(P); }

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {

thread_local std::string g_err;

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
    if (opt.large_per_cu > 0) per_cu = opt.large_per_cu; // (tuning aid)
    if (opt.debug)
        fprintf(stderr, "[copra] large grid: %d CUs x %d workgroups of %d threads, %zu B LDS\n", cus, per_cu, threads, lds_bytes);
    long long g = (long long)cus * per_cu;
    if (opt.large_grid > 0) g = opt.large_grid; // (tuning aid)
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
    if (opt.large_no_w4) return false; // (tuning aid)
    return large_per_cu(w4, threads, lds_bytes) > large_per_cu(full, threads, lds_bytes);
}

typedef void (*large_kernel_t)(const FusedPlan);
large_kernel_t choose_large_kernel(const HostPlan& hp)
{
    if (prefer_w4(hp.opt, reinterpret_cast<const void*>(copra_lmpc_large_kernel), reinterpret_cast<const void*>(copra_lmpc_large_kernel_w4),
            hp.plan.large.threads, hp.lds_bytes))
        return copra_lmpc_large_kernel_w4;
    return copra_lmpc_large_kernel;
}

copra_status_t fail(copra_status_t code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                                 \
    do {                                                                                                              \
        hipError_t e_ = (expr);                                                                                       \
        if (e_ != hipSuccess) return fail(COPRA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

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
#define LDS_OPT_IN(fn, bytes) HIP_TRY(lds_opt_in(reinterpret_cast<const void*>(fn), (bytes)))

template <class T>
hipError_t upload(T** dst, const std::vector<T>& src)
{
    const size_t bytes = (src.empty() ? 1 : src.size()) * sizeof(T);
    hipError_t e = hipMalloc((void**)dst, bytes);
    if (e != hipSuccess) return e;
    if (!src.empty()) e = hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

} // namespace

constexpr size_t kSmallSlab = 1u << 20; // result slabs up to this size are fetched with one copy through pinned memory

struct copra_batch {
    HostPlan hp;
    // device copies of the plan tables
    int *d_row_step = nullptr, *d_row_ekind = nullptr, *d_row_eoff = nullptr, *d_row_gkind = nullptr,
        *d_row_goff = nullptr;
    double *d_row_f = nullptr, *d_params = nullptr, *d_lb = nullptr, *d_ub = nullptr;
    int *d_row_prev = nullptr, *d_warm = nullptr; // warm start of the shared-model path (copra_batch_set_warm_start)
    // system (owned copies, or borrowed device pointers)
    double *own_A = nullptr, *own_B = nullptr, *own_d = nullptr, *own_x0 = nullptr;
    const double *A = nullptr, *B = nullptr, *d = nullptr, *x0 = nullptr;
    // results
    double *d_control = nullptr, *d_traj = nullptr; // (carved from ONE allocation, d_results: small batches fetch it with one copy)
    int *d_status = nullptr, *d_iter = nullptr;
    unsigned char* d_results = nullptr;
    size_t results_bytes = 0, off_traj = 0, off_status = 0, off_iter = 0;
    unsigned char* h_results = nullptr; // pinned staging copy of the slab (batches whose slab is at most kSmallSlab bytes)
    // caller-provided device result buffers (copra_batch_set_outputs); override the engine-owned ones
    double *ext_control = nullptr, *ext_traj = nullptr;
    int *ext_status = nullptr, *ext_iter = nullptr;
    int *d_ovf_count = nullptr, *d_ovf_list = nullptr; // two-tier queue (TWO counters, used in turn: begin_overflow_queue)
    int ovf_cur = 0; // the counter the last solve appended to
    bool ovf_clean[2] = { false, false }; // known to hold zero on the device
    // shared-model fast path: one (A, B, d) for the whole batch, factorised once (copra_batch_set_shared_system)
    bool shared = false, model_dirty = true, shared_attr_set = false;
    // shared-model mode of the Riccati-factor tier (lmpc_fused_ric.hpp, FusedPlan::ric_model): the layout the plan builder chose
    // for that tier (kept when copra_batch_set_shared_system moves the plan to an LDS-Q1 layout), whether the next solve uses it,
    // and the batch-wide records
    bool has_lds_ric = false, shared_ric = false;
    LdsLayout lds_ric {};
    double* d_ric_model = nullptr;
    int model_ref_off[kMaxCosts]; // columns of C2 per cost as prepared (-1: none)
    size_t model_doubles = 0; // allocated size of d_model
    int model_rtot = 0; // columns of C2 / K2 as prepared
    double *d_shA = nullptr, *d_shB = nullptr, *d_shd = nullptr, *d_model = nullptr;
    std::vector<double> shA, shB, shd;
    double *d_row_f_inst = nullptr, *d_lb_inst = nullptr, *d_ub_inst = nullptr; // per-instance rhs / control bounds
    double* d_cost_p[kMaxCosts] = {}; // per-instance cost references (owned copies) ...
    const double* cost_p[kMaxCosts] = {}; // ... or borrowed device pointers (copra_batch_set_cost_reference)
    // copra_batch_specialise: this controller's shape compiled into its own kernels (hipcc --genco, cached on disk)
    hipModule_t jit_module = nullptr;
    hipFunction_t jit_fused = nullptr, jit_shared = nullptr;
    hipFunction_t jit_fused_q0 = nullptr; // Riccati-factor tier compiled for this shape: Q1 in LDS (further down the layout ladder)
    hipFunction_t jit_lane = nullptr; // ... and the one-instance-per-lane pass in front of it (lmpc_lane.hpp)
    bool jit_ric = false; // the code object holds the Riccati-factor tier (lmpc_fused_ric.hpp) of this controller's shape
    int jit_lanes = 64; // lanes per instance the code object was compiled for
    int jit_tri = 0; // ... and whether for the factor-only layout
    // one-instance-per-lane pass in front of the Riccati-factor tier (lmpc_lane.hpp)
    int *d_lane_count = nullptr, *d_lane_list = nullptr; // (two counters, used in turn like the overflow queue's)
    int* d_lane_hist = nullptr; // histogram of the violated-row counts the pass leaves (kLaneHistBins; read once, before the first tier launch)
    int lane_predict_left = 1; // solves whose first-tier layout is still chosen from that histogram
    double* d_lane_ws = nullptr;
    int lane_cur = 0; // the counter the last solve appended to
    bool lane_ran = false; // the last solve ran the pass
    bool lane_off = false; // switched off for this controller: too few instances end in it (adapt_lane_pass)
    int lane_adapt_left = 2;
    long long lane_solves = 0; // solves seen by adapt_lane_pass (it samples the share again every 256)
    bool lane_off_by_share = false; // lane_off was set by adapt_lane_pass (too few instances ended in the pass), not for lack of memory
    int adapt_left = 3; // solves after which the overflow count of a compact layout is still checked
    bool solved_once = false;
    int packed = 0; // lanes per instance when several small problems share a wavefront (16 / 32; 0: one wave each)
    void (*large_fn)(const FusedPlan) = nullptr; // workgroup-per-instance kernel variant chosen at creation
    double* d_ws = nullptr; // workgroup-per-instance kernel: [large_grid][ws_total] doubles (J, factor, Phi, ...)
    int large_grid = 0;
    // InitialStateLMPC variant
    double *d_isR = nullptr, *d_isr = nullptr, *d_x0opt = nullptr, *own_x0lb = nullptr, *own_x0ub = nullptr;
    const double *x0lb = nullptr, *x0ub = nullptr;
    // stage-wise Riccati interior-point path (copra_batch_select_solver; lmpc_riccati.hpp)
    int solver = COPRA_SOLVER_DEFAULT;
    HostStagePlan hs;
    bool ric_fast = false; // the LDS-resident kernel (lmpc_riccati_mfma.hpp) runs it
    bool ric_refs = false; // ... decided for this state of the per-instance cost references
    bool ric_built = false; // hs describes this controller (eligible or not) ...
    bool ric_all_bounds = false; // ... with bound rows for every control
    std::vector<void*> ric_dev; // device copies of its tables
    double* d_ric_ws = nullptr;
    int* d_ric_next = nullptr; // work-queue counter of the Riccati kernel
    int ric_grid = 0;
    long long* d_prof_fine = nullptr; // profiling builds only
    long long* d_prof = nullptr; // optional per-instance phase cycle counts (copra_batch_enable_phase_profile)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evm = nullptr; // start | end of the solve | end of its first launch (packet-borne timing)
    bool tier_timed = false; // evm was written by the last solve
    hipStream_t last_stream = nullptr;
    bool timed = false;
    bool lds_attr_set = false;
};

static FusedPlan device_plan(const copra_batch* h)
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
    P.lane_handover = 0;
    P.lane_dbg = 0;
    P.lane_ws = nullptr;
    P.lane_list = nullptr;
    P.lane_count = P.lane_zero = nullptr;
    P.lane_hist = nullptr;
    P.lane_bp = 0;
    P.lane_group = 0;
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
    if (P.nx == 6 && P.nu == 3) return P.stage_refs ? copra_lmpc_lane_kernel<6, 3, true> : copra_lmpc_lane_kernel<6, 3>; // (reference trajectories)
    if (P.nx == 2 && P.nu == 1) return P.stage_refs ? copra_lmpc_lane_kernel<2, 1, true> : copra_lmpc_lane_kernel<2, 1>; // (the reference's falling-mass system: BASELINE configs[1])
    if (P.nx == 4 && P.nu == 2) return P.stage_refs ? copra_lmpc_lane_kernel<4, 2, true> : copra_lmpc_lane_kernel<4, 2>; // (a planar point mass; copra_hip_ric.hip)
    return nullptr;
}
static fused_kernel_t select_lane_shared_kernel(const FusedPlan& P)
{
    if (P.nx == 6 && P.nu == 3) return copra_lmpc_lane_shared_kernel<6, 3>;
    return nullptr;
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
static bool lane_batch_ok(const copra_options_t& opt, int batch, bool ric_tier)
{
    int least = ric_tier ? 20480 : 4096;
    if (opt.lane_min_batch != 0) least = opt.lane_min_batch < 0 ? 0 : opt.lane_min_batch;
    return batch >= least;
}
static bool lane_pass_wanted(const copra_batch* h, const FusedPlan& P, bool jit_launch)
{
    const copra_options_t& opt = h->hp.opt;
    if (h->lane_off || opt.no_lane_pass || !lane_batch_ok(opt, P.batch, P.lds.ric != 0)) return false;
    if ((P.prof && !(opt.lane_dbg & 8)) || P.prof_fine) return false;
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
        e = hipMalloc((void**)&h->d_lane_count, 2 * sizeof(int));
        if (e == hipSuccess) e = hipMemset(h->d_lane_count, 0, 2 * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void**)&h->d_lane_list, bp * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void**)&h->d_lane_hist, kLaneHistBins * sizeof(int));
        if (e == hipSuccess) e = hipMemset(h->d_lane_hist, 0, kLaneHistBins * sizeof(int));
        h->lane_cur = 1;
    }
    if (e == hipSuccess && need_ws && !h->d_lane_ws) // (the shared-model form of the pass has no sweep: no workspace)
        e = hipMalloc((void**)&h->d_lane_ws, (size_t)P.N * lane_ws_rows(P.nx, P.nu) * bp * sizeof(double));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(h->d_lane_count);
        (void)hipFree(h->d_lane_list);
        (void)hipFree(h->d_lane_hist);
        (void)hipFree(h->d_lane_ws);
        h->d_lane_count = h->d_lane_list = h->d_lane_hist = nullptr;
        h->d_lane_ws = nullptr;
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
    h->lane_solves += 1;
    if (h->lane_solves % kLaneResample == 0 && h->lane_adapt_left <= 0) {
        h->lane_adapt_left = 1;
        if (h->lane_off && h->lane_off_by_share) h->lane_off = h->lane_off_by_share = false;
    }
    if (!h->lane_ran || h->lane_adapt_left <= 0) return COPRA_OK;
    // (when the first tier takes the stage records over from the pass -- compact variant of the tier -- the pass pays for every instance:
    //  it is the tier's sweep, done at several times the efficiency; nothing to decide)
    //  -- except in shared-model mode, where there is no sweep to take over: there the pass must finish one instance in four to pay, measured)
    if (!h->shared && h->hp.plan.lds.ricC && !h->hp.opt.no_lane_handover) return COPRA_OK;
    h->lane_adapt_left -= 1;
    int left = 0;
    HIP_TRY(hipStreamSynchronize(h->last_stream));
    HIP_TRY(hipMemcpy(&left, h->d_lane_count + h->lane_cur, sizeof(int), hipMemcpyDeviceToHost));
    const long long done = (long long)h->hp.plan.batch - left;
    long long share = h->shared ? 4 : 8;
    if (h->hp.opt.lane_share > 0) share = h->hp.opt.lane_share; // (experiments)
    if (done * share < (long long)h->hp.plan.batch && !h->hp.opt.lane_keep) h->lane_off = h->lane_off_by_share = true;
    if (h->hp.opt.debug)
        fprintf(stderr, "[copra] one-instance-per-lane pass: %lld of %d instances ended in it%s\n", done, h->hp.plan.batch,
            h->lane_off ? " -- switched off" : "");
    return COPRA_OK;
}

// Compact LDS layouts (R capped, overflow finished by the second tier) bet on small active sets.  After each of the
// first solves the overflow queue tells whether the bet holds; if more than one instance in eight had to be redone by
// the second tier, step to the next safer layout: dense -> safe (quarter-CU compact or full) -> full.
static copra_status_t adapt_layout(copra_batch* h)
{
    if (!h->hp.two_tier || h->hp.large || h->adapt_left <= 0 || !h->solved_once) return COPRA_OK;
    h->adapt_left -= 1;
    int count = 0;
    HIP_TRY(hipStreamSynchronize(h->last_stream));
    HIP_TRY(hipMemcpy(&count, h->d_ovf_count + h->ovf_cur, sizeof(int), hipMemcpyDeviceToHost));
    // (the Riccati-factor tier is so much faster than its second tier -- the square-layout kernel -- that it pays to step down
    //  the ladder until only one instance in 64 is left over -- measured in round 3 over five constraint levels: 64 and 128 equal, 32
    //  leaves a mid-constrained workload on five columns at 42.8 instead of 50.4 M solves/s, 512 goes too far; the other first tiers
    //  keep the round-1 threshold of one in 8)
    long long share = h->hp.plan.lds.ric ? 64 : 8;
    if (h->hp.opt.overflow_share > 0) share = h->hp.opt.overflow_share; // (experiments)
    if ((long long)count * share <= (long long)h->hp.plan.batch) return COPRA_OK;
    LdsLayout roomier {};
    if (next_tri_layout(h->hp.plan, h->hp.plan.lds, roomier, h->hp.opt.no_ladder != 0)) { // factor-only: one instance per CU fewer, more columns
        h->hp.plan.lds = roomier;
        h->hp.lds_bytes = (size_t)roomier.total * sizeof(double);
        h->lds_attr_set = false;
        h->adapt_left += 1; // (a step down the ladder does not use up the budget of attempts)
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

static copra_status_t ensure_lds_attr(copra_batch* h)
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
static copra_status_t prepare_riccati(copra_batch* h)
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
    if (h->hp.opt.riccati_per_cu > 0) per_cu = h->hp.opt.riccati_per_cu; // (tuning aid)
    long long g = (long long)cus * per_cu;
    const int batch = h->hp.plan.batch > 0 ? h->hp.plan.batch : 1;
    h->ric_grid = (int)(g < batch ? g : batch);
    if (e == hipSuccess && !h->ric_fast) // (the LDS-resident kernel has no workspace in HBM)
        e = hipMalloc((void**)&h->d_ric_ws, (size_t)h->ric_grid * (size_t)sp.ws_total * sizeof(double));
    sp.ws = h->d_ric_ws;
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
static bool use_riccati(copra_batch* h)
{
    if (h->hp.ric_only) return prepare_riccati(h) == COPRA_OK && h->hs.eligible;
    if (h->solver == COPRA_SOLVER_QUADPROG_DENSE || h->shared) return false;
    if (h->solver == COPRA_SOLVER_DEFAULT && (!h->hp.large || h->hp.opt.no_riccati)) return false;
    if (prepare_riccati(h) != COPRA_OK) return false;
    return h->hs.eligible;
}

extern "C" {

int copra_abi_version(void) { return 5; } // 5: + copra_options_t, copra_options_init, copra_set_default_options, copra_batch_create_with_options; 3: + copra_batch_last_first_tier_seconds, copra_batch_set_system_rowmajor_async; 4: + copra_batch_lane_pass_info, copra_batch_set_cost_reference_all

copra_status_t copra_preview_update(int nx, int nu, int N, const double* A, const double* B, const double* d, double* Phi,
    double* Psi, double* xi)
{
    if (nx <= 0 || nu <= 0 || N <= 0) return fail(COPRA_ERR_DOMAIN, "copra_preview_update: dimensions and number of steps must be positive");
    if (!A || !B || !d || !Phi || !Psi || !xi) return fail(COPRA_ERR_ARG, "copra_preview_update: null argument");
    const size_t X = (size_t)nx * (N + 1), U = (size_t)nu * N;
    double *dA = nullptr, *dB = nullptr, *dd = nullptr, *dPhi = nullptr, *dPsi = nullptr, *dxi = nullptr, *dG = nullptr;
    hipError_t e = hipMalloc((void**)&dA, (size_t)nx * nx * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&dB, (size_t)nx * nu * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&dd, (size_t)nx * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&dPhi, X * nx * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&dPsi, X * U * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&dxi, X * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&dG, (size_t)N * nx * nu * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(dA, A, (size_t)nx * nx * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dB, B, (size_t)nx * nu * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dd, d, (size_t)nx * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(copra_preview_recursion_kernel, dim3(1), dim3(256), 0, nullptr, nx, nu, N, dA, dB, dd, dPhi, dG, dxi);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        const size_t total = X * U;
        hipLaunchKernelGGL(copra_preview_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nullptr, nx, nu, N, dG, dPsi);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(Phi, dPhi, X * nx * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(Psi, dPsi, X * U * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(xi, dxi, X * sizeof(double), hipMemcpyDeviceToHost);
    for (double* q : { dA, dB, dd, dPhi, dPsi, dxi, dG }) (void)hipFree(q);
    if (e != hipSuccess) return fail(COPRA_ERR_HIP, std::string("copra_preview_update: ") + hipGetErrorString(e));
    return COPRA_OK;
}

copra_status_t copra_batch_set_warm_start(copra_batch_t* h, int enable)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_set_warm_start: null handle");
    if (h->hp.plan.initial_state || h->hp.large)
        return fail(COPRA_ERR_UNSUPPORTED, "the warm start belongs to the shared-model path (LMPC, at most 64 decision variables)");
    if (!enable) {
        (void)hipFree(h->d_warm);
        h->d_warm = nullptr;
        return COPRA_OK;
    }
    const size_t count = (size_t)(h->hp.plan.batch > 0 ? h->hp.plan.batch : 1) * kWarmCap;
    if (!h->d_warm) HIP_TRY(hipMalloc((void**)&h->d_warm, count * sizeof(int)));
    HIP_TRY(hipMemset(h->d_warm, 0xff, count * sizeof(int))); // every entry -1: the first solve starts cold
    return COPRA_OK;
}

copra_status_t copra_batch_select_solver(copra_batch_t* h, int solver)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_select_solver: null handle");
    if (solver != COPRA_SOLVER_DEFAULT && solver != COPRA_SOLVER_QUADPROG_DENSE && solver != COPRA_SOLVER_RICCATI_IPM)
        return fail(COPRA_ERR_ARG, "copra_batch_select_solver: unknown solver flag");
    if (h->hp.ric_only && solver == COPRA_SOLVER_QUADPROG_DENSE)
        return fail(COPRA_ERR_UNSUPPORTED, "the condensed Goldfarb-Idnani kernels cover at most 512 decision variables (InitialStateLMPC: xDim <= 16)");
    if (solver == COPRA_SOLVER_RICCATI_IPM) {
        const copra_status_t rc = prepare_riccati(h);
        if (rc != COPRA_OK) return rc;
        if (!h->hs.eligible)
            return fail(COPRA_ERR_UNSUPPORTED, "the Riccati interior-point solver needs a stage-wise controller: " + h->hs.why);
        if (!h->hp.large) // the queue of non-converged instances is finished by the workgroup-per-instance kernel
            return fail(COPRA_ERR_UNSUPPORTED, "the Riccati interior-point solver covers controllers with more than 64 decision variables");
    }
    h->solver = solver;
    return COPRA_OK;
}

int copra_batch_solver_info(const copra_batch_t* h)
{
    if (!h) return -1;
    return use_riccati(const_cast<copra_batch_t*>(h)) ? COPRA_SOLVER_RICCATI_IPM : COPRA_SOLVER_QUADPROG_DENSE;
}

const char* copra_last_error(void) { return g_err.c_str(); }

#ifndef COPRA_SRC_HASH
#error "build through copra_amd/csrc/Makefile (it defines COPRA_SRC_HASH)"
#endif
const char* copra_source_hash(void) { return COPRA_SRC_HASH; }

copra_status_t copra_device_info(int* n_devices, int* cu_count, char* arch_name, int arch_name_len)
{
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (n_devices) *n_devices = n;
    if (n > 0) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        if (cu_count) *cu_count = prop.multiProcessorCount;
        if (arch_name && arch_name_len > 0) {
            strncpy(arch_name, prop.gcnArchName, (size_t)arch_name_len - 1);
            arch_name[arch_name_len - 1] = 0;
        }
    }
    return COPRA_OK;
}

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
        g_err = h->hp.error;
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
            const std::string why = rr != COPRA_OK ? g_err : h->hs.why;
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
        g_err = std::string("copra_batch_create: ") + hipGetErrorString(e);
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
    (void)hipFree(h->d_lane_ws);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->evm) (void)hipEventDestroy(h->evm);
    delete h;
}

copra_status_t copra_batch_set_system(copra_batch_t* h, const double* A, const double* B, const double* d,
    const double* x0, int on_device)
{
    if (!h || !A || !B || !d || !x0) return fail(COPRA_ERR_ARG, "copra_batch_set_system: null argument");
    const FusedPlan& P = h->hp.plan;
    const size_t b = (size_t)P.batch;
    h->shared = false; // per-instance systems again (leaves the shared-model fast path)
    if (on_device) {
        h->A = A;
        h->B = B;
        h->d = d;
        h->x0 = x0;
        return COPRA_OK;
    }
    const size_t nA = b * P.nx * P.nx, nB = b * P.nx * P.nu, nd = b * P.nx;
    if (!h->own_A) {
        HIP_TRY(hipMalloc((void**)&h->own_A, (nA ? nA : 1) * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&h->own_B, (nB ? nB : 1) * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&h->own_d, (nd ? nd : 1) * sizeof(double)));
    }
    if (!h->own_x0) HIP_TRY(hipMalloc((void**)&h->own_x0, (nd ? nd : 1) * sizeof(double)));
    HIP_TRY(hipMemcpy(h->own_A, A, nA * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->own_B, B, nB * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->own_d, d, nd * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->own_x0, x0, nd * sizeof(double), hipMemcpyHostToDevice));
    h->A = h->own_A;
    h->B = h->own_B;
    h->d = h->own_d;
    h->x0 = h->own_x0;
    return COPRA_OK;
}

// [batch][rows x cols] row-major -> per-instance column-major (what Eigen holds and every kernel here reads)
__global__ void copra_rowmajor_to_colmajor_kernel(const double* __restrict__ src, double* __restrict__ dst, int rows, int cols,
    long long total)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int per = rows * cols;
    const long long inst = e / per;
    const int w = (int)(e - inst * per), j = w / rows, i = w - j * rows; // destination: column j, row i
    dst[e] = src[inst * per + (long long)i * cols + j];
}

copra_status_t copra_batch_set_system_rowmajor_async(copra_batch_t* h, const double* A, const double* B, const double* d,
    const double* x0, void* hip_stream)
{
    if (!h || !A || !B || !d || !x0) return fail(COPRA_ERR_ARG, "copra_batch_set_system_rowmajor_async: null argument");
    const FusedPlan& P = h->hp.plan;
    const size_t b = (size_t)P.batch;
    const size_t nA = b * P.nx * P.nx, nB = b * P.nx * P.nu, nd = b * P.nx;
    h->shared = false;
    if (!h->own_A) {
        HIP_TRY(hipMalloc((void**)&h->own_A, (nA ? nA : 1) * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&h->own_B, (nB ? nB : 1) * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&h->own_d, (nd ? nd : 1) * sizeof(double)));
    }
    hipStream_t s = (hipStream_t)hip_stream;
    if (nA) hipLaunchKernelGGL(copra_rowmajor_to_colmajor_kernel, dim3((unsigned)((nA + 255) / 256)), dim3(256), 0, s, A, h->own_A, P.nx, P.nx, (long long)nA);
    if (nB) hipLaunchKernelGGL(copra_rowmajor_to_colmajor_kernel, dim3((unsigned)((nB + 255) / 256)), dim3(256), 0, s, B, h->own_B, P.nx, P.nu, (long long)nB);
    HIP_TRY(hipGetLastError());
    h->A = h->own_A;
    h->B = h->own_B;
    h->d = d; // (vectors have no layout: used in place)
    h->x0 = x0;
    return COPRA_OK;
}

int copra_batch_lanes_per_instance(const copra_batch_t* h)
{
    if (!h) return 0;
    if (h->hp.ric_only) return kWave;
    if (h->hp.large) return h->hp.plan.large.threads;
    return h->packed ? h->packed : kWave;
}

copra_status_t copra_plan_check(const copra_dims_t* dims, int n_costs, const copra_cost_desc_t* costs, int n_cstrs,
    const copra_cstr_desc_t* cstrs, const copra_initial_state_desc_t* is)
{
    if (!dims) return fail(COPRA_ERR_ARG, "copra_plan_check: null dims");
    HostPlan hp; // (the process-wide default options)
    const copra_status_t rc = build_plan(hp, *dims, n_costs, costs, n_cstrs, cstrs, is);
    if (rc != COPRA_OK) g_err = hp.error;
    if (rc == COPRA_OK && hp.ric_only) { // beyond the condensed kernels' sizes: covered if (and only if) the controller is stage-wise
        HostStagePlan hs;
        build_stage_plan(hp, hs, false);
        if (!hs.eligible)
            return fail(COPRA_ERR_UNSUPPORTED, "more than 512 decision variables (or InitialStateLMPC with xDim > 16) need a stage-wise controller for the Riccati interior-point kernel: " + hs.why);
    }
    return rc;
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
    h->shared = true;
    h->model_dirty = true;
    if (h->hp.plan.lds.ric && h->hp.plan.lds.q1regs && !h->jit_ric && ric_aot_exact(P.nx, P.nu, P.N)) { // (the Riccati-factor tier has a shared-model mode of its own: copra_batch_solve;
                                                                        //  only the library's instantiations: a run-time-compiled one has no prepare kernel)
        h->has_lds_ric = true;
        h->lds_ric = h->hp.plan.lds;
    }
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
            rtot += HP.cost[t].rows;
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
        const int r = HP.cost[t].rows;
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

copra_status_t copra_batch_set_cost_reference(copra_batch_t* h, int cost_index, const double* p, int on_device)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_set_cost_reference: null handle");
    const FusedPlan& P = h->hp.plan;
    if (cost_index < 0 || cost_index >= (int)h->hp.cost_slot.size()) return fail(COPRA_ERR_ARG, "copra_batch_set_cost_reference: no such cost");
    cost_index = h->hp.cost_slot[(size_t)cost_index]; // (dense costs are not among the kernel-evaluated terms)
    if (cost_index < 0) return fail(COPRA_ERR_UNSUPPORTED, "copra_batch_set_cost_reference: a dense (host-evaluated) cost has no reference p");
    if ((p != nullptr) != (h->cost_p[cost_index] != nullptr)) h->model_dirty = true; // shared model: c0 / C2 change
    if (!p) { // back to the controller-wide reference given at creation
        h->cost_p[cost_index] = nullptr;
        return COPRA_OK;
    }
    if (on_device) {
        h->cost_p[cost_index] = p;
        return COPRA_OK;
    }
    const size_t count = (size_t)(P.batch > 0 ? P.batch : 1) * P.cost[cost_index].prows; // (a reference trajectory: rows x steps per instance)
    if (!h->d_cost_p[cost_index]) HIP_TRY(hipMalloc((void**)&h->d_cost_p[cost_index], count * sizeof(double)));
    HIP_TRY(hipMemcpy(h->d_cost_p[cost_index], p, count * sizeof(double), hipMemcpyHostToDevice));
    h->cost_p[cost_index] = h->d_cost_p[cost_index];
    return COPRA_OK;
}

copra_status_t copra_batch_set_cost_reference_all(copra_batch_t* h, int cost_index, const double* p, int on_device)
{
    if (!h || !p) return fail(COPRA_ERR_ARG, "copra_batch_set_cost_reference_all: null argument");
    const FusedPlan& P = h->hp.plan;
    if (cost_index < 0 || cost_index >= (int)h->hp.cost_slot.size()) return fail(COPRA_ERR_ARG, "copra_batch_set_cost_reference_all: no such cost");
    const int t = h->hp.cost_slot[(size_t)cost_index];
    if (t < 0) return fail(COPRA_ERR_UNSUPPORTED, "copra_batch_set_cost_reference_all: a dense (host-evaluated) cost has no reference p");
    if (h->shared && P.cost[t].pstride)
        return fail(COPRA_ERR_UNSUPPORTED, "copra_batch_set_cost_reference_all: a new reference trajectory in shared-model mode needs a new controller");
    // Every kernel already reads a per-instance reference where one is set: the new reference is written once per instance into the
    // library's own buffer (a broadcast on the device: 66 MB at the headline's batch for a reference trajectory, ~ 10 us) and that
    // path is taken -- nothing that was derived from the creation-time p (tables of the plan builder, the shared model's c0) can go stale.
    const size_t b = (size_t)(P.batch > 0 ? P.batch : 1), rows = (size_t)P.cost[t].prows;
    if (!h->d_cost_p[t]) HIP_TRY(hipMalloc((void**)&h->d_cost_p[t], b * rows * sizeof(double)));
    double* const out = h->d_cost_p[t];
    if (p == out) return fail(COPRA_ERR_ARG, "copra_batch_set_cost_reference_all: p aliases the library's buffer");
    HIP_TRY(hipStreamSynchronize(h->last_stream)); // (a solve that still reads the buffer)
    const double* src = p;
    if (!on_device) {
        HIP_TRY(hipMemcpy(out, p, rows * sizeof(double), hipMemcpyHostToDevice)); // instance 0's slot, then read from there
        src = out;
    }
    const long long first = on_device ? 0 : (long long)rows, total = (long long)(b * rows) - first;
    if (total > 0)
        hipLaunchKernelGGL(copra_broadcast_reference_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->last_stream, src, out + first,
            (int)rows, total);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->last_stream);
    if (e != hipSuccess) return fail(COPRA_ERR_HIP, std::string("copra_batch_set_cost_reference_all: ") + hipGetErrorString(e));
    if (!h->cost_p[t]) h->model_dirty = true; // shared model: c0 / C2 change
    h->cost_p[t] = out;
    return COPRA_OK;
}

copra_status_t copra_batch_set_constraint_rhs(copra_batch_t* h, int cstr_index, const double* f, int on_device)
{
    if (!h || !f) return fail(COPRA_ERR_ARG, "copra_batch_set_constraint_rhs: null argument");
    const FusedPlan& P = h->hp.plan;
    if (cstr_index < 0 || cstr_index >= (int)h->hp.cstr_row0.size() || h->hp.cstr_row0[(size_t)cstr_index] < 0)
        return fail(COPRA_ERR_UNSUPPORTED,
            "copra_batch_set_constraint_rhs: not a Trajectory / Control / Mixed constraint of this controller "
            "(bound constraints: copra_batch_set_control_bounds)");
    const int r = h->hp.cstr_per_step[(size_t)cstr_index], steps = h->hp.cstr_steps[(size_t)cstr_index];
    const int row0 = h->hp.cstr_row0[(size_t)cstr_index];
    const size_t b = (size_t)(P.batch > 0 ? P.batch : 1);
    if (!h->d_row_f_inst) { // first use: every instance starts from the controller-wide right-hand sides
        HIP_TRY(hipMalloc((void**)&h->d_row_f_inst, b * (size_t)P.mgen * sizeof(double)));
        std::vector<double> rep(b * (size_t)P.mgen);
        for (size_t i = 0; i < b; ++i) std::copy(h->hp.row_f.begin(), h->hp.row_f.begin() + P.mgen, rep.begin() + i * P.mgen);
        HIP_TRY(hipMemcpy(h->d_row_f_inst, rep.data(), rep.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    const double* src = f;
    double* tmp = nullptr;
    if (!on_device) {
        HIP_TRY(hipMalloc((void**)&tmp, b * (size_t)r * sizeof(double)));
        hipError_t e = hipMemcpy(tmp, f, b * (size_t)r * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(tmp);
            return fail(COPRA_ERR_HIP, std::string("copra_batch_set_constraint_rhs: ") + hipGetErrorString(e));
        }
        src = tmp;
    }
    const long long total = (long long)P.batch * r * steps;
    if (total > 0) {
        hipLaunchKernelGGL(copra_scatter_rhs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->last_stream, src,
            h->d_row_f_inst, P.batch, r, steps, row0, P.mgen);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->last_stream);
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) return fail(COPRA_ERR_HIP, std::string("copra_batch_set_constraint_rhs: ") + hipGetErrorString(e));
    return COPRA_OK; // (the shared-model factorisation does not depend on right-hand sides)
}

copra_status_t copra_batch_set_control_bounds(copra_batch_t* h, const double* lower, const double* upper, int on_device)
{
    if (!h || !lower || !upper) return fail(COPRA_ERR_ARG, "copra_batch_set_control_bounds: null argument");
    const FusedPlan& P = h->hp.plan;
    const size_t count = (size_t)(P.batch > 0 ? P.batch : 1) * P.n;
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (!h->d_lb_inst) HIP_TRY(hipMalloc((void**)&h->d_lb_inst, count * sizeof(double)));
    if (!h->d_ub_inst) HIP_TRY(hipMalloc((void**)&h->d_ub_inst, count * sizeof(double)));
    HIP_TRY(hipMemcpy(h->d_lb_inst, lower, count * sizeof(double), kind));
    HIP_TRY(hipMemcpy(h->d_ub_inst, upper, count * sizeof(double), kind));
    return COPRA_OK;
}

// ---- run-time specialisation ------------------------------------------------------------------------------------
// The kernel bodies are templates on (xDim, uDim, nrStep, cost rows); the library ships instantiations for the
// BASELINE shapes and a run-time-shape one that is ~2.5x slower on the same problem (headline shape: 13.4 vs 5.4 M
// solves/s).  copra_batch_specialise compiles the instantiation for THIS controller's shape with hipcc --genco from the
// headers next to the library, keeps the code object in a cache directory and launches it through the module API.
static std::string library_dir()
{
    Dl_info info;
    if (dladdr(reinterpret_cast<const void*>(&copra_abi_version), &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        const size_t k = p.find_last_of('/');
        return k == std::string::npos ? std::string(".") : p.substr(0, k);
    }
    return ".";
}

// compile `source` (a translation unit that includes headers from the library's directory) into a code object named
// `key` in the cache directory, unless it is already there; returns its path in `obj`
static copra_status_t jit_compile(const std::string& key, const std::string& source, const char* cache_dir, std::string& obj)
{
    const std::string src_dir = library_dir();
    std::string dir = cache_dir ? cache_dir : "";
    if (dir.empty()) {
        const char* e = std::getenv("COPRA_JIT_CACHE");
        const char* home = std::getenv("HOME");
        dir = e ? e : (std::string(home ? home : "/tmp") + "/.cache/copra_amd");
    }
    (void)mkdir((dir.substr(0, dir.find_last_of('/'))).c_str(), 0755);
    (void)mkdir(dir.c_str(), 0755);
    // the code object depends on the exact sources it was compiled from: the hash of those sources is compiled into this
    // library (Makefile: COPRA_SRC_HASH), so a cache left by another build of the library is never picked up
#ifndef COPRA_SRC_HASH
#error "build through copra_amd/csrc/Makefile (it defines COPRA_SRC_HASH, the key of the run-time-compilation cache)"
#endif
    const std::string stamp = std::string(COPRA_SRC_HASH).substr(0, 12);
    obj = dir + "/" + key + "_" + stamp + ".hsaco";
    if (access(obj.c_str(), R_OK) == 0) return COPRA_OK;
    const std::string src = obj + "." + std::to_string((long)getpid()) + ".hip";
    FILE* f = fopen(src.c_str(), "w");
    if (!f) return fail(COPRA_ERR_RUNTIME, "run-time specialisation: cannot write to the cache directory " + dir);
    fputs(source.c_str(), f);
    fclose(f);
    const char* hipcc_env = std::getenv("HIPCC");
    const std::string hipcc = hipcc_env ? hipcc_env : "/opt/rocm/bin/hipcc";
    const std::string tmp = obj + "." + std::to_string((long)getpid()) + ".tmp";
    const std::string log = src + ".log";
    const std::string inc = "-I" + src_dir;
    // argv, no shell: paths with quotes or spaces cannot break (or inject into) the command
    const char* argv[] = { hipcc.c_str(), "--offload-arch=gfx950", "-O3", "-std=c++17", "--genco", inc.c_str(), "-o", tmp.c_str(),
        src.c_str(), nullptr };
    int rc = -1;
    {
        posix_spawn_file_actions_t fa;
        posix_spawn_file_actions_init(&fa);
        posix_spawn_file_actions_addopen(&fa, 1, log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        posix_spawn_file_actions_adddup2(&fa, 1, 2);
        pid_t pid = 0;
        if (posix_spawn(&pid, hipcc.c_str(), &fa, nullptr, const_cast<char* const*>(argv), environ) == 0) {
            int st = 0;
            if (waitpid(pid, &st, 0) == pid && WIFEXITED(st)) rc = WEXITSTATUS(st);
        }
        posix_spawn_file_actions_destroy(&fa);
    }
    (void)unlink(src.c_str());
    if (rc != 0 || rename(tmp.c_str(), obj.c_str()) != 0)
        return fail(COPRA_ERR_RUNTIME, "run-time specialisation: hipcc --genco failed (see " + log + ")");
    (void)unlink(log.c_str());
    return COPRA_OK;
}

// dense-QP kernels compiled for a fixed number of variables (copra_qp_dense_specialise): (n, lanes per QP) -> kernel
struct DenseJit {
    int n, lanes;
    hipFunction_t fn;
};
static std::vector<DenseJit> g_dense_jit; // guarded by g_dense_jit_mu
static std::mutex g_dense_jit_mu;

copra_status_t copra_qp_dense_specialise(int n, const char* cache_dir)
{
    if (n <= 0 || n > kWave) return COPRA_OK; // (the workgroup-per-problem kernel has no shape parameters)
    std::lock_guard<std::mutex> lock(g_dense_jit_mu);
    for (int lanes : { 64, 32, 16 }) {
        if (lanes < n) continue;
        bool have = false;
        for (const DenseJit& d : g_dense_jit) have = have || (d.n == n && d.lanes == lanes);
        if (have) continue;
        char key[96], source[1024];
        snprintf(key, sizeof key, "copra_jit_dense_%d_l%d", n, lanes);
        if (lanes == 64)
            snprintf(source, sizeof source,
                "#include <hip/hip_runtime.h>\n#include \"qp_dense.hpp\"\nusing namespace copra_hip;\n"
                "extern \"C\" __global__ __launch_bounds__(64) void copra_jit_dense(const DensePlan P)\n"
                "{ qp_dense_body<%d>(P, (int)blockIdx.x); }\n", n);
        else
            snprintf(source, sizeof source,
                "#define COPRA_WAVE_WIDTH %d\n#include \"packed_impl.inc\"\n"
                "extern \"C\" __global__ __launch_bounds__(64) void copra_jit_dense(const DensePlan P)\n"
                "{ const int inst = instance_id(); if (inst < P.batch) qp_dense_body<%d>(P, inst); }\n", lanes, n);
        std::string obj;
        const copra_status_t rc = jit_compile(key, source, cache_dir, obj);
        if (rc != COPRA_OK) return rc;
        hipModule_t mod = nullptr;
        HIP_TRY(hipModuleLoad(&mod, obj.c_str()));
        hipFunction_t fn = nullptr;
        HIP_TRY(hipModuleGetFunction(&fn, mod, "copra_jit_dense"));
        g_dense_jit.push_back(DenseJit { n, lanes, fn }); // (modules stay loaded for the life of the process)
    }
    return COPRA_OK;
}

copra_status_t copra_batch_specialise(copra_batch_t* h, const char* cache_dir)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_specialise: null handle");
    const FusedPlan& P = h->hp.plan;
    if (h->jit_fused) return COPRA_OK;
    const int rp0 = specialised_cost_rows(P.nx, P.nu, P.N, P.rmax, P.rfull);
    if (h->hp.large || P.initial_state || P.rfull > 0 || rp0 > 0 || P.n > kWave || P.nu > kMaxNu || (P.lds.ric && ric_aot_exact(P.nx, P.nu, P.N)))
        return COPRA_OK; // nothing to gain: the shape already runs on dedicated kernels (or on bodies without shape parameters)
    // ---- the Riccati-factor tier (lmpc_fused_ric.hpp; what the headline runs on) for THIS shape: per-step costs, xDim (xDim + uDim + 1)
    //      <= 64, two or three controls, at most 64 decision variables.  Compiled with Q1 in registers and in LDS (the layout ladder
    //      moves between the two); the controller takes the tier's layout once the kernels exist.
    // Single-control systems with fewer than 48 variables stay on the packed / factor-only kernels: the reference's falling-mass
    // problems hold most of their control bounds active, far beyond this tier's five register columns (measured, M solves/s,
    // this tier vs the others compiled for the shape: N = 5: 94 vs 339, 16: 6.8 vs 55, 32: 7.2 vs 16, 48: 29 vs 22, 64: 75 vs 29).
    const bool ric_pays = P.nu >= 2 || P.n >= 48 || h->hp.opt.ric_any_shape;
    if (!h->shared && ric_pays && !h->hp.opt.no_ric && !h->hp.opt.no_tri) {
        HostPlan trial = h->hp; // (the layout and the tables are only kept if everything below succeeds)
        if (take_ric_layout(trial)) {
            char keyr[128], srcr[3072];
            const char* const sr = P.stage_refs ? "true" : "false"; // (reference trajectories: the builds with the stage-varying affine term)
            snprintf(keyr, sizeof keyr, "copra_jit_ric_%d_%d_%d%s", P.nx, P.nu, P.N, P.stage_refs ? "_srefs" : "");
            snprintf(srcr, sizeof srcr,
                "#include <hip/hip_runtime.h>\n#include \"lmpc_fused_ric.hpp\"\n#include \"lmpc_lane.hpp\"\nusing namespace copra_hip;\n"
                "extern \"C\" __global__ __launch_bounds__(64, 3) void copra_jit_fused(const FusedPlan P)\n"
                "{ if (P.ovf_zero && blockIdx.x == 0 && threadIdx.x == 0) *P.ovf_zero = 0;\n"
                "  int inst; bool failed; if (!tier_instance(P, (int)blockIdx.x, inst, failed)) return;\n"
                "  lmpc_fused_ric_body<%d, %d, %d, 6, %d, %s>(P, inst, failed); }\n"
                "extern \"C\" __global__ __launch_bounds__(64, 3) void copra_jit_fused_q0(const FusedPlan P)\n"
                "{ if (P.ovf_zero && blockIdx.x == 0 && threadIdx.x == 0) *P.ovf_zero = 0;\n"
                "  int inst; bool failed; if (!tier_instance(P, (int)blockIdx.x, inst, failed)) return;\n"
                "  lmpc_fused_ric_body<%d, %d, %d, 6, 0, %s>(P, inst, failed); }\n"
                "extern \"C\" __global__ __launch_bounds__(64, 1) void copra_jit_lane(const FusedPlan P)\n"
                "{ lmpc_lane_body<%d, %d, %s>(P, (int)blockIdx.x); }\n",
                P.nx, P.nu, P.N, kFusedQ1Regs, sr, P.nx, P.nu, P.N, sr, P.nx, P.nu, sr);
            std::string objr;
            const copra_status_t rcr = jit_compile(keyr, srcr, cache_dir, objr);
            if (rcr != COPRA_OK) return rcr;
            hipModule_t modr = nullptr;
            HIP_TRY(hipModuleLoad(&modr, objr.c_str()));
            hipFunction_t fr = nullptr, fq = nullptr, fl = nullptr;
            hipError_t er = hipModuleGetFunction(&fr, modr, "copra_jit_fused");
            if (er == hipSuccess) er = hipModuleGetFunction(&fq, modr, "copra_jit_fused_q0");
            if (er == hipSuccess) er = hipModuleGetFunction(&fl, modr, "copra_jit_lane");
            double* dparams = nullptr;
            if (er == hipSuccess) er = upload(&dparams, trial.params); // (the stage-cost tables were appended)
            if (er != hipSuccess) {
                (void)hipGetLastError();
                (void)hipModuleUnload(modr);
                (void)hipFree(dparams);
                return fail(COPRA_ERR_HIP, std::string("copra_batch_specialise (Riccati-factor tier): ") + hipGetErrorString(er));
            }
            (void)hipFree(h->d_params);
            h->d_params = dparams;
            h->hp = trial;
            h->packed = 0; // (one instance per wavefront on this tier)
            h->lds_attr_set = false;
            h->adapt_left = h->adapt_left > 4 ? h->adapt_left : 4;
            h->jit_module = modr;
            h->jit_lanes = 64;
            h->jit_tri = 1;
            h->jit_ric = true;
            h->jit_fused = fr;
            h->jit_fused_q0 = fq;
            h->jit_lane = fl;
            h->jit_shared = nullptr;
            return COPRA_OK;
        }
    }
    // (always the full register budget: with compile-time trip counts the unrolled bodies spill at 128 VGPRs -- double
    //  integrator N = 32: 9.2 M solves/s at four waves per SIMD, 15.6 M at two, 11.8 M for the run-time-shape kernel)
    char key[128], source[1536];
    snprintf(key, sizeof key, "copra_jit_%d_%d_%d_%d_l%d%s", P.nx, P.nu, P.N, P.rmax, h->packed ? h->packed : 64, P.lds.tri ? "t" : "");
    if (h->packed) // several small instances per wavefront: the same bodies on the group-wide primitives
        snprintf(source, sizeof source,
            "#define COPRA_WAVE_WIDTH %d\n#include \"packed_impl.inc\"\n"
            "extern \"C\" __global__ __launch_bounds__(64) void copra_jit_fused(const FusedPlan P)\n"
            "{ const int inst = P.inst_offset + instance_id(); if (inst < P.batch) lmpc_fused_body<%d, %d, %d, %d>(P, inst); }\n"
            "extern \"C\" __global__ __launch_bounds__(64) void copra_jit_shared(const FusedPlan P)\n"
            "{ const int inst = instance_id(); if (inst < P.batch) lmpc_shared_body<%d, %d, %d>(P, inst); }\n",
            h->packed, P.nx, P.nu, P.N, P.rmax, P.nx, P.nu, P.N);
    else
        snprintf(source, sizeof source,
            "#include <hip/hip_runtime.h>\n#include \"lmpc_fused.hpp\"\n#include \"lmpc_shared.hpp\"\nusing namespace copra_hip;\n"
            "extern \"C\" __global__ __launch_bounds__(64%s) void copra_jit_fused(const FusedPlan P)\n"
            "{ lmpc_fused_body<%d, %d, %d, %d, %s>(P, P.inst_offset + (int)blockIdx.x); }\n"
            "extern \"C\" __global__ __launch_bounds__(64%s) void copra_jit_shared(const FusedPlan P)\n"
            "{ lmpc_shared_body<%d, %d, %d, %s>(P, (int)blockIdx.x); }\n",
            P.lds.tri ? ", 2" : "", P.nx, P.nu, P.N, P.rmax, P.lds.tri ? "true" : "false", P.lds.tri ? ", 2" : "", P.nx, P.nu,
            P.N, P.lds.tri ? "true" : "false");
    std::string obj;
    {
        const copra_status_t rcj = jit_compile(key, source, cache_dir, obj);
        if (rcj != COPRA_OK) return rcj;
    }
    hipModule_t mod = nullptr;
    HIP_TRY(hipModuleLoad(&mod, obj.c_str()));
    hipFunction_t f1 = nullptr, f2 = nullptr;
    hipError_t e = hipModuleGetFunction(&f1, mod, "copra_jit_fused");
    if (e == hipSuccess) e = hipModuleGetFunction(&f2, mod, "copra_jit_shared");
    const size_t jit_lds = (size_t)(h->packed ? 64 / h->packed : 1) * h->hp.lds_bytes;
    if (e == hipSuccess && jit_lds > 48 * 1024) { // more than the default dynamic-LDS limit
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(f1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)jit_lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(f2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)jit_lds);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipModuleUnload(mod);
        return fail(COPRA_ERR_HIP, std::string("copra_batch_specialise: ") + hipGetErrorString(e));
    }
    h->jit_module = mod;
    h->jit_lanes = h->packed ? h->packed : 64;
    h->jit_tri = P.lds.tri;
    h->jit_fused = f1;
    h->jit_shared = f2;
    return COPRA_OK;
}

copra_status_t copra_batch_set_x0(copra_batch_t* h, const double* x0, int on_device)
{
    if (!h || !x0) return fail(COPRA_ERR_ARG, "copra_batch_set_x0: null argument");
    const FusedPlan& P = h->hp.plan;
    if (on_device) {
        h->x0 = x0;
        return COPRA_OK;
    }
    const size_t nd = (size_t)P.batch * P.nx;
    if (!h->own_x0) HIP_TRY(hipMalloc((void**)&h->own_x0, (nd ? nd : 1) * sizeof(double)));
    HIP_TRY(hipMemcpy(h->own_x0, x0, nd * sizeof(double), hipMemcpyHostToDevice));
    h->x0 = h->own_x0;
    return COPRA_OK;
}

copra_status_t copra_batch_set_outputs(copra_batch_t* h, double* control, double* trajectory, int* status, int* iter)
{
    if (!h || !control || !trajectory || !status || !iter)
        return fail(COPRA_ERR_ARG, "copra_batch_set_outputs: null argument");
    h->ext_control = control;
    h->ext_traj = trajectory;
    h->ext_status = status;
    h->ext_iter = iter;
    return COPRA_OK;
}

copra_status_t copra_batch_layout_info(const copra_batch_t* h, int* lds_bytes, int* active_capacity, int* factor_only,
    int* two_tier)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_layout_info: null handle");
    if (lds_bytes) *lds_bytes = (int)h->hp.lds_bytes;
    if (active_capacity) *active_capacity = h->hp.large ? h->hp.plan.n : h->hp.plan.lds.rcap;
    if (factor_only) *factor_only = h->hp.large ? 0 : h->hp.plan.lds.tri;
    if (two_tier) *two_tier = h->hp.two_tier ? 1 : 0;
    return COPRA_OK;
}

copra_status_t copra_batch_solve(copra_batch_t* h, void* hip_stream)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_solve: null handle");
    {
        copra_status_t rca = adapt_lane_pass(h);
        if (rca != COPRA_OK) return rca;
        rca = adapt_layout(h);
        if (rca != COPRA_OK) return rca;
        h->solved_once = true;
        h->lane_ran = false;
    }
    if (h->shared) {
        if (!h->x0) return fail(COPRA_ERR_RUNTIME, "copra_batch_solve: no initial states set (copra_batch_set_x0)");
        if (h->hp.plan.stage_refs) // (the model's dc/dp holds one column per cost row, not per row and step)
            for (int t = 0; t < kMaxCosts; ++t)
                if (h->cost_p[t] && h->hp.plan.cost[t].pstride)
                    return fail(COPRA_ERR_UNSUPPORTED, "copra_batch_solve: per-instance reference trajectories are not available in shared-model mode");

        hipStream_t s = (hipStream_t)hip_stream;
        h->last_stream = s;
        if (h->hp.plan.batch == 0) return COPRA_OK;
        { // which first tier: the Riccati-factor tier in shared-model mode (cold starts, controller-wide references), or lmpc_shared.hpp
            bool want = h->has_lds_ric && !h->d_warm && !h->hp.opt.no_ric_shared;
            for (int t = 0; t < kMaxCosts; ++t) want = want && !h->cost_p[t];
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
        }
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
            FusedPlan Pr = P;
            Pr.ric_model = h->d_ric_model;
            // in front of it the one-instance-per-lane pass in its shared-model form (lmpc_lane.hpp): the roll-out of every instance from
            // the batch-wide records; the tier solves what it leaves over, starting from the U and X it wrote
            unsigned g1 = (unsigned)P.batch;
            if (!h->lane_off && !h->hp.opt.no_lane_pass && lane_batch_ok(h->hp.opt, P.batch, true) && P.lane_tab >= 0 && P.lds.ricC && !P.prof && !P.prof_fine
                && !P.row_f_inst && select_lane_shared_kernel(P) && (ensure_lane_buffers(h, false) == COPRA_OK || (h->lane_off = true, false))) {
                // (no room for the pass's list: the tier alone, from now on -- as on the per-instance path below)
                h->lane_cur ^= 1;
                h->lane_ran = true;
                Pr.lane_list = h->d_lane_list;
                Pr.lane_count = h->d_lane_count + h->lane_cur;
                Pr.lane_zero = h->d_lane_count + (h->lane_cur ^ 1);
                Pr.lane_bp = (int)(((size_t)P.batch + kWave - 1) / kWave * kWave);
                hipLaunchKernelGGL(select_lane_shared_kernel(Pr), dim3((unsigned)(Pr.lane_bp / kWave)), dim3(64), lane_lds_bytes(Pr), s, Pr);
                HIP_TRY(hipGetLastError());
                Pr.lane_from_list = 1;
                Pr.lane_handover = 1;
                Pr.lane_zero = nullptr;
                g1 = ((unsigned)P.batch + 7u) & ~7u; // (the list is dealt out in eighths: ric_tier_instance)
            }
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
    if (!h->A || !h->B || !h->d || !h->x0)
        return fail(COPRA_ERR_RUNTIME, "copra_batch_solve: no preview system set (copra_batch_set_system)");
    FusedPlan P = device_plan(h);
    hipStream_t s = (hipStream_t)hip_stream;
    h->last_stream = s;
    if (P.batch == 0) return COPRA_OK;
    copra_status_t rc = ensure_lds_attr(h);
    if (rc != COPRA_OK) return rc;
    // Timing (copra_batch_last_solve_seconds).  The plain one-wave launches carry their events IN their dispatch packets
    // (hipExtLaunchKernelGGL): start of the first launch, end of the first launch (copra_batch_last_first_tier_seconds) and end
    // of the second one (the whole solve, what LMPC::solveTime() reports) -- no barrier packets in the stream: two
    // hipEventRecord per solve cost ~ 20 us between consecutive solves, 3 % of the headline step.  The other paths bracket
    // their launches with recorded events as before.
    const bool jit_launch = h->jit_fused && h->jit_lanes == (h->packed ? h->packed : 64) && h->jit_tri == P.lds.tri
        && h->jit_ric == (P.lds.ric != 0);
    const bool ext_timed = !h->hp.large && !P.initial_state && !jit_launch && !h->packed && !h->hp.opt.recorded_events;
    if (!ext_timed) HIP_TRY(hipEventRecord(h->ev0, s));
    if (h->hp.large) {
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
    if (P.initial_state) {
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
    // the one-instance-per-lane pass (lmpc_lane.hpp): every instance whose unconstrained minimiser violates nothing ends in it, the
    // first tier below runs for the others only
    bool lane_pass = lane_pass_wanted(h, P, jit_launch);
    if (lane_pass && ensure_lane_buffers(h, true) != COPRA_OK) { // (no room for its workspace: the tier alone, from now on)
        (void)hipGetLastError();
        h->lane_off = true;
        lane_pass = false;
    }
    if (lane_pass) {
        h->lane_cur ^= 1;
        h->lane_ran = true;
        P.lane_ws = h->d_lane_ws;
        P.lane_list = h->d_lane_list;
        P.lane_count = h->d_lane_count + h->lane_cur;
        P.lane_zero = h->d_lane_count + (h->lane_cur ^ 1);
        P.lane_bp = (int)(((size_t)P.batch + kWave - 1) / kWave * kWave) + kWave;
        // instances per wave: 64.  (copra_options_t::lane_group = 32 runs HALF-WAVES -- twice the waves, lanes 32.. idle -- which was meant
        // to fill the machine at a shard of BASELINE configs[3], 32 768 instances = 512 full waves on 1024 SIMDs: measured NO faster,
        // 0.295 vs 0.286 ms per 32 768 and 0.575 vs 0.471 ms per 65 536 -- the pass's wave time is its arithmetic and its own dependent memory
        // trips, not contention: profiles/r04/lane_half_waves.txt.  Kept as an experiment switch.)
        P.lane_group = h->hp.opt.lane_group == 32 ? 32 : 64;
        const unsigned g0 = (unsigned)(((long long)P.batch + P.lane_group - 1) / P.lane_group);
        P.lane_dbg = h->hp.opt.lane_dbg;
        P.lane_handover = (P.lds.ricC && !h->hp.opt.no_lane_handover) ? 1 : 0; // (the pass leaves Lam^-1 and the norm sums only for a tier that takes them)
        // first solve of a controller on a factor-only tier with a layout ladder: the pass also counts, per instance it leaves over, the
        // rows its unconstrained minimiser violates; the layout the tier STARTS on is chosen from that histogram (below)
        const bool predict = h->lane_predict_left > 0 && h->hp.two_tier && P.lds.tri && !h->shared && !h->hp.opt.no_ladder;
        if (predict) P.lane_hist = h->d_lane_hist;
        if (jit_launch) {
            FusedPlan Pl = P;
            void* largs[] = { &Pl };
            LDS_OPT_IN(h->jit_lane, lane_lds_bytes(P));
            HIP_TRY(hipModuleLaunchKernel(h->jit_lane, g0, 1, 1, 64, 1, 1, (unsigned)lane_lds_bytes(P), s, largs, nullptr));
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
            h->lane_predict_left -= 1;
            int hist[kLaneHistBins];
            HIP_TRY(hipStreamSynchronize(s));
            HIP_TRY(hipMemcpy(hist, h->d_lane_hist, sizeof hist, hipMemcpyDeviceToHost));
            long long share = h->hp.plan.lds.ric ? 64 : 8;
            if (h->hp.opt.overflow_share > 0) share = h->hp.opt.overflow_share;
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
        P.lane_handover = (P.lds.ricC && !h->hp.opt.no_lane_handover) ? 1 : 0;
        P.lane_zero = nullptr;
    }
    if (h->hp.two_tier) HIP_TRY(begin_overflow_queue(h, s, P.lds.ric && (!jit_launch || h->jit_ric) && !h->packed, P));
    if (jit_launch) {
        FusedPlan Pj = P;
        void* args[] = { &Pj };
        const unsigned per = 64u / (unsigned)h->jit_lanes;
        const hipFunction_t jfn = (h->jit_ric && P.lds.q1regs == 0) ? h->jit_fused_q0 : h->jit_fused;
        LDS_OPT_IN(jfn, (size_t)per * h->hp.lds_bytes);
        const unsigned gj = lane_pass ? (((unsigned)P.batch + 7u) & ~7u) : ((unsigned)P.batch + per - 1) / per; // (the list is dealt out in eighths)
        HIP_TRY(hipModuleLaunchKernel(jfn, gj, 1, 1, 64, 1, 1, per * (unsigned)h->hp.lds_bytes, s, args, nullptr));
    } else if (h->packed) {
        HIP_TRY(h->packed == 16 ? packed_launch_w16(P, false, h->hp.lds_bytes, s) : packed_launch_w32(P, false, h->hp.lds_bytes, s));
    } else {
        LDS_OPT_IN(select_fused_kernel(P), h->hp.lds_bytes);
        if (h->hp.opt.debug) {
            int per_cu = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(select_fused_kernel(P)), 64, h->hp.lds_bytes);
            fprintf(stderr, "[copra] fused first tier: %zu B LDS per instance, %d columns, q1regs %d, occupancy API: %d instances per CU\n",
                h->hp.lds_bytes, P.lds.rcap, P.lds.q1regs, per_cu);
        }
        const unsigned g1 = lane_pass ? (((unsigned)P.batch + 7u) & ~7u) : (unsigned)P.batch; // (the list is dealt out in eighths)
        if (ext_timed) // (start | end of the first launch; with a second launch the solve ends with THAT kernel's packet)
            hipExtLaunchKernelGGL(select_fused_kernel(P), dim3(g1), dim3(64), h->hp.lds_bytes, s,
                lane_pass ? nullptr : h->ev0, h->hp.two_tier ? h->evm : h->ev1, 0, P);
        else
            hipLaunchKernelGGL(select_fused_kernel(P), dim3(g1), dim3(64), h->hp.lds_bytes, s, P);
        HIP_TRY(hipGetLastError());
    }
    if (h->hp.two_tier) {
        // second tier: same kernel, full LDS layout, instances taken from the overflow queue (usually empty)
        FusedPlan P2 = P;
        P2.lds = h->hp.lds_full;
        P2.from_list = 1;
        const unsigned g2 = (unsigned)(P.batch < 1024 ? P.batch : 1024);
        LDS_OPT_IN(select_tier2_kernel(P2), h->hp.lds_full_bytes);
        if (ext_timed)
            hipExtLaunchKernelGGL(select_tier2_kernel(P2), dim3(g2), dim3(64), h->hp.lds_full_bytes, s, nullptr, h->ev1, 0, P2);
        else
            hipLaunchKernelGGL(select_tier2_kernel(P2), dim3(g2), dim3(64), h->hp.lds_full_bytes, s, P2);
        HIP_TRY(hipGetLastError());
    }
    if (!ext_timed) HIP_TRY(hipEventRecord(h->ev1, s));
    h->tier_timed = ext_timed && h->hp.two_tier;
    h->timed = true;
    return COPRA_OK;
}

copra_status_t copra_batch_synchronize(copra_batch_t* h)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_synchronize: null handle");
    HIP_TRY(hipStreamSynchronize(h->last_stream));
    return COPRA_OK;
}

const double* copra_batch_control_device(const copra_batch_t* h)
{
    return h ? (h->ext_control ? h->ext_control : h->d_control) : nullptr;
}
const double* copra_batch_trajectory_device(const copra_batch_t* h)
{
    return h ? (h->ext_traj ? h->ext_traj : h->d_traj) : nullptr;
}
const int* copra_batch_status_device(const copra_batch_t* h)
{
    return h ? (h->ext_status ? h->ext_status : h->d_status) : nullptr;
}
const int* copra_batch_iter_device(const copra_batch_t* h)
{
    return h ? (h->ext_iter ? h->ext_iter : h->d_iter) : nullptr;
}

copra_status_t copra_batch_get_results(copra_batch_t* h, double* control, double* trajectory, int* status, int* iter)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_get_results: null handle");
    const FusedPlan& P = h->hp.plan;
    const size_t b = (size_t)P.batch;
    if (h->h_results && !h->ext_control && !h->ext_traj && !h->ext_status && !h->ext_iter) {
        // small batch (the single-problem use of copra::LMPC::solve() above all): ONE asynchronous copy of the whole slab into
        // pinned memory behind the solve, one synchronisation -- instead of a stream synchronisation and four blocking copies
        HIP_TRY(hipMemcpyAsync(h->h_results, h->d_results, h->results_bytes, hipMemcpyDeviceToHost, h->last_stream));
        HIP_TRY(hipStreamSynchronize(h->last_stream));
        if (control) std::memcpy(control, h->h_results, b * P.n * sizeof(double));
        if (trajectory) std::memcpy(trajectory, h->h_results + h->off_traj, b * P.X * sizeof(double));
        if (status) std::memcpy(status, h->h_results + h->off_status, b * sizeof(int));
        if (iter) std::memcpy(iter, h->h_results + h->off_iter, b * 2 * sizeof(int));
        return COPRA_OK;
    }
    HIP_TRY(hipStreamSynchronize(h->last_stream));
    if (control) HIP_TRY(hipMemcpy(control, copra_batch_control_device(h), b * P.n * sizeof(double), hipMemcpyDeviceToHost));
    if (trajectory) HIP_TRY(hipMemcpy(trajectory, copra_batch_trajectory_device(h), b * P.X * sizeof(double), hipMemcpyDeviceToHost));
    if (status) HIP_TRY(hipMemcpy(status, copra_batch_status_device(h), b * sizeof(int), hipMemcpyDeviceToHost));
    if (iter) HIP_TRY(hipMemcpy(iter, copra_batch_iter_device(h), b * 2 * sizeof(int), hipMemcpyDeviceToHost));
    return COPRA_OK;
}

copra_status_t copra_batch_set_initial_state_bounds(copra_batch_t* h, const double* x0lb, const double* x0ub,
    int on_device)
{
    if (!h || !x0lb || !x0ub) return fail(COPRA_ERR_ARG, "copra_batch_set_initial_state_bounds: null argument");
    if (!h->hp.plan.initial_state) return fail(COPRA_ERR_RUNTIME, "not an InitialStateLMPC controller");
    const size_t nd = (size_t)h->hp.plan.batch * h->hp.plan.nx;
    if (on_device) {
        h->x0lb = x0lb;
        h->x0ub = x0ub;
        return COPRA_OK;
    }
    if (!h->own_x0lb) {
        HIP_TRY(hipMalloc((void**)&h->own_x0lb, (nd ? nd : 1) * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&h->own_x0ub, (nd ? nd : 1) * sizeof(double)));
    }
    HIP_TRY(hipMemcpy(h->own_x0lb, x0lb, nd * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->own_x0ub, x0ub, nd * sizeof(double), hipMemcpyHostToDevice));
    h->x0lb = h->own_x0lb;
    h->x0ub = h->own_x0ub;
    return COPRA_OK;
}

copra_status_t copra_batch_get_initial_state(copra_batch_t* h, double* x0_opt)
{
    if (!h || !x0_opt) return fail(COPRA_ERR_ARG, "copra_batch_get_initial_state: null argument");
    if (!h->hp.plan.initial_state) return fail(COPRA_ERR_RUNTIME, "not an InitialStateLMPC controller");
    HIP_TRY(hipStreamSynchronize(h->last_stream));
    HIP_TRY(hipMemcpy(x0_opt, h->d_x0opt, (size_t)h->hp.plan.batch * h->hp.plan.nx * sizeof(double), hipMemcpyDeviceToHost));
    return COPRA_OK;
}

copra_status_t copra_batch_qp_sizes(const copra_batch_t* h, int* nvar, int* neq, int* nineq)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_qp_sizes: null handle");
    if (nvar) *nvar = h->hp.plan.initial_state ? h->hp.plan.nx + h->hp.plan.n : h->hp.plan.n;
    if (neq) *neq = h->hp.plan.meq;
    if (nineq) *nineq = h->hp.plan.mineq;
    return COPRA_OK;
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

copra_status_t copra_batch_phase_profile(copra_batch_t* h, int enable, long long* cycles_out)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_phase_profile: null handle");
    const size_t b = (size_t)(h->hp.plan.batch > 0 ? h->hp.plan.batch : 1);
    if (enable && !h->d_prof) {
        HIP_TRY(hipMalloc((void**)&h->d_prof, b * 8 * sizeof(long long)));
        HIP_TRY(hipMemset(h->d_prof, 0, b * 8 * sizeof(long long)));
    }
    if (cycles_out) {
        if (!h->d_prof) return fail(COPRA_ERR_RUNTIME, "copra_batch_phase_profile: profiling was not enabled");
        HIP_TRY(hipStreamSynchronize(h->last_stream));
        HIP_TRY(hipMemcpy(cycles_out, h->d_prof, b * 8 * sizeof(long long), hipMemcpyDeviceToHost));
    }
    if (!enable && h->d_prof) {
        (void)hipFree(h->d_prof);
        h->d_prof = nullptr;
    }
    return COPRA_OK;
}

#ifdef COPRA_FINE_PROFILE
// profiling builds only (libcopra_hip_prof.so): 32 raw shader-clock stamps per instance, -1 = unused
copra_status_t copra_batch_fine_profile(copra_batch_t* h, long long* out)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_fine_profile: null handle");
    const size_t b = (size_t)(h->hp.plan.batch > 0 ? h->hp.plan.batch : 1);
    if (!h->d_prof_fine) {
        HIP_TRY(hipMalloc((void**)&h->d_prof_fine, b * 32 * sizeof(long long)));
        HIP_TRY(hipMemset(h->d_prof_fine, 0xff, b * 32 * sizeof(long long)));
    }
    if (out) {
        HIP_TRY(hipStreamSynchronize(h->last_stream));
        HIP_TRY(hipMemcpy(out, h->d_prof_fine, b * 32 * sizeof(long long), hipMemcpyDeviceToHost));
    }
    return COPRA_OK;
}
#endif

copra_status_t copra_batch_last_solve_seconds(copra_batch_t* h, double* seconds)
{
    if (!h || !seconds) return fail(COPRA_ERR_ARG, "copra_batch_last_solve_seconds: null argument");
    if (!h->timed) return fail(COPRA_ERR_RUNTIME, "copra_batch_last_solve_seconds: no solve has been launched");
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *seconds = (double)ms * 1e-3;
    return COPRA_OK;
}

copra_status_t copra_batch_last_first_tier_seconds(copra_batch_t* h, double* seconds)
{
    if (!h || !seconds) return fail(COPRA_ERR_ARG, "copra_batch_last_first_tier_seconds: null argument");
    if (!h->timed) return fail(COPRA_ERR_RUNTIME, "copra_batch_last_first_tier_seconds: no solve has been launched");
    if (!h->tier_timed) return copra_batch_last_solve_seconds(h, seconds);
    HIP_TRY(hipEventSynchronize(h->evm));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->evm));
    *seconds = (double)ms * 1e-3;
    return COPRA_OK;
}

copra_status_t copra_batch_lane_pass_info(copra_batch_t* h, int* ran, int* finished)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_lane_pass_info: null handle");
    if (ran) *ran = h->lane_ran ? 1 : 0;
    if (finished) {
        *finished = 0;
        if (h->lane_ran) {
            int left = 0;
            HIP_TRY(hipStreamSynchronize(h->last_stream));
            HIP_TRY(hipMemcpy(&left, h->d_lane_count + h->lane_cur, sizeof(int), hipMemcpyDeviceToHost));
            *finished = h->hp.plan.batch - left;
        }
    }
    return COPRA_OK;
}

copra_status_t copra_qp_solve_dense_batch(int batch, int n, int neq, int nineq, const double* Q, const double* c,
    const double* Aeq, const double* beq, const double* Aineq, const double* bineq, const double* XL,
    const double* XU, double* x, int* failv, int* iter, int on_device, void* hip_stream)
{
    if (batch < 0 || n <= 0 || neq < 0 || nineq < 0) // SI_problem(nrVar, nrEq, nrInEq)
        return fail(COPRA_ERR_DOMAIN, "copra_qp_solve_dense_batch: bad problem sizes");
    if (!Q || !c || !XL || !XU || !x || !failv || (neq > 0 && (!Aeq || !beq)) || (nineq > 0 && (!Aineq || !bineq)))
        return fail(COPRA_ERR_ARG, "copra_qp_solve_dense_batch: null argument");
    if (n > kLargeMaxN) return fail(COPRA_ERR_UNSUPPORTED, "dense QP with more than 512 variables is not covered yet");
    if (batch == 0) return COPRA_OK;
    const bool large = n > kWave;
    hipStream_t s = (hipStream_t)hip_stream;
    DensePlan P {};
    P.n = n;
    P.meq = neq;
    P.mineq = nineq;
    P.mgen = neq + nineq;
    P.mtotal = P.mgen + 2 * n; // QuadProgSolver.cpp:51
    P.batch = batch;
    P.vsmall = qpgen2_vsmall();
    P.max_iter = 50 * (n + P.mtotal) + 100;
    size_t lds_bytes;
    if (large) {
        layout_large_solver(P.llds, 0, n, P.mgen, P.meq, P.mtotal);
        lds_bytes = (size_t)P.llds.total * sizeof(double);
    } else {
        (void)layout_lds(P.lds, 0, 0, 0, n, 0, 1, P.mgen, P.meq, P.mtotal, false);
        lds_bytes = (size_t)P.lds.total * sizeof(double);
    }
    if (lds_bytes > 160u * 1024u) return fail(COPRA_ERR_UNSUPPORTED, "dense QP does not fit LDS");
    if (lds_bytes > 48 * 1024) {
        HIP_TRY(hipFuncSetAttribute(large ? reinterpret_cast<const void*>(copra_qp_dense_large_kernel)
                                          : reinterpret_cast<const void*>(copra_qp_dense_kernel),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        if (large)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(copra_qp_dense_large_kernel_w4),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }
    const size_t b = (size_t)batch;
    std::vector<void*> owned;
    auto release = [&]() {
        for (void* p : owned) (void)hipFree(p);
    };
    hipError_t e = hipSuccess;
    auto to_dev = [&](const double* src, size_t count) -> const double* {
        if (on_device) return src;
        double* dptr = nullptr;
        hipError_t r = hipMalloc((void**)&dptr, (count ? count : 1) * sizeof(double));
        if (r == hipSuccess && count) r = hipMemcpyAsync(dptr, src, count * sizeof(double), hipMemcpyHostToDevice, s);
        if (r != hipSuccess && e == hipSuccess) e = r;
        owned.push_back(dptr);
        return dptr;
    };
    P.Q = to_dev(Q, b * n * n);
    P.c = to_dev(c, b * n);
    P.Aeq = to_dev(Aeq, b * neq * n);
    P.beq = to_dev(beq, b * neq);
    P.Aineq = to_dev(Aineq, b * nineq * n);
    P.bineq = to_dev(bineq, b * nineq);
    P.XL = to_dev(XL, b * n);
    P.XU = to_dev(XU, b * n);
    double* dx = x;
    int *dfail = failv, *diter = iter;
    if (!on_device) {
        hipError_t r = hipMalloc((void**)&dx, b * n * sizeof(double));
        if (r == hipSuccess) r = hipMalloc((void**)&dfail, b * sizeof(int));
        if (r == hipSuccess) r = hipMalloc((void**)&diter, b * 2 * sizeof(int));
        if (r != hipSuccess && e == hipSuccess) e = r;
        owned.push_back(dx);
        owned.push_back(dfail);
        owned.push_back(diter);
    } else if (!diter) {
        hipError_t r = hipMalloc((void**)&diter, b * 2 * sizeof(int));
        if (r != hipSuccess && e == hipSuccess) e = r;
        owned.push_back(diter);
    }
    if (e != hipSuccess) {
        release();
        return fail(COPRA_ERR_HIP, std::string("copra_qp_solve_dense_batch: ") + hipGetErrorString(e));
    }
    P.x = dx;
    P.fail = dfail;
    P.iter = diter;
    if (large) {
        const int threads = (n + kWave - 1) & ~(kWave - 1);
        const bool w4 = prefer_w4(default_options(), reinterpret_cast<const void*>(copra_qp_dense_large_kernel),
            reinterpret_cast<const void*>(copra_qp_dense_large_kernel_w4), threads, lds_bytes);
        auto dense_kernel = w4 ? copra_qp_dense_large_kernel_w4 : copra_qp_dense_large_kernel;
        const int grid = large_grid(default_options(), reinterpret_cast<const void*>(dense_kernel), batch, threads, lds_bytes);
        double* ws = nullptr;
        e = hipMalloc((void**)&ws, (size_t)grid * 2 * n * large_ld(n) * sizeof(double));
        if (e != hipSuccess) {
            release();
            return fail(COPRA_ERR_HIP, std::string("copra_qp_solve_dense_batch: ") + hipGetErrorString(e));
        }
        owned.push_back(ws);
        P.ws = ws;
        hipLaunchKernelGGL(dense_kernel, dim3((unsigned)grid), dim3((unsigned)threads), lds_bytes, s, P);
    } else {
        const int pw = default_options().no_packed ? 0 : packed_width(n, 0, false, lds_bytes);
        hipFunction_t jit = nullptr;
        {
            std::lock_guard<std::mutex> lock(g_dense_jit_mu);
            for (const DenseJit& d : g_dense_jit)
                if (d.n == n && d.lanes == (pw ? pw : 64)) jit = d.fn;
        }
        const unsigned per = pw ? 64u / (unsigned)pw : 1u;
        if (jit && (size_t)per * lds_bytes <= 48 * 1024) {
            DensePlan Pj = P;
            void* args[] = { &Pj };
            e = hipModuleLaunchKernel(jit, ((unsigned)batch + per - 1) / per, 1, 1, 64, 1, 1, per * (unsigned)lds_bytes, s, args, nullptr);
        } else if (pw == 16)
            e = packed_dense_launch_w16(P, lds_bytes, s);
        else if (pw == 32)
            e = packed_dense_launch_w32(P, lds_bytes, s);
        else
            hipLaunchKernelGGL(copra_qp_dense_kernel, dim3((unsigned)batch), dim3(64), lds_bytes, s, P);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess && !on_device) {
        e = hipMemcpyAsync(x, dx, b * n * sizeof(double), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(failv, dfail, b * sizeof(int), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && iter) e = hipMemcpyAsync(iter, diter, b * 2 * sizeof(int), hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess && !owned.empty()) e = hipStreamSynchronize(s);
    release();
    if (e != hipSuccess) return fail(COPRA_ERR_HIP, std::string("copra_qp_solve_dense_batch: ") + hipGetErrorString(e));
    return COPRA_OK;
}

} // extern "C"
