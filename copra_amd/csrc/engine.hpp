// engine.hpp -- what the translation units of libcopra_hip.so share on the HOST side: the controller handle, the error slot, the HIP
// helpers and the handful of functions one unit calls in another.  Units:
//   copra_hip.hip          kernels of the controller paths, kernel selection, create / destroy, copra_batch_solve, shared-model and
//                          Riccati preparation, parity dump
//   copra_hip_ric.hip      run-time-horizon builds of the headline's kernels (ric_kernels.hpp)
//   copra_hip_setters.hip  everything that hands data in or out: set_* / get_* / results / timing / profile
//   copra_hip_jit.hip      copra_batch_specialise, copra_qp_dense_specialise (hipcc --genco at run time)
//   copra_hip_qp.hip       plug-in point 1: copra_qp_solve_dense_batch and its kernels
//   copra_hip_packed16/32.hip  the one-wave bodies with 16 / 32 lanes per instance
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "../../include/copra_hip.h"
#include "packed_launch.hpp"
#include "plan_builder.hpp"
#include "stage_plan.hpp"

#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace copra_hip;

extern thread_local std::string g_copra_err; // what copra_last_error() returns
inline copra_status_t fail(copra_status_t code, const std::string& msg)
{
    g_copra_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                                 \
    do {                                                                                                              \
        hipError_t e_ = (expr);                                                                                       \
        if (e_ != hipSuccess) return fail(COPRA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

// Dynamic LDS beyond 48 KiB is an opt-in PER KERNEL SYMBOL (copra_hip.hip); remembered per function
hipError_t lds_opt_in(const void* fn, size_t bytes);
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

// persistent grids of the workgroup-per-instance kernels (copra_hip.hip)
int large_per_cu(const void* kernel, int threads, size_t lds_bytes);
int large_grid(const copra_options_t& opt, const void* kernel, int batch, int threads, size_t lds_bytes);
void see_axis_order(copra_batch* h, const double* A, const double* B, bool on_device);
bool prefer_w4(const copra_options_t& opt, const void* full, const void* w4, int threads, size_t lds_bytes);

constexpr size_t kSmallSlab = 1u << 20; // result slabs up to this size are fetched with one copy through pinned memory

// What the engine has LEARNT about a controller's workload.  Every decision that makes the choice of kernels (never their results) depend on
// earlier solves reads and writes this struct and nothing else: round-4 verdict -- four adaptive controllers spread over the handle were the
// place where the variants interacted.  copra_hip.hip: adapt_lane_pass, adapt_layout, rechoose_layout, the pass's histogram, the shared tier's choice.
struct AdaptState {
    bool solved_once = false;
    // the layout ladder of the first tier
    int adapt_left = 3; // solves after which the overflow count of a compact layout is still checked (adapt_layout)
    int lane_predict_left = 1; // solves whose first-tier layout is still chosen from the pass's violated-row histogram
    LdsLayout lds_top {}; // the layout the ladder started on: the choice is made again every 256 solves, from the top (rechoose_layout) -- a
    bool lds_top_set = false; // controller whose constraints relax gets its denser layout back
    long long layout_solves = 0; // solves completed on the current ladder
    // the one-instance-per-lane pass
    bool lane_ran = false; // the last solve ran it
    bool lane_off = false; // switched off for this controller: too few instances end in it (adapt_lane_pass), or no memory for its workspace
    bool lane_off_by_share = false; // ... the former: sampled again every 256 solves
    int lane_adapt_left = 2;
    bool lane_spec_shared_off = false; // the shared-model form of the pass does not take steps itself (too few instances end by them)
    bool lane_form_handover = false; // the pass runs in its hand-over form (too few instances end in the speculating one: adapt_lane_pass)
    bool lane_ws2_failed = false; // ... whose blocks found no room once (ensure_lane_buffers): the speculating form stays
    long long lane_solves = 0; // solves seen by adapt_lane_pass
    // the one-(instance, axis)-per-lane solver (lmpc_axis.hpp)
    bool axis_ran = false; // the last solve ran it
    bool axis_quiet = false; // ... without a second chance and a first tier behind it (its recent lists were empty: solve_one_wave)
    bool axis_off = false; // switched off for this controller: it leaves more than half of the batch to the tier (adapt_axis_solver), or no memory for its list
    bool axis_off_by_share = false; // ... the former: sampled again every 256 solves
    int axis_adapt_left = 2; // solves after which its share is still looked at
    long long axis_solves = 0;
    // shared-model tick on the records tier
    long long shared_ric_solves = 0; // solves launched on the tier's shared-model mode
    bool shared_ric_off = false; // ... which a small, constraint-heavy controller leaves after its first solve
    // where the LAST LAUNCHED solve wrote its counters (round-4 advisor: rechoose_layout and the shared tier's choice read the buffers set for
    // the NEXT solve -- with rotating result slabs an uninitialised one)
    const int* last_iter = nullptr;
    const int* last_status = nullptr;
};

struct copra_batch {
    HostPlan hp;
    AdaptState ad;
    // device copies of the plan tables
    int *d_row_step = nullptr, *d_row_ekind = nullptr, *d_row_eoff = nullptr, *d_row_gkind = nullptr,
        *d_row_goff = nullptr;
    double *d_row_f = nullptr, *d_params = nullptr, *d_lb = nullptr, *d_ub = nullptr;
    int *d_row_prev = nullptr, *d_warm = nullptr; // warm start of the shared-model path (copra_batch_set_warm_start)
    // system (owned copies, or borrowed device pointers)
    double *own_A = nullptr, *own_B = nullptr, *own_d = nullptr, *own_x0 = nullptr;
    int axis_order = 0; // the state order of this controller's systems as the (instance, axis)-per-lane solver sees it (FusedPlan::axis_order) ...
    bool axis_order_seen = false; // ... looked at when the first systems were set (see_axis_order, copra_hip.hip)
    bool shared_as_batch = false; // copra_batch_set_shared_system on a controller the (instance, axis)-per-lane solver takes: the model written out per instance (copra_hip.hip)
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
    hipFunction_t jit_lane = nullptr; // ... and the one-instance-per-lane pass in front of it (lmpc_lane.hpp),
    hipFunction_t jit_lane_plain = nullptr; // ... its build without the speculative steps
    bool jit_ric = false; // the code object holds the Riccati-factor tier (lmpc_fused_ric.hpp) of this controller's shape
    int jit_lanes = 64; // lanes per instance the code object was compiled for
    int jit_tri = 0; // ... and whether for the factor-only layout
    // one-instance-per-lane pass in front of the Riccati-factor tier (lmpc_lane.hpp)
    int *d_lane_count = nullptr, *d_lane_list = nullptr; // (two counters, used in turn like the overflow queue's)
    int* d_lane_hist = nullptr; // histogram of the violated-row counts the pass leaves (kLaneHistBins; read once, before the first tier launch)
    double* d_lane_ws = nullptr;
    int* d_lane_seen = nullptr; // ... the same words as the device sees them
    int* h_lane_seen = nullptr; // pinned: [2] the lengths of the last solves' first-tier lists as they arrive (solve_one_wave: the tier's grid follows them)
    int lane_seen_slot = 0, lane_seen_max = 0, lane_seen_first_max = 0; // (of the tier's list | of the first launch's list, which the second chance walks)
    long long lane_seen_solves = 0;
    int *d_axis_list2 = nullptr, *d_axis_count2 = nullptr; // the list the second chance of the (instance, axis)-per-lane solver appends to (the first tier's, then) and its length
    int* d_axis_acc = nullptr; // lmpc_axis.hpp: the words in which the counters of instances on spare lanes meet (FusedPlan::axis_acc; zero between solves)
    double* d_lane_ws2 = nullptr; // the instance-major hand-over blocks of the pass (FusedPlan::lane_ws2)
    int lane_cur = 0; // the counter the last solve appended to
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

// ---- copra_hip.hip ----
typedef void (*fused_kernel_t)(const FusedPlan);
FusedPlan device_plan(const copra_batch* h);
copra_status_t ensure_lds_attr(copra_batch* h);
copra_status_t prepare_riccati(copra_batch* h);
bool use_riccati(copra_batch* h);
fused_kernel_t select_fused_kernel(const FusedPlan& P);
fused_kernel_t select_tier2_kernel(const FusedPlan& P);
// ---- copra_hip_jit.hip: dense-QP kernels compiled at run time for one problem size (copra_qp_dense_specialise) ----
hipFunction_t dense_jit_lookup(int n, int lanes);
